#!/usr/bin/env python3
"""cst_beam_step alone (the two kernels at the end of every decode step: per-row log-softmax + top-2*beam, per-sentence merge and
bookkeeping), timed with an event pair per call on random logits.  usage: python tools/bench_beam_step.py [bsz beam vocab]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
importlib.import_module("chimera-st_amd")
L = importlib.import_module("chimera-st_amd.lib")
from test_decode_engine_gpu import _beam_state

cases = [(32, 5, 10000), (32, 1, 10000), (32, 5, 1000), (32, 10, 10000)]
if len(sys.argv) == 4:
    cases = [tuple(int(a) for a in sys.argv[1:])]
lib = L.load()
for bsz, beam, V in cases:
    for dtype in (torch.bfloat16,):
        max_len = 400
        Vp = (V + 7) // 8 * 8
        g = torch.Generator().manual_seed(1)
        logits = (torch.randn(bsz * beam, Vp, generator=g) * 2).to(dtype).cuda()
        logits[:, 2] -= 20.0  # eos never wins: nothing finishes
        st, d = _beam_state(L, bsz, beam, V, max_len, 1, dtype, logits)
        L.check(lib.cst_beam_init(ctypes.byref(d), L.stream_ptr()), "init")
        for _ in range(5):
            L.check(lib.cst_beam_step(ctypes.byref(d), L.stream_ptr()), "step")
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            L.check(lib.cst_beam_step(ctypes.byref(d), L.stream_ptr()), "step")
        e1.record()
        torch.cuda.synchronize()
        print("bsz %d beam %d V %d %s: %.2f us per cst_beam_step (2 launches), step counter %d" % (bsz, beam, V, dtype, e0.elapsed_time(e1) / n * 1e3, int(st["step"].item())))
