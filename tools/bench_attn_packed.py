#!/usr/bin/env python3
"""Packed (variable-length) self-attention as the wav2vec2 layers of the training step call it: 32 sequences with lengths uniform in
[500, 1499] frames (10-30 s), sorted descending, 12 heads x 64, dropout 0.1 — DMA-staged kernels vs the generic ones, same process.
usage (GPU box): python tools/bench_attn_packed.py [rounds]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CF = importlib.import_module("chimera-st_amd.functional")
K = importlib.import_module("chimera-st_amd.kernels")
torch.manual_seed(1)
B, T, H, D = 32, 1499, 12, 64
C = H * D
lens = torch.sort(torch.randint(500, 1500, (B,)), descending=True).values
lens[0] = T
pm = (torch.arange(T)[None, :] >= lens[:, None]).cuda()
plan = CF.plan_packed_rows(pm, int(os.environ.get("MARGIN", 64)))
dt = torch.bfloat16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
qkv = torch.randn(1, plan.rows, 3 * C, device="cuda").to(dt).requires_grad_(True)
do = torch.randn(1, plan.rows, C, device="cuda").to(dt)
pairs = float((lens.double() * (lens.double() + int(os.environ.get("MARGIN", 64)) + 1).clamp(max=T)).sum())
fl = 4.0 * H * D * pairs


def t(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {}
for r in range(rounds):
    for generic in (0, 1):
        if generic:
            os.environ["CST_ATTN_GENERIC"] = "1"
        else:
            os.environ.pop("CST_ATTN_GENERIC", None)
        for p in (0.0, 0.1):
            def fwd():
                return CF.attention_packed(qkv, H, None, False, None, p, plan)
            o = fwd()
            res.setdefault(("fwd", p, generic), []).append(t(fwd))
            res.setdefault(("fwd+bwd", p, generic), []).append(t(lambda: fwd().backward(do)))
print("rows %d of %d (B x T), key/query pairs %.3g" % (plan.rows, B * T, pairs))
for (what, p, generic), v in sorted(res.items()):
    m = sorted(v)[len(v) // 2]
    print("%-8s dropout %.1f %-8s median %.3f ms  min %.3f ms  (%.0f TF/s at %s gemm-equivalents)" % (what, p, "generic" if generic else "dma", m, min(v), (fl if what == "fwd" else 3.5 * fl) / m / 1e9, "2" if what == "fwd" else "7 (2 fwd + 5 bwd)"))
print(dict(K.STATS))
