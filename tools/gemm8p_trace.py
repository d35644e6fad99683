#!/usr/bin/env python3
"""Cycle breakdown of the persistent 8-phase GEMM per work item (debug build with -DCST_TRACE, see tools/build_trace_lib.sh):
stamps: 0 item start, 1 K loop start, 2 K loop end, 3 next item's first K tile issued, 4 epilogue done."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("chimera-st_amd.lib")
L.LIB_PATH = os.environ["CST_TRACE_LIB"]
K = importlib.import_module("chimera-st_amd.kernels")
M, N, Kd = (int(x) for x in sys.argv[1:4])
epi = sys.argv[4] if len(sys.argv) > 4 else "plain"
dt = torch.bfloat16
A = (torch.rand(M, Kd, device="cuda") * 2 - 1).to(dt); B = (torch.rand(N, Kd, device="cuda") * 2 - 1).to(dt)
C = torch.empty(M, N, device="cuda", dtype=dt)
kw = {}
if epi == "fc1":
    kw = dict(bias=torch.zeros(N, device="cuda", dtype=dt), act=L.ACT_GELU, aux_out=torch.empty(M, N, device="cuda", dtype=dt), ld_aux_out=N)
trace = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
lib = L.load()
d = L.GemmDesc()
def run(ws):
    d.dtype = d.c_dtype = L.BF16; d.a_kmajor = d.b_kmajor = 1; d.M, d.N, d.K = M, N, Kd
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A.data_ptr(), Kd, B.data_ptr(), Kd, C.data_ptr(), N
    d.alpha = 1.0; d.batch0 = d.batch1 = 1; d.split_k = 1
    if kw:
        d.bias, d.bias_mode, d.act, d.aux_out, d.ld_aux_out = kw["bias"].data_ptr(), L.BIAS_COL, kw["act"], kw["aux_out"].data_ptr(), N
    d.workspace, d.workspace_bytes = (ws.data_ptr(), ws.numel() * 8) if ws is not None else (None, 0)
    L.check(lib.cst_gemm(ctypes.byref(d), L.stream_ptr()))
for _ in range(3):
    run(None)
run(trace)
torch.cuda.synchronize()
t = trace.view(256, 8, 8).cpu()
import numpy as np
t = t.numpy().astype(np.float64)
ok = t[:, :, 4] > 0
names = ["setup+prologueB+wait", "K loop", "next setup+DMA issue", "epilogue"]
for k in range(4):
    dts = (t[:, :, k + 1] - t[:, :, k])[ok]
    print("%-24s median %8.0f  mean %8.0f cycles" % (names[k], np.median(dts), dts.mean()))
ok2 = ok & (t[:, :, 5] > 0)
print("  of which: claim read-back + next item setup %8.0f, 4 half-tile DMA issues %8.0f, claim issue %8.0f (medians)" % (
    np.median((t[:, :, 5] - t[:, :, 2])[ok2]), np.median((t[:, :, 6] - t[:, :, 5])[ok2]), np.median((t[:, :, 3] - t[:, :, 6])[ok2])))
tot = (t[:, :, 4] - t[:, :, 0])[ok]
print("item total median %.0f cycles over %d items; item-to-item %s" % (np.median(tot), ok.sum(), np.median((t[:, 1:, 0] - t[:, :-1, 0])[ok[:, 1:]])))
