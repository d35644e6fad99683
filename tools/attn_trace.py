#!/usr/bin/env python3
"""Where a workgroup of the DMA-staged attention forward spends its time (debug build with -DCST_TRACE: tools/build_trace_lib.sh).
Stamps of wave 0: 0 kernel entry, 1 prologue work issued (DMA of tile 0, Q fragments, hashes), 2 first barrier passed, 3 tile loop done,
4 outputs stored.  Also: how busy the CU slots are over the launch (100 MHz real-time counter).
usage (GPU box): CST_TRACE_LIB=tools/trace/libcst_trace.so python tools/attn_trace.py [dropout p]"""
import ctypes, importlib, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("chimera-st_amd.lib")
L.LIB_PATH = os.environ["CST_TRACE_LIB"]
K = importlib.import_module("chimera-st_amd.kernels")
B, T, H, D = int(os.environ.get("ATT_B", 32)), int(os.environ.get("ATT_T", 1499)), 12, 64
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dt = torch.bfloat16
q = torch.randn(B, T, H * D, device="cuda").to(dt); k = torch.randn_like(q); v = torch.randn_like(q)
o = torch.empty_like(q); lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
nblk = ((T + 127) // 128) * H * B
trace = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
d = K.attn_desc(q, k, v, o, lse, H, D, None, False, 0.125, "bt", "bt", p, 99)
for _ in range(3):
    K.attn_fwd_desc(d)
d.delta = trace.data_ptr()
K.attn_fwd_desc(d)
torch.cuda.synchronize()
t = trace.view(nblk, 16).cpu().numpy().astype(np.float64)
cyc, rt = t[:, 0:10:2], t[:, 1:10:2]
names = ["entry -> prologue issued", "wait for tile 0 + barrier", "tile loop", "epilogue (stores)"]
for i in range(4):
    dd = cyc[:, i + 1] - cyc[:, i]
    print("%-28s median %8.0f  mean %8.0f  p90 %8.0f shader cycles" % (names[i], np.median(dd), dd.mean(), np.percentile(dd, 90)))
nt = t[:, 11]
loop = cyc[:, 3] - cyc[:, 2]
print("tiles per workgroup %d; loop cycles per tile: median %.0f  mean %.0f" % (int(nt.max()), np.median(loop / nt), (loop / nt).mean()))
total = cyc[:, 4] - cyc[:, 0]
print("workgroup lifetime: median %.0f cycles; fixed part (everything but the loop) %.1f %%" % (np.median(total), 100 * (1 - loop.sum() / total.sum())))
# occupancy over time from the real-time counter (100 MHz ticks)
t0, t1 = rt[:, 0].min(), rt[:, 4].max()
span = t1 - t0
busy = (rt[:, 4] - rt[:, 0]).sum()
print("launch span %.1f us; sum of workgroup lifetimes %.1f us = %.2f workgroups resident on average (768 slots at 3 per CU)" % (span / 100, busy / 100, busy / span))
edges = np.linspace(t0, t1, 11)
res = []
for a, b_ in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(rt[:, 4], b_) - np.maximum(rt[:, 0], a), 0, None).sum() / (b_ - a)
    res.append(ov)
print("resident workgroups per tenth of the launch: " + " ".join("%.0f" % r for r in res))
