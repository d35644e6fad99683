#!/usr/bin/env python3
"""Which weight-gradient GEMMs of one full-size update receive live-tile stamps, and the fraction of 64-token blocks that are live
(the rest is skipped: cst_gemm_desc.k_live)."""
import os, sys, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from argparse import Namespace
K = importlib.import_module("chimera-st_amd.kernels")
args = Namespace(gpus=1, steps=1, warmup=0, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
tr, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
orig = K.gemm
fr = []
def g(*a, **kw):
    kl = kw.get("k_live")
    if kl is not None:
        fr.append((a[3], a[4], a[5], float((kl[0] == kl[1]).float().mean())))
    return orig(*a, **kw)
K.gemm = g
importlib.import_module("chimera-st_amd.functional").K.gemm = g
tr.train_step([sample])
print("launches with stamps:", len(fr))
for r in fr[:6] + fr[-6:]: print(r)
sl = sample["net_input"]["src_lengths"].float()
print("mean valid fraction of frames:", float((sl / sl.max()).mean()))
