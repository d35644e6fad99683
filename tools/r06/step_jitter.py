#!/usr/bin/env python3
"""Per-update wall time over N updates of the bench batch (every update ends with its one host read, so perf_counter per update IS the
wall time) beside the kernel-time sum: how much of an update is the GPU waiting for the host, and how much that varies run to run.
usage: python tools/r06/step_jitter.py [updates] [gc]   (gc: collector disabled during the updates)"""
import gc
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from argparse import Namespace

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nogc = len(sys.argv) > 2 and sys.argv[2] == "gc"
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
if nogc:
    gc.collect(); gc.disable()
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    trainer.train_step([sample])
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
s = sorted(ts)
print("gc %s: updates %d  min %.2f  median %.2f  p90 %.2f  max %.2f ms" % ("off" if nogc else "on", n, s[0], s[n // 2], s[int(n * 0.9)], s[-1]))
print(" ".join("%.1f" % t for t in ts))
# the same with every C-ABI launch between hipEvents: is a slow update slow on the GPU (kernel-time sum up) or waiting for the host?
L = importlib.import_module("chimera-st_amd.lib")
rows = []
for _ in range(n):
    L.prof_enable(True)
    t0 = time.perf_counter()
    trainer.train_step([sample])
    torch.cuda.synchronize()
    w = 1e3 * (time.perf_counter() - t0)
    tab = L.prof_query(); L.prof_enable(False)
    rows.append((w, sum(v["ms"] for v in tab.values()), tab["gemm"]["ms"], tab["attn_bwd"]["ms"]))
print("bracketed updates: wall / kernel sum / gemm / attn_bwd")
print("  ".join("%.1f/%.1f/%.1f/%.1f" % r for r in rows))
crit, model = trainer.criterion, trainer.model
smp = trainer._prepare_sample(sample)
for _ in range(2):
    loss, ss, log = crit(model, smp); loss.backward(); trainer.optimizer.zero_grad()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); loss, ss, log = crit(model, smp); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter(); trainer.optimizer.zero_grad()
    print("enqueue forward %.1f ms, backward %.1f ms (host), GPU done after %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)))
