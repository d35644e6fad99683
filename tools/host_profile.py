#!/usr/bin/env python3
"""Where does the HOST spend its time while it enqueues one update?  cProfile over a few updates, functions of this package by
cumulative and by own time (per update).  The step is GPU-bound on a fast host (enqueue ~27 ms against ~62 ms of GPU time) and partly
host-bound on a slow one: the backward pass of the 512-wide layers is ~23 launches of 10-40 us per layer."""
import cProfile, importlib, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from argparse import Namespace
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model=os.environ.get("MODEL", "s2t_w2v2"), dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
N = 5
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    trainer.train_step([sample])
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    buf = io.StringIO()
    st = pstats.Stats(pr, stream=buf)
    st.sort_stats(key).print_stats(70)
    print("==== by %s (totals over %d updates) ====" % (key, N))
    for line in buf.getvalue().splitlines():
        if "chimera-st_amd" in line or "built-in" in line or "method" in line or "ncalls" in line:
            print(line[:190])
