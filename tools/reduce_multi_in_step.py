#!/usr/bin/env python3
"""The cst_reduce_multi launches of one update of the bench workload: items, bytes read, device time, TB/s."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
args = argparse.Namespace(batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
device = torch.device("cuda", 0)
L.load()
trainer, task, tasks, ns = bench.build(args, device)
sample = bench.make_batch(tasks, task, args, 0, device)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
rec = []
orig = K.reduce_multi
def spy(items):
    for at in range(0, len(items), 64):
        chunk = items[at:at + 64]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(chunk); e1.record()
        rec.append((chunk, e0, e1))
K.reduce_multi = spy
trainer.train_step([sample]); torch.cuda.synchronize()
tb = tm = 0.0
for chunk, e0, e1 in rec:
    b0 = sum(P * Lr * 4 for (_, _, _, Lr, P, _, o) in chunk if o == 0); b1 = sum(P * Lr * 4 for (_, _, _, Lr, P, _, o) in chunk if o == 1)
    ms = e0.elapsed_time(e1); tb += b0 + b1; tm += ms
    big = sorted(chunk, key=lambda it: -it[3] * it[4])[:3]
    print("%3d items  order0 %8.1f MB  order1 %7.1f MB  %.3f ms  %.2f TB/s   largest: %s" % (len(chunk), b0 / 1e6, b1 / 1e6, ms, (b0 + b1) / ms / 1e9,
          ", ".join("L=%d P=%d o%d" % (it[3], it[4], it[6]) for it in big)))
print("total %.1f MB in %.3f ms = %.2f TB/s over %d launches" % (tb / 1e6, tm, tb / tm / 1e9, len(rec)))
