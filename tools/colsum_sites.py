#!/usr/bin/env python3
"""Which column sums (bias gradients) run as their own launches in one update of the bench workload: shape, call site, device time."""
import argparse, collections, importlib, os, sys, traceback
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
def wrap(name, fn):
    def f(x, *a, **kw):
        fr = [f for f in traceback.extract_stack() if "chimera-st_amd/" in f.filename and "kernels.py" not in f.filename]
        where = "%s:%d" % (fr[-1].filename.split("chimera-st_amd/")[-1], fr[-1].lineno) if fr else "?"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(x, *a, **kw); e1.record()
        rec.append((name, tuple(x.shape), where, e0, e1))
        return r
    return f
K.colsum = wrap("colsum", K.colsum)
K.dropout_colsum = wrap("dropout_colsum", K.dropout_colsum)
F = importlib.import_module("chimera-st_amd.functional")
trainer.train_step([sample]); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shp, where, e0, e1 in rec:
    a = agg.setdefault((name, shp, where), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
print("%d column-sum launches in one update" % len(rec))
for (name, shp, where), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%3d x %-15s %-18s %-28s %.3f ms" % (n, name, shp, where, ms))
