#!/usr/bin/env python3
"""Per-shape GEMM times INSIDE one training update (hipEvent pair per launch: cst_prof_dump): which launches the 48 ms GEMM class is
made of, and how far each shape is from its standalone rate.  usage (GPU box): python tools/gemm_shapes_in_step.py [--model chimera]"""
import argparse, collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
lib = importlib.import_module("chimera-st_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s2t_w2v2")
a = ap.parse_args()
args = argparse.Namespace(batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model=a.model, dropout=0.1, layerdrop=0.0)
device = torch.device("cuda", 0)
lib.load()
trainer, task, tasks, ns = bench.build(args, device)
sample = bench.make_batch(tasks, task, args, 0, device)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
lib.prof_enable(True)
trainer.train_step([sample])
torch.cuda.synchronize()
recs = lib.prof_dump(0)  # CST_K_GEMM
lib.prof_enable(False)
agg = collections.OrderedDict()
for ms, fl, by, tag in recs:
    e = agg.setdefault(tag, [0, 0.0, 0.0])
    e[0] += 1; e[1] += ms; e[2] += fl
tot = sum(e[1] for e in agg.values())
print("%d GEMM-class launches, %.2f ms per update" % (len(recs), tot))
print("%5s %9s %8s %8s  %s" % ("calls", "ms/upd", "avg ms", "TF/s", "launch"))
for tag, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%5d %9.3f %8.4f %8.0f  %s" % (n, ms, ms / n, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, tag))
