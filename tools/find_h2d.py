#!/usr/bin/env python3
"""Which Python lines of one training update copy host memory to the device (each pageable copy waits for the stream): torch.profiler
with stacks, CPU-side `aten::_to_copy` / `aten::copy_` ops whose child is a `Memcpy HtoD`, grouped by the innermost package frame."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from argparse import Namespace
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model=os.environ.get("MODEL", "s2t_w2v2"), dropout=0.1, layerdrop=0.0)
device = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, device)
sample = bench.make_batch(tasks, task, args, 0, device)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_step([sample]); torch.cuda.synchronize()
evs = prof.events()
h2d = [e for e in evs if "Memcpy HtoD" in e.name]
print("%d host-to-device copies in one update" % len(h2d))
from collections import Counter
sites = Counter()
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    kids = [k for k in e.kernels if "HtoD" in k.name]
    if not kids or any(any("HtoD" in kk.name for kk in c.kernels) for c in e.cpu_children):
        continue  # keep the innermost op that owns the copy
    frames = [f for f in (e.stack or []) if "chimera-st_amd" in f or "bench.py" in f]
    sites[(e.name, frames[0] if frames else (e.stack[0] if e.stack else "?"))] += 1
for (name, fr), n in sites.most_common():
    print("%3d x %-18s %s" % (n, name, fr))
