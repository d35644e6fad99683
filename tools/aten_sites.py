#!/usr/bin/env python3
"""Where do the ATen kernels of one update come from (copies, reductions, fills — everything that is not a libcst_hip launch)?
Runs bench.py's update under torch.profiler with Python stacks and prints, per (kernel-launching ATen op, first frame inside
chimera-st_amd/), the calls per update and the device time.  usage: python tools/aten_sites.py [bench.py flags]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

ap_args = sys.argv[1:]
sys.argv = [sys.argv[0]] + ap_args
import argparse  # noqa: E402

ns = argparse.Namespace(gpus=1, steps=4, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
for a in ap_args:
    if a.startswith("--model="):
        ns.model = a.split("=", 1)[1]
dev = torch.device("cuda", 0)
trainer, task, tasks, margs = B.build(ns, dev)
sample = B.make_batch(tasks, task, ns, 0, dev)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import traceback  # noqa: E402

# (torch.profiler's with_stack comes back empty on this build: take the Python stack ourselves, per dispatched ATen op)
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

py_sites = collections.Counter()
py_bytes = collections.Counter()  # bytes of the result tensors (copies, fills and elementwise ops: what the launch has to write)


class _Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        fr = [f for f in traceback.extract_stack() if "chimera-st_amd/" in f.filename]
        where = "%s:%d" % (fr[-1].filename.split("chimera-st_amd/")[-1], fr[-1].lineno) if fr else "?"
        out = func(*args, **(kwargs or {}))
        if any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or "empty" in name or "zeros" in name:
            py_sites[(name, where)] += 1
            if torch.is_tensor(out) and out.is_cuda:
                py_bytes[(name, where)] += out.numel() * out.element_size()
        return out


with _Sites():
    trainer.train_step([sample])
torch.cuda.synchronize()
print("ATen ops dispatched on device tensors in one update, by call site (views and metadata ops included):")
skip = ("view", "transpose", "permute", "as_strided", "slice", "select", "expand", "unsqueeze", "squeeze", "detach", "alias", "reshape", "t.default", "empty", "_unsafe_view", "unbind", "split", "narrow", "size", "stride", "is_")
for (name, where), n in sorted(py_sites.items(), key=lambda kv: -kv[1]):
    if not any(k in name for k in skip):
        print("%5d  %-40s %-32s %10.3f MB written" % (n, name, where, py_bytes[(name, where)] / 1e6))
print()

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_step([sample])
    torch.cuda.synchronize()
sites = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::") or not ev.kernels:
        continue
    if any(c.kernels for c in ev.cpu_children if c.name.startswith("aten::")):
        continue  # keep the innermost op that owns the launch
    frames = [f for f in (ev.stack or []) if "chimera-st_amd/" in f]
    frame = frames[0].split("chimera-st_amd/")[-1] if frames else ((ev.stack or ["?"])[0])
    k = (ev.name + " -> " + ev.kernels[0].name[:40], frame)
    sites[k][0] += 1
    sites[k][1] += sum(kk.duration for kk in ev.kernels)
tot = sum(v[1] for v in sites.values())
print("ATen ops that launched device work in one update: %d calls, %.3f ms" % (sum(v[0] for v in sites.values()), tot / 1e3))
for (name, frame), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%5d  %8.1f us  %-70s %s" % (n, us, name, frame))
