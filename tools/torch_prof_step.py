#!/usr/bin/env python3
"""torch.profiler view of one training step: which ATen ops (by input shape) run outside the C-ABI kernels."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s2t_w2v2"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=30.0); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--lengths", default="uniform"); ap.add_argument("--dropout", type=float, default=0.1); ap.add_argument("--layerdrop", type=float, default=0.0)
args = ap.parse_args()
device = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, device)
sample = bench.make_batch(tasks, task, args, 0, device)
trainer.train_step([sample]); trainer.train_step([sample]); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    trainer.train_step([sample]); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
print(ka.table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
print("==== ATen ops / memcpy by self device time ====")
rows = [e for e in ka if e.key.startswith("aten::") or "emcpy" in e.key or "emset" in e.key or "copyBuffer" in e.key or "fillBuffer" in e.key]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print("%9.1f us self-dev %9.1f us dev-total %5d x  %-28s %s" % (e.self_device_time_total, e.device_time_total, e.count, e.key[:28], str(e.input_shapes)[:110]))
