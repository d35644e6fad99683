#!/usr/bin/env python3
"""Which gradient buckets are launched from backward hooks (overlapped) and which parameters nobody accounted for, per update,
for the two bench models on a 1-rank RCCL group with the collective path forced on (CST_DDP_FORCE=1)."""
import os, sys, importlib, torch
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CST_DDP_FORCE="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from argparse import Namespace
importlib.import_module("chimera-st_amd.distributed").distributed_init()
for model in ("s2t_w2v2", "chimera"):
    args = Namespace(gpus=1, steps=1, warmup=0, batch=4, seconds=4.0, lengths="uniform", dtype="bf16", model=model, dropout=0.1, layerdrop=0.05)
    dev = torch.device("cuda", 0)
    tr, task, tasks, ns = bench.build(args, dev)
    sample = bench.make_batch(tasks, task, args, 0, dev)
    names = [n for n, _ in tr.get_model().named_parameters()]
    for it in range(3):
        tr.train_step([sample])
        r = tr.model.reducer
        print(model, "step", it, "buckets", len(r.buckets), "launched from hooks", r.last_early, "missing", [names[i] for i in r.last_missing][:12], flush=True)
