#!/usr/bin/env python3
"""Host-side time of one training update vs its wall time: the update is ENQUEUED (no sync) and the Python time to enqueue it is
compared with the synchronised step time.  If host ~= wall the step is launch-bound, not GPU-bound.  With `prof` as argv[1] a
cProfile of 3 updates is printed."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from argparse import Namespace
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
importlib.import_module("chimera-st_amd.distributed").distributed_init()  # RANK/WORLD_SIZE(+CST_DDP_FORCE=1): time the collective path
trainer, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
host, wall = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.train_step([sample])     # ends with the one .tolist() sync of the update
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    wall.append(time.perf_counter() - t0)
    host.append(t1 - t0)
print("wall %.1f ms  (train_step returns after %.1f ms)" % (1e3 * sorted(wall)[3], 1e3 * sorted(host)[3]))
# host time without the final sync: patch tolist away by timing forward+backward enqueue only
crit, model = trainer.criterion, trainer.model
s = trainer._prepare_sample(sample)
for _ in range(2):
    loss, ss, log = crit(model, s); loss.backward(); trainer.optimizer.zero_grad()
torch.cuda.synchronize()
t0 = time.perf_counter(); loss, ss, log = crit(model, s); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter()
torch.cuda.synchronize(); t3 = time.perf_counter()
print("enqueue forward %.1f ms, backward %.1f ms (host), GPU done after %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        trainer.train_step([sample])
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
