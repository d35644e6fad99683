import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
n = int(sys.argv[1])
if n: torch.set_num_threads(n)
import bench
from argparse import Namespace
def stat():
    d = {}
    for l in open("/sys/fs/cgroup/cpu.stat"):
        k, v = l.split(); d[k] = int(v)
    return d
s0 = stat(); t0 = time.perf_counter()
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, dev)
t1 = time.perf_counter()
sample = bench.make_batch(tasks, task, args, 0, dev)
t2 = time.perf_counter()
for _ in range(2): trainer.train_step([sample])
torch.cuda.synchronize(); t3 = time.perf_counter()
s1 = stat()
print("threads %d: build %.1f s, batch %.2f s, 2 warm-up updates %.1f s; throttled periods %d of %d, throttled %.1f s" % (torch.get_num_threads(), t1 - t0, t2 - t1, t3 - t2, s1["nr_throttled"] - s0["nr_throttled"], s1["nr_periods"] - s0["nr_periods"], (s1["throttled_usec"] - s0["throttled_usec"]) / 1e6))
