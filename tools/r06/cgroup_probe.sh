echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "--- cpu.stat before"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null
python - <<'PY'
import os, time, sys, importlib
sys.path.insert(0, os.getcwd())
import torch, bench
from argparse import Namespace
print("affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
for _ in range(3): trainer.train_step([sample])
torch.cuda.synchronize()
def stat():
    d = {}
    try:
        for l in open("/sys/fs/cgroup/cpu.stat"): k, v = l.split(); d[k] = int(v)
    except Exception as e: d["err"] = str(e)
    return d
s0 = stat(); c0 = os.times(); t0 = time.perf_counter()
ts = []
for _ in range(60):
    a = time.perf_counter(); trainer.train_step([sample]); ts.append(1e3 * (time.perf_counter() - a))
torch.cuda.synchronize()
w = time.perf_counter() - t0; c1 = os.times(); s1 = stat()
print("wall %.2f s  user %.2f  sys %.2f  -> %.2f CPUs busy" % (w, c1.user - c0.user, c1.system - c0.system, (c1.user - c0.user + c1.system - c0.system) / w))
print("cgroup delta:", {k: s1[k] - s0[k] for k in s1 if isinstance(s1[k], int)})
print(" ".join("%.1f" % t for t in ts))
import threading, subprocess
print("threads in process:", len(os.listdir("/proc/self/task")))
PY
