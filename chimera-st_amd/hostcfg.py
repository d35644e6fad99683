"""Host-side thread budget.  PyTorch sizes its intra-op pool from the machine's thread count; under a container CPU quota (cgroup
`cpu.max`) that is far more threads than the process may run, and every parallel CPU op — parameter initialisation, collating and
pinning a batch, a checkpoint's dtype casts — then burns the quota on spinning pool threads and gets the whole process throttled
for the rest of the scheduler period, the thread that enqueues GPU work included.  Measured on an MI355X box (256 hardware threads,
quota 16): building the bench model 2.0 -> 1.2 s, one synthetic batch 0.30-0.49 -> 0.03 s, throttled scheduler periods 12 of 27 -> 0
(profiles/r06_host_threads.txt).  The reference leaves the pool at PyTorch's default (it has no such code); this is host plumbing
of this build, switched off with CST_HOST_THREADS=0 or set explicitly with CST_HOST_THREADS=N."""
import math
import os

import torch


def _read(path):
    try:
        with open(path) as f:
            return f.read().split()
    except OSError:
        return None


def usable_cpus(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may actually keep busy: the scheduler affinity mask, cut to the cgroup CPU quota (v2 `cpu.max`, or v1
    `cpu/cpu.cfs_quota_us` / `cpu.cfs_period_us`) when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    v2 = _read(os.path.join(cgroup_root, "cpu.max"))
    if v2 and len(v2) == 2 and v2[0] != "max":
        n = min(n, max(1, math.ceil(int(v2[0]) / int(v2[1]))))
    else:
        q, per = _read(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")), _read(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us"))
        if q and per and int(q[0]) > 0 and int(per[0]) > 0:
            n = min(n, max(1, math.ceil(int(q[0]) / int(per[0]))))
    return max(1, n)


def limit_host_threads(ranks_on_host=1):
    """Cap torch's intra-op pool at this process's share of the usable CPUs (never raises it).  Returns the pool size in force."""
    env = os.environ.get("CST_HOST_THREADS")
    if env is not None:
        if int(env) > 0:
            torch.set_num_threads(int(env))
        return torch.get_num_threads()
    share = max(1, usable_cpus() // max(1, int(ranks_on_host)))
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)
    return torch.get_num_threads()
