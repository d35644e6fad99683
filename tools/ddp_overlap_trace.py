#!/usr/bin/env python3
"""Per-bucket gradient all-reduce against the backward kernels, from one rocprofv3 kernel trace per rank.

The data-parallel update (chimera-st_amd/distributed.py) launches one RCCL all-reduce per gradient bucket from autograd hooks
while the backward pass is still queueing GEMMs; whether the wire time hides behind the backward kernels is the whole of the
multi-GPU scaling story (reference: fairseq legacy_distributed_data_parallel.py:130-170 reduces AFTER backward and
hides nothing).  This tool produces the evidence on an N-GPU node:

    python tools/ddp_overlap_trace.py --gpus 8 [--steps 4 --warmup 2] [--out gpurun_out/ddp_trace]

  * starts N children, one per GPU, each one `rocprofv3 --kernel-trace --output-format csv -d <out>/rank<r> -- python3 bench.py
    --gpus N ...` — the program stands directly behind `--` and RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* travel in the child's
    environment (no `env`, no shell, no torchrun re-exec between the profiler and the program);
  * reads every rank's kernel trace, cuts it into updates at the optimizer kernel, and prints per update and per rank: the
    backward window (first to last non-collective kernel between the loss kernel and the optimizer), every collective kernel's
    start / end relative to that window, how much of it ran while a compute kernel of the same rank was executing ("hidden") and
    how much ran alone ("exposed"), and the exposed tail after the last backward kernel;
  * `--parse DIR` only re-reads traces collected earlier.

On one GPU there is no collective (world size 1 builds no process group): the tool then reports the backward window and says so.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLLECTIVE = ("ncclDevKernel", "rccl", "ncclKernel")
OPTIMIZER = ("adam_kernel", "adam_step", "cst_adam")
LOSS = ("ls_ce", "contrastive")


def launch(args):
    os.makedirs(args.out, exist_ok=True)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(args.port), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "TMPDIR": env.get("TMPDIR", "/tmp")})
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(args.out, "rank%d" % r), "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(args.gpus), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-roofline", "--model", args.model]
        log = open(os.path.join(args.out, "rank%d.log" % r), "w")
        procs.append((subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd="/tmp"), log))
    rc = 0
    for p, log in procs:
        try:
            rc |= p.wait(timeout=args.timeout)
        except subprocess.TimeoutExpired:
            p.kill()  # the exact child this tool started
            rc |= 1
        log.close()
    return rc


def read_trace(rank_dir):
    """[(start_ns, end_ns, name)] of one rank, sorted by start."""
    rows = []
    for f in glob.glob(os.path.join(rank_dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for rec in csv.DictReader(fh):
                rows.append((int(rec["Start_Timestamp"]), int(rec["End_Timestamp"]), rec["Kernel_Name"]))
    rows.sort()
    return rows


def is_any(name, keys):
    return any(k in name for k in keys)


def covered(seg, busy):
    """Length of seg = (a, b) covered by the union of the sorted, merged intervals in busy."""
    a, b = seg
    tot = 0
    for s, e in busy:
        if e <= a:
            continue
        if s >= b:
            break
        tot += min(b, e) - max(a, s)
    return tot


def merge(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def split_updates(rows):
    """Cut a rank's trace at the end of each run of optimizer kernels."""
    ups, cur, in_opt = [], [], False
    for r in rows:
        opt = is_any(r[2], OPTIMIZER)
        if in_opt and not opt:
            ups.append(cur)
            cur = []
        cur.append(r)
        in_opt = opt
    if cur:
        ups.append(cur)
    return [u for u in ups if any(is_any(r[2], OPTIMIZER) for r in u)]


def analyse_update(rows):
    coll = [r for r in rows if is_any(r[2], COLLECTIVE)]
    comp = [r for r in rows if not is_any(r[2], COLLECTIVE)]
    loss_i = [i for i, r in enumerate(comp) if is_any(r[2], LOSS)]
    opt_i = [i for i, r in enumerate(comp) if is_any(r[2], OPTIMIZER)]
    lo = loss_i[0] if loss_i else 0                      # the loss kernel: forward ends, backward starts
    hi = opt_i[0] if opt_i else len(comp)                # first optimizer kernel
    bwd = comp[lo:hi]
    if not bwd:
        return None
    b0, b1 = bwd[0][0], max(r[1] for r in bwd)
    busy = merge([(r[0], r[1]) for r in comp])
    out = {"backward_ms": (b1 - b0) / 1e6, "collectives": []}
    hidden = exposed = 0
    for s, e, name in coll:
        h = covered((s, e), busy)
        hidden += h
        exposed += (e - s) - h
        out["collectives"].append({"start_ms": (s - b0) / 1e6, "end_ms": (e - b0) / 1e6, "ms": (e - s) / 1e6,
                                   "hidden_ms": h / 1e6, "after_backward_ms": max(0, e - max(s, b1)) / 1e6, "kernel": name[:60]})
    out["collective_ms"] = sum(c["ms"] for c in out["collectives"])
    out["hidden_ms"] = hidden / 1e6
    out["exposed_ms"] = exposed / 1e6
    out["tail_after_backward_ms"] = max([0.0] + [(c["end_ms"] * 1e6 + b0 - b1) / 1e6 for c in out["collectives"]])
    out["update_ms"] = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    return out


def report(out_dir, verbose=True):
    ranks = sorted(glob.glob(os.path.join(out_dir, "rank*[0-9]")))
    summary = {}
    for rd in ranks:
        rows = read_trace(rd)
        ups = [analyse_update(u) for u in split_updates(rows)]
        ups = [u for u in ups if u]
        summary[os.path.basename(rd)] = ups
        if not verbose:
            continue
        print("== %s: %d kernels, %d updates" % (os.path.basename(rd), len(rows), len(ups)))
        for i, u in enumerate(ups):
            print("  update %d: %.2f ms, backward window %.2f ms, %d collective kernels %.2f ms (hidden %.2f, exposed %.2f, tail after backward %.2f)"
                  % (i, u["update_ms"], u["backward_ms"], len(u["collectives"]), u["collective_ms"], u["hidden_ms"], u["exposed_ms"],
                     u["tail_after_backward_ms"]))
            if i == len(ups) - 1:  # the last (steady-state) update in full
                for j, c in enumerate(u["collectives"]):
                    print("    bucket %2d: %8.3f -> %8.3f ms of the backward window (%.3f ms, %.3f hidden)  %s"
                          % (j, c["start_ms"], c["end_ms"], c["ms"], c["hidden_ms"], c["kernel"]))
        if ups and not ups[-1]["collectives"]:
            print("  no collective kernels in this trace (world size 1 builds no process group)")
    with open(os.path.join(out_dir, "ddp_overlap_summary.json"), "w") as fh:
        json.dump(summary, fh)
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="s2t_w2v2")
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ddp_trace"))
    ap.add_argument("--parse", default=None, help="only analyse the traces under this directory")
    ap.add_argument("--collective-pattern", action="append", default=[],
                    help="extra kernel-name substring to count as a collective (a 1-rank RCCL all-reduce is an "
                         "__amd_rocclr_copyBuffer, not an ncclDevKernel)")
    args = ap.parse_args()
    global COLLECTIVE
    COLLECTIVE = tuple(COLLECTIVE) + tuple(args.collective_pattern)
    if args.parse:
        report(args.parse)
        return 0
    rc = launch(args)
    report(args.out)
    return rc


if __name__ == "__main__":
    sys.exit(main())
