#!/usr/bin/env python3
"""Average the model parameters of several checkpoints (what chimera/tools/eval-average-checkpoint.py drives through fairseq's
scripts/average_checkpoints.py before the final evaluation: `--num-epoch-checkpoints N --checkpoint-upper-bound E` picks
checkpoint{E-N+1..E}.pt of a directory).  Parameters are accumulated in fp32 (integer buffers are taken from the first file) and
written back in the first file's dtype as a reference-format checkpoint (args / extra_state of the first input, no optimizer state).

  python tools/average_checkpoints.py --inputs <dir> --num-epoch-checkpoints 7 --checkpoint-upper-bound 30 --output avg.pt
  python tools/average_checkpoints.py --inputs a.pt b.pt c.pt --output avg.pt"""
import argparse
import collections
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def average(paths):
    CU = importlib.import_module("chimera-st_amd.checkpoint_utils")
    first, acc, n = None, collections.OrderedDict(), 0
    for p in paths:
        state = CU.load_checkpoint_to_cpu(p)
        if first is None:
            first = state
        keys = list(state["model"].keys())
        if n and keys != list(acc.keys()):
            raise KeyError("checkpoint %s has a different parameter set than %s" % (p, paths[0]))
        for k, v in state["model"].items():
            if n == 0:
                acc[k] = v.clone().float() if v.is_floating_point() else v.clone()
            elif v.is_floating_point():
                acc[k] += v.float()
        n += 1
    model = collections.OrderedDict((k, (v / n).to(first["model"][k].dtype) if v.is_floating_point() else v) for k, v in acc.items())
    out = {"cfg": None, "args": first.get("args"), "model": model, "optimizer_history": first["optimizer_history"][-1:],
           "extra_state": first.get("extra_state", {})}
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", nargs="+", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--num-epoch-checkpoints", type=int, default=None)
    ap.add_argument("--checkpoint-upper-bound", type=int, default=None)
    args = ap.parse_args(argv)
    paths = args.inputs
    if args.num_epoch_checkpoints is not None:
        assert len(paths) == 1 and os.path.isdir(paths[0]), "--num-epoch-checkpoints takes one checkpoint directory"
        hi = args.checkpoint_upper_bound
        if hi is None:
            hi = max(int(f[len("checkpoint"):-3]) for f in os.listdir(paths[0]) if f.startswith("checkpoint") and f[len("checkpoint"):-3].isdigit())
        paths = [os.path.join(paths[0], "checkpoint%d.pt" % e) for e in range(hi - args.num_epoch_checkpoints + 1, hi + 1)]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:
        raise FileNotFoundError("checkpoints not found: %s" % missing)
    torch.save(average(paths), args.output)
    print("averaged %d checkpoints -> %s" % (len(paths), args.output))


if __name__ == "__main__":
    main()
