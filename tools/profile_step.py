#!/usr/bin/env python3
"""Per-call timing of every C-ABI launch in one training step (events around each call), aggregated by op + shape.
Usage (GPU box): python tools/profile_step.py [--model s2t_w2v2|chimera] [--batch 32] [--seconds 30]"""
import argparse, collections, importlib, os, sys
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
K = importlib.import_module("chimera-st_amd.kernels")
recs = []

def wrap(name, fn, keyfn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record()
        recs.append((name, keyfn(*a, **k), s, e))
        return r
    return w

def gemm_key(A, B, C, M, N, Kd, **k):
    return "M=%d N=%d K=%d %s%s b=%dx%d act=%s sk=%s" % (M, N, Kd, "k" if k["a_kmajor"] else "m", "k" if k["b_kmajor"] else "m",
                                                       k.get("batch0", 1), k.get("batch1", 1), k.get("act", 0), k.get("split_k", -1))
K.gemm = wrap("gemm", K.gemm, gemm_key)
K.attn_fwd_desc = wrap("attn_fwd", K.attn_fwd_desc, lambda d: "B%d H%d Tq%d Tk%d" % (d.B, d.H, d.Tq, d.Tk))
K.attn_bwd_desc = wrap("attn_bwd", K.attn_bwd_desc, lambda d: "B%d H%d Tq%d Tk%d" % (d.B, d.H, d.Tq, d.Tk))
for n in ("layernorm_fwd", "layernorm_bwd", "attn_fwd", "attn_bwd", "conv0_fwd", "conv0_bwd", "glu_fwd", "glu_bwd", "act_bwd", "colsum",
          "col2im1d", "mask_rows", "ls_ce_fwd", "ls_ce_bwd", "adam_step", "sumsq"):
    setattr(K, n, wrap(n, getattr(K, n), lambda *a, **k: "x".join(str(tuple(t.shape)) for t in a[:1] if torch.is_tensor(t))))
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record(); trainer.train_step([sample]); e0.record(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, key, s, e in recs:
    a = agg[(name, key)]; a[0] += 1; a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print("step %.1f ms; sum of C-ABI calls %.1f ms over %d calls" % (s0.elapsed_time(e0), tot, len(recs)))
for (name, key), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    fl = ""
    if name == "gemm":
        import re
        M, N, Kd = (int(x) for x in re.findall(r"[MNK]=(\d+)", key)); b = re.search(r"b=(\d+)x(\d+)", key)
        fl = "%7.0f TF/s" % (2.0 * M * N * Kd * int(b.group(1)) * int(b.group(2)) * n / ms / 1e9)
    print("%8.2f ms %4d x %-14s %-62s %s" % (ms, n, name, key, fl))
