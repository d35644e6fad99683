#!/usr/bin/env python3
"""cst_reduce_multi on the item mix one training update leaves behind: the split-K slabs of the 512-wide layers'
weight gradients (order 0) and the LayerNorm / column-sum row-block partials (order 1), timed separately.
usage: python tools/bench_reduce_multi.py"""
import os
import sys
from importlib import import_module

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = import_module("chimera-st_amd.kernels")
L = import_module("chimera-st_amd.lib")


def items_of(specs):
    keep, items, nbytes = [], [], 0
    for Lr, P, stride, order in specs:
        src = torch.randn(P * stride, device="cuda")
        dst = torch.empty(Lr, dtype=torch.bfloat16, device="cuda")
        keep += [src, dst]
        items.append((src.data_ptr(), dst.data_ptr(), stride, Lr, P, L.dtype_code(dst.dtype), order))
        nbytes += P * Lr * 4 + Lr * 2
    return keep, items, nbytes


def timed(name, specs, reps=20):
    keep, items, nbytes = items_of(specs)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    K.reduce_multi(items)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        flush.zero_()  # the partials were written long before the flush: cold
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.reduce_multi(items)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / reps
    print("%-58s %4d items %7.1f MB  %.3f ms  %.2f TB/s" % (name, len(items), nbytes / 1e6, ms, nbytes / ms / 1e9))


enc = []
for _ in range(12):
    enc += [(1536 * 512, 5, 1536 * 512, 0), (512 * 512, 15, 512 * 512, 0), (2048 * 512, 4, 2048 * 512, 0), (512 * 2048, 4, 512 * 2048, 0)]
timed("order 0: split-K slabs of 12 encoder layers", enc)
ln = []
for _ in range(25):
    ln += [(768, 1024, 2 * 768, 1), (768, 1024, 2 * 768, 1)]
for _ in range(45):
    ln += [(512, 1024, 2 * 512, 1), (512, 1024, 2 * 512, 1)]
timed("order 1: LayerNorm dgamma / dbeta partials (70 LayerNorms)", ln)
cs = [(768, 256, 768, 1)] * 24 + [(2304, 86, 2304, 1)] * 12 + [(3072, 64, 3072, 1)] * 12
timed("order 1: column-sum partials of the wav2vec2 layers", cs)
