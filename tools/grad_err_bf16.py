#!/usr/bin/env python3
"""bf16 gradient error of the tiny golden models against the reference fixtures: global and worst per-tensor relative L2."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from importlib import import_module
import numpy as np, torch
from conftest import golden_sample, load_golden
import test_model_gpu as T
g = load_golden("chimera_tiny.npz")
model, task, args = T.build_from_golden(g, "chimera", torch.bfloat16)
crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
sample = T.to_cuda(golden_sample(g)); model.train(); model.zero_grad()
loss, _, log = crit(model, sample); loss.backward()
num = den = 0.0; per = []
for name, p in model.named_parameters():
    ref = np.asarray(g["grad/" + name], dtype=np.float64)
    got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().cpu().numpy()
    e, r = float(((got - ref) ** 2).sum()), float((ref ** 2).sum()); num += e; den += r; per.append((name, e, r))
print("global rel-L2 %.4e" % ((num / den) ** 0.5))
rows = sorted(((e / r) ** 0.5, r / den, n) for n, e, r in per if r >= 1e-6 * den)
print("median %.3e" % rows[len(rows) // 2][0])
for x in rows[-6:]: print("%.3e  share %.2e  %s" % x)
