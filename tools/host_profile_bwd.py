#!/usr/bin/env python3
"""cProfile of the autograd engine's thread during the backward pass of one update (the Python of this package's backward
functions runs there, invisible to a profile of the calling thread): enabled by a hook on the loss, disabled by the hook of the
last gradient to arrive (the first CNN layer's weight)."""
import cProfile, importlib, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from argparse import Namespace
args = Namespace(gpus=1, steps=8, warmup=2, batch=32, seconds=30.0, lengths="uniform", dtype="bf16", model=os.environ.get("MODEL", "s2t_w2v2"), dropout=0.1, layerdrop=0.0)
dev = torch.device("cuda", 0)
trainer, task, tasks, ns = bench.build(args, dev)
sample = bench.make_batch(tasks, task, args, 0, dev)
for _ in range(3):
    trainer.train_step([sample])
torch.cuda.synchronize()
pr = cProfile.Profile()
model, crit = trainer.model, trainer.criterion
last = dict(model.named_parameters())["encoder.wav2vec_model.feature_extractor.conv_layers.0.0.weight"]
h = last.register_post_accumulate_grad_hook(lambda p: pr.disable())
K = importlib.import_module("chimera-st_amd.kernels")
N = 4
import time
tb = 0.0
for _ in range(N):
    trainer.optimizer.zero_grad(); trainer._set_seed()
    s = trainer._prepare_sample(sample)
    loss, ss, log = crit(model, s)
    loss.register_hook(lambda g: pr.enable())
    t0 = time.perf_counter()
    with K.deferred_reductions(True):
        loss.backward()
    tb += time.perf_counter() - t0
    torch.cuda.synchronize()
h.remove()
print("backward enqueue: %.1f ms per update (host, profiled)" % (1e3 * tb / N))
for key in ("cumulative", "tottime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45)
    print("==== by %s (totals over %d backward passes) ====" % (key, N))
    for line in buf.getvalue().splitlines():
        if "chimera-st_amd" in line or "built-in" in line or "method" in line or "ncalls" in line:
            print(line[:180])
