"""Size-independent properties at the BASELINE dimensions (wav2vec2-small front end + s2t_transformer_m dims, 10 000-way vocabulary,
bf16, 30 s audio) — where the CPU oracle would take minutes per utterance:
  * one update is reproducible up to the rounding noise of the few fp32 atomic reductions on the path (conv0 lag moments for the
    GroupNorm statistics, the loss sum, bias column sums) amplified by bf16 storage: loss to 1e-3, gradient to 8 % in L2 (measured 1.3e-4 and 2.3 %),
  * skipping the all-padding key tiles (cst_attn_desc.kv_len; bit-identical at kernel level, tests/test_kernels_gpu.py) stays
    inside that same noise at model level,
  * the loss decreases when the same batch is fitted (the whole path learns end to end at full size)."""
import os
import sys
from argparse import Namespace

import pytest
import torch

from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu


def _build(batch=4):
    sys.path.insert(0, ROOT)
    import bench
    load_pkg()
    args = Namespace(batch=batch, seconds=30.0, lengths="uniform", dtype="bf16", model="s2t_w2v2", dropout=0.0, layerdrop=0.0)
    torch.manual_seed(1)
    trainer, task, tasks, ns = bench.build(args, torch.device("cuda", 0))
    return trainer, bench.make_batch(tasks, task, args, 0, torch.device("cuda", 0))


def _loss_and_grads(trainer, sample):
    trainer.optimizer.zero_grad()
    trainer._set_seed()
    s = trainer._prepare_sample(sample)
    loss, ss, log = trainer.criterion(trainer.model, s)
    loss.backward()
    trainer.buffers.gather_grads()
    return float(loss), trainer.buffers.flat_grad.clone()


def test_full_size_update_properties():
    trainer, sample = _build()
    assert sum(p.numel() for p in trainer.get_model().parameters()) > 150e6
    l1, g1 = _loss_and_grads(trainer, sample)
    l2, g2 = _loss_and_grads(trainer, sample)
    assert torch.isfinite(g1.float()).all() and float(g1.float().norm()) > 0

    def close(la, ga, lb, gb):
        rel = float((ga.float() - gb.float()).norm()) / float(ga.float().norm())
        print("loss %.4f vs %.4f, gradient relative L2 difference %.3e" % (la, lb, rel))
        return abs(la - lb) <= 1e-3 * abs(la) and rel <= 8e-2  # bf16 storage: the fp32-vs-bf16 gradient gap of this model is 1.8e-2

    assert close(l1, g1, l2, g2), (l1, l2)
    os.environ["CST_ATTN_NO_KVLEN"] = "1"
    try:
        l3, g3 = _loss_and_grads(trainer, sample)
    finally:
        del os.environ["CST_ATTN_NO_KVLEN"]
    assert close(l1, g1, l3, g3), (l1, l3)
    # fitting the batch: 6 updates reduce the loss
    losses = [trainer.train_step([sample])["loss"] for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]
