"""At the BASELINE dimensions (wav2vec2-small front end + s2t_transformer_m dims, 10 000-way vocabulary):
  * size-independent properties at 30 s audio in bf16: one update is BIT-reproducible (no floating-point atomics anywhere on the
    path: the same loss bits and the same gradient bits run to run), skipping the all-padding key tiles (cst_attn_desc.kv_len)
    changes no bit either, and the loss decreases when the batch is fitted;
  * full-dimension parity against the oracle on two ragged utterances (fp32 at 1e-3; bf16 against storage-rounding emulation)."""
import os
import sys
from argparse import Namespace

import pytest
import torch

from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu


def _build(batch=4, dropout=0.0, model="s2t_w2v2"):
    sys.path.insert(0, ROOT)
    import bench
    load_pkg()
    args = Namespace(batch=batch, seconds=30.0, lengths="uniform", dtype="bf16", model=model, dropout=dropout, layerdrop=0.0)
    torch.manual_seed(1)
    trainer, task, tasks, ns = bench.build(args, torch.device("cuda", 0))
    return trainer, bench.make_batch(tasks, task, args, 0, torch.device("cuda", 0))


def _loss_and_grads(trainer, sample, overlap=False):
    """overlap: the trainer's backward context (deferred reductions + the small layers' weight gradients on the side stream)."""
    from importlib import import_module
    K = import_module("chimera-st_amd.kernels")
    trainer.optimizer.zero_grad()
    trainer._set_seed()
    s = trainer._prepare_sample(sample)
    loss, ss, log = trainer.criterion(trainer.model, s)
    with K.deferred_reductions(overlap):
        loss.backward()
    trainer.buffers.gather_grads()
    return float(loss), trainer.buffers.flat_grad.clone()


def test_the_bench_batch_itself_is_bit_reproducible_and_its_loss_is_the_one_bench_prints():
    """The batch bench.py times — `bench.make_batch` at B = 32 x <= 30 s uniform lengths (31 512 packed wav2vec2 rows, 32 sequences), the
    training recipe's dropout 0.1 — is the one shape no other test executes.  Properties that do not need the oracle at this size:
    loss and every gradient bit-equal run to run (counter-based dropout masks, fixed-order reductions) and with the exact key-tile /
    dead-tile skipping switched off; the padded (CST_NO_PACK) stack gives the same loss to bf16 rounding of a different row layout;
    and the first update's loss is bit-equal to what `bench.py --steps 1 --warmup 0` prints on its line for that update."""
    import json
    import subprocess
    trainer, sample = _build(batch=32, dropout=0.1)
    assert sample["net_input"]["src_tokens"].shape[0] == 32
    l1, g1 = _loss_and_grads(trainer, sample, overlap=True)
    l2, g2 = _loss_and_grads(trainer, sample, overlap=True)
    assert torch.isfinite(g1.float()).all() and float(g1.float().norm()) > 0
    assert l1 == l2 and torch.equal(g1, g2), "B = 32 update differs run to run: %r vs %r, %d elements" % (l1, l2, int((g1 != g2).sum()))
    os.environ["CST_ATTN_NO_KVLEN"] = "1"
    os.environ["CST_GEMM_NO_KLIVE"] = "1"
    try:
        l3, g3 = _loss_and_grads(trainer, sample, overlap=True)
    finally:
        del os.environ["CST_ATTN_NO_KVLEN"], os.environ["CST_GEMM_NO_KLIVE"]
    assert l3 == l1, "switching the tile skipping off changed the loss: %r vs %r" % (l3, l1)
    # (m_live / k_live change the split-K cuts of the weight gradients: same values to fp32 summation order, not the same bits)
    rel = float((g3.float() - g1.float()).norm() / g1.float().norm())
    assert rel < 2e-3, rel
    out = trainer.train_step([sample])  # the update bench.py times: same seed (args.seed + num_updates = 1), same batch
    assert out is not None and abs(float(out["loss"]) - l1) <= 1e-6 * abs(l1), (out["loss"], l1)
    del trainer, sample, g1, g2, g3
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline",
                        "--no-extra"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["batch_per_gpu"] == 32 and d["config"]["loss"] == float(out["loss"]), (d["config"]["loss"], float(out["loss"]))


def test_the_chimera_bench_batch_is_bit_reproducible_and_its_loss_is_the_one_bench_prints():
    """The Chimera leg of the bench line (`extra.chimera`: BASELINE configs[3], B = 32 x <= 30 s, triplet_st_mt_contrastive, dropout
    0.1) at the size it is timed: the packed triplet path (criterions.py: audio and text pass walk the shared encoder / memory /
    decoder layers ONCE, > 31 k packed wav2vec2 rows + the text rows) has no other test at this size.  Run to run every gradient bit
    is equal; the unpaired, padded route (CST_NO_PACK=1: two separate passes over padded batches — different row layout, different
    dropout sites) lands at the same loss to bf16 rounding with dropout off; and the first update's loss is the one
    `bench.py --model chimera --steps 1 --warmup 0` prints."""
    import json
    import subprocess
    trainer, sample = _build(batch=32, dropout=0.1, model="chimera")
    assert sample["net_input"]["src_tokens"].shape[0] == 32 and "src_text" in sample
    l1, g1 = _loss_and_grads(trainer, sample, overlap=True)
    l2, g2 = _loss_and_grads(trainer, sample, overlap=True)
    assert torch.isfinite(g1.float()).all() and float(g1.float().norm()) > 0
    assert l1 == l2 and torch.equal(g1, g2), "B = 32 Chimera update differs run to run: %r vs %r, %d elements" % (l1, l2, int((g1 != g2).sum()))
    out = trainer.train_step([sample])
    assert out is not None and abs(float(out["loss"]) - l1) <= 1e-6 * abs(l1), (out["loss"], l1)
    for k in ("st_loss", "mt_loss", "contrastive_loss"):
        assert out[k] == out[k] and out[k] > 0, (k, out[k])
    del trainer, sample, g1, g2
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "chimera", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-roofline", "--no-extra"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["batch_per_gpu"] == 32 and d["config"]["loss"] == float(out["loss"]), (d["config"]["loss"], float(out["loss"]))
    assert "Chimera" in d["config"]["workload"]
    # dropout off (the masks of the two routes are different sites): packed pair route vs two padded passes, the same loss to bf16
    # rounding of a different row layout
    trainer, sample = _build(batch=32, dropout=0.0, model="chimera")
    lp, gp = _loss_and_grads(trainer, sample, overlap=True)
    os.environ["CST_NO_PACK"] = "1"
    try:
        # the unpaired route sends two gradients to every shared parameter: which parameters may take the deferred-reduction route
        # is a property of the route (trainer._defer_ok re-reads the switches and re-marks the shared ones); launch-each here
        trainer.optimizer.defer_reductions = trainer._defer_ok()
        lu, gu = _loss_and_grads(trainer, sample, overlap=False)
    finally:
        del os.environ["CST_NO_PACK"]
        trainer.optimizer.defer_reductions = trainer._defer_ok()
    assert torch.isfinite(gu.float()).all()
    assert abs(lp - lu) <= 2e-3 * abs(lu), "packed %.5f vs padded %.5f" % (lp, lu)
    rel = float((gp.float() - gu.float()).norm() / gu.float().norm())
    assert rel < 5e-2, rel  # two bf16 roundings of every activation apart (the bf16-vs-oracle gradient distance is 1.8e-2)


def test_full_size_update_properties():
    trainer, sample = _build()
    assert sum(p.numel() for p in trainer.get_model().parameters()) > 150e6
    l1, g1 = _loss_and_grads(trainer, sample)
    l2, g2 = _loss_and_grads(trainer, sample)
    assert torch.isfinite(g1.float()).all() and float(g1.float().norm()) > 0
    # bit-reproducible: every reduction on the path has a fixed order (block partials + fixed-order second stage; no atomics)
    assert l1 == l2, "loss differs run to run: %r vs %r" % (l1, l2)
    assert torch.equal(g1, g2), "gradient differs run to run: %d of %d elements" % (int((g1 != g2).sum()), g1.numel())
    os.environ["CST_ATTN_NO_KVLEN"] = "1"
    try:
        l3, g3 = _loss_and_grads(trainer, sample)
    finally:
        del os.environ["CST_ATTN_NO_KVLEN"]
    assert l1 == l3 and torch.equal(g1, g3), "walking the all-padding key tiles changed the result"
    # the trainer's backward context — second stages of the small reductions deferred to one launch, the small layers' weight
    # gradients on a side stream — is bit-reproducible too (a race between the streams would show here), every gradient is written
    # (the suite poisons deferred destinations with NaN) and has the bits of the launch-each route (cst_reduce_multi keeps its order)
    from importlib import import_module
    K = import_module("chimera-st_amd.kernels")
    assert trainer.optimizer.defer_reductions
    K.STATS.clear()
    n0 = K.DEFER.flushes
    l4, g4 = _loss_and_grads(trainer, sample, overlap=True)
    assert K.DEFER.flushes > n0
    l5, g5 = _loss_and_grads(trainer, sample, overlap=True)
    assert l4 == l1 and l5 == l1 and torch.isfinite(g4.float()).all() and torch.equal(g4, g5)
    assert torch.equal(g4, g1), "deferred gradients differ from the launch-each route on %d elements" % int((g4 != g1).sum())
    # ... and with the small layers' weight-gradient GEMMs on the side stream (CST_SIDE_STREAM=1: a switch, off by default — it
    # measured no faster): still the same bits, run to run and against the single-stream route
    os.environ["CST_SIDE_STREAM"] = "1"
    try:
        l6, g6 = _loss_and_grads(trainer, sample, overlap=True)
        l7, g7 = _loss_and_grads(trainer, sample, overlap=True)
    finally:
        del os.environ["CST_SIDE_STREAM"]
    assert K.STATS.get("side_gemm", 0) > 50, dict(K.STATS)
    assert l6 == l1 and torch.equal(g6, g7) and torch.equal(g6, g1)
    # fitting the batch: 6 updates reduce the loss
    losses = [trainer.train_step([sample])["loss"] for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]


# ---------------------------------------------------------------------------------------------------------------------
# Full-dimension parity against the oracle (CPU fp32 restatement pinned to the reference): the BASELINE model dimensions
# (wav2vec2-small front end: 7-layer CNN + 12 x 768 / 12 heads / hd 64; d 512 / 8 heads / ffn 2048; V = 10 000; M = 64), two
# ragged utterances (10 s and 6 s) so that every exact-skipping path of the HIP kernels is live (padded key tiles, dead query
# tiles, dead token blocks of the weight-gradient GEMMs, the CNN frame bound), the SAME parameters on both sides.
# ---------------------------------------------------------------------------------------------------------------------
def _build_full(model, dtype, samples=(160000, 96000)):
    sys.path.insert(0, ROOT)
    import bench
    load_pkg()
    args = Namespace(batch=2, seconds=max(samples) / 16000.0, lengths="uniform", dtype=dtype, model=model, dropout=0.0, layerdrop=0.0)
    trainer, task, tasks, ns = bench.build(args, torch.device("cuda", 0))
    sample = tasks.synthetic_sample(task.target_dictionary, 2, list(samples), [37, 21], [20, 33], seed=3)
    return trainer, task, ns, sample, bench.oracle_cfg(ns)


def _hip_forward_backward(trainer, sample, chimera):
    from importlib import import_module
    K = import_module("chimera-st_amd.kernels")
    model = trainer.get_model()
    model.train()
    s = trainer._prepare_sample(sample)
    with torch.no_grad():
        if chimera:
            (logits, _), memory = model.forward_with_internal(**s["net_input"])
        else:
            (logits, _), memory = model(**s["net_input"]), None
    trainer.optimizer.zero_grad()
    K.STATS.clear()
    loss, sample_size, log = trainer.criterion(model, s)
    loss.backward()
    stats = dict(K.STATS)
    grads = {n: (p.grad.detach().float().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}
    return float(loss), log, logits.float().cpu(), (memory.float().cpu() if memory is not None else None), grads, stats


@pytest.mark.parametrize("model,samples", [("s2t_w2v2", (160000, 96000)), ("chimera", (160000, 96000)), ("s2t_w2v2", (480000, 272000)),
                                           ("chimera", (480000, 272000))],
                         ids=["s2t_w2v2", "chimera", "s2t_w2v2-30s", "chimera-30s"])
def test_full_dimension_fp32_parity_with_oracle(model, samples):
    """fp32 storage: loss <= 1e-4 relative, logits / memory / EVERY parameter gradient <= 1e-3 * max(1, |ref|max)
    (BASELINE north_star: 'logits/grads within 1e-3'), no exemptions.  The random parameters are first moved off the ReLU
    kink (parity_util.detie: a pre-activation within rounding of 0 makes the gradient itself ill-defined).
    The 30 s + 17 s case is the shape bench.py runs: T1 = 1499 wav2vec2 frames (24 key tiles per sequence, packed row offsets
    beyond 2 k, the positional convolution and the subsampler at their full reach)."""
    from oracle import chimera_oracle as O
    from parity_util import assert_grads_close_fp32, cpu_sample, max_abs_rel, run_oracle
    from parity_util import detie
    chimera = model == "chimera"
    trainer, task, ns, sample, cfg = _build_full(model, "f32", samples)
    fn = O.triplet_criterion if chimera else O.lsce_criterion
    sd, moved = detie(fn, {k: v.detach().cpu() for k, v in trainer.get_model().state_dict().items()}, cpu_sample(sample), cfg)
    trainer.get_model().load_state_dict(sd)  # the same tensors on both sides; no ReLU pre-activation within 1e-4 of the kink
    loss, log, logits, memory, grads, stats = _hip_forward_backward(trainer, sample, chimera)
    # every exact-skipping path must have been live in this run
    assert stats.get("attn_kv_len", 0) > 0 and stats.get("attn_q_flags", 0) > 0 and stats.get("gemm_k_len", 0) > 0, stats
    if not chimera:  # the Chimera memory attends every padded frame (quirk Q1): no dead token blocks inside wav2vec2 there
        assert stats.get("gemm_k_live", 0) > 0 and stats.get("gemm_m_live", 0) > 0, stats
    ref, rgrads = run_oracle(fn, sd, cpu_sample(sample), cfg)
    rl = float(ref["loss"])
    assert abs(loss - rl) <= 1e-4 * abs(rl), "loss %.6f vs oracle %.6f" % (loss, rl)
    if chimera:
        for k in ("st_loss", "mt_loss", "contrastive_loss"):
            assert abs(float(log[k]) - float(ref[k])) <= 1e-4 * abs(float(ref[k])) + 1e-3, k
        assert max_abs_rel(memory, ref["memory_audio"]) <= 1e-3
    e = max_abs_rel(logits, ref["st_logits"] if chimera else ref["logits"])
    assert e <= 1e-3, "logits: %.3e" % e
    n, worst, excused = assert_grads_close_fp32({k: grads.get(k) for k in rgrads if k in grads}, {k: v for k, v in rgrads.items() if k in grads},
                                                ref["relu_min_abs"])
    print("%s fp32 full-dim: loss %.4f (oracle %.4f), %d gradients, worst %s %.2e, %d fc1 biases moved off the ReLU kink, skip stats %s"
          % (model, loss, rl, n, worst[0], worst[1], moved, stats))
    assert n > 300 and excused == 0  # every gradient entry within 1e-3, no exemption used


@pytest.mark.parametrize("model,samples", [("s2t_w2v2", (160000, 96000)), ("chimera", (160000, 96000)), ("s2t_w2v2", (480000, 272000)),
                                           ("chimera", (480000, 272000))],
                         ids=["s2t_w2v2", "chimera", "s2t_w2v2-30s", "chimera-30s"])
def test_full_dimension_bf16_gap_is_storage_rounding(model, samples):
    """bf16 storage (what bench.py measures).  The 30 s + 17 s case puts the kernels the bench times — the DMA-staged attention
    kernels (bf16 only: 24 key tiles per sequence, packed row offsets) and the persistent 256 x 256 GEMM — under the oracle at the
    bench's sequence length.  bf16 keeps 8 mantissa bits, so the distance to the fp32 oracle is not 1e-3; what
    must hold is that the distance is ROUNDING and nothing else: the oracle re-run with every stored tensor rounded to bf16 at
    the storage points of the HIP path (oracle.STORAGE) lands at some distance from the fp32 oracle, and the HIP path must not
    be farther away than 1.5x that — globally, and for every tensor that carries >= 1 % of the gradient norm (2x + 1e-2)."""
    from oracle import chimera_oracle as O
    from parity_util import cpu_sample, grad_errors, max_abs_rel, run_oracle
    chimera = model == "chimera"
    trainer, task, ns, sample, cfg = _build_full(model, "bf16", samples)
    loss, log, logits, memory, grads, stats = _hip_forward_backward(trainer, sample, chimera)
    assert stats.get("attn_fast", 0) > 0, stats  # the DMA-staged attention kernels ran (not the generic fp32-capable ones)
    sd = {k: v.detach().cpu() for k, v in trainer.get_model().state_dict().items()}  # bf16 values: exactly representable in fp32
    fn = O.triplet_criterion if chimera else O.lsce_criterion
    cs = cpu_sample(sample)
    ref, rgrads = run_oracle(fn, sd, cs, cfg)
    emu, egrads = run_oracle(fn, sd, cs, cfg, storage=torch.bfloat16)
    key = "st_logits" if chimera else "logits"
    rl, el = float(ref["loss"]), float(emu["loss"])
    emu_loss_gap = abs(el - rl) / abs(rl)
    assert abs(loss - rl) / abs(rl) <= max(2 * emu_loss_gap, 5e-4), "loss %.5f, oracle %.5f, emulated %.5f" % (loss, rl, el)
    e_logit_emu = max_abs_rel(emu[key], ref[key])
    e_logit = max_abs_rel(logits, ref[key])
    assert e_logit <= 2 * e_logit_emu + 1e-3, "logits: hip %.3e vs emulated rounding %.3e" % (e_logit, e_logit_emu)
    g_emu, per_emu = grad_errors(egrads, rgrads)
    g_hip, per_hip = grad_errors({k: grads.get(k) for k in rgrads}, rgrads)
    print("%s bf16 full-dim: loss hip %.4f / oracle %.4f / emulated %.4f; logits err hip %.3e / emulated %.3e; "
          "gradient rel-L2 hip %.3e / emulated %.3e" % (model, loss, rl, el, e_logit, e_logit_emu, g_hip, g_emu))
    assert g_hip <= 1.5 * g_emu + 1e-3, "global gradient error %.3e vs %.3e from storage rounding alone" % (g_hip, g_emu)
    for k, (e, share) in per_hip.items():
        if share >= 1e-2:
            assert e <= 2 * per_emu[k][0] + 1e-2, "grad %s: %.3e vs emulated %.3e (share %.3f)" % (k, e, per_emu[k][0], share)
