"""GPU tests of the decode path (bit-exact token ids vs the reference's SequenceGenerator fixtures) and of the fused
optimizer / trainer step (vs two trainer-equivalent reference updates)."""
from argparse import Namespace
from importlib import import_module

import numpy as np
import pytest
import torch

from conftest import golden_sample, load_golden
from test_model_gpu import assert_close, build_from_golden, to_cuda

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("beam", [1, 5])
def test_decode_matches_reference_generator(beam):
    g = load_golden("decode_tiny.npz")
    model, task, args = build_from_golden(g, "chimera", torch.float32)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    gen = SG([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=12, min_len=1)
    sample = to_cuda(golden_sample(g))
    hyps = gen.generate([model], sample)
    for b in range(len(hyps)):
        for r in range(min(beam, 3)):
            key = "gen/beam%d/b%d/r%d/" % (beam, b, r)
            assert hyps[b][r]["tokens"].tolist() == g[key + "tokens"].tolist(), key  # bit-exact token ids
            assert abs(float(hyps[b][r]["score"]) - float(g[key + "score"])) < 1e-3
            assert_close(hyps[b][r]["positional_scores"], g[key + "pos_scores"], 1e-3, key + "pos_scores")


RECIPE_SETTINGS = {  # tools/ref_harness/make_decode_recipe_goldens.py SETTINGS
    "recipe": dict(beam_size=10, len_penalty=1.5),                                   # chimera/generate/generate-mustc-final.sh:5-8
    "recipe_unk": dict(beam_size=10, len_penalty=1.5, unk_penalty=0.5, min_len=4),
    "short": dict(beam_size=5, len_penalty=0.6),
    "nonorm": dict(beam_size=4, len_penalty=1.5, normalize_scores=False),
}


@pytest.mark.parametrize("fused", [True, False], ids=["engine", "host_loop"])
@pytest.mark.parametrize("name", sorted(RECIPE_SETTINGS))
def test_final_decoding_recipe_matches_reference_generator(name, fused):
    """`--beam 10 --lenpen 1.5` (chimera/generate/generate-mustc-final.sh:5-8) and the other branches of finalize_hypos' length
    normalisation, the unk penalty and min_len (sequence_generator.py:321-329, :623-624), against what the REAL reference's
    SequenceGenerator produced on the same inputs (decode_recipe_tiny.npz): every finalized hypothesis of every sentence, in the
    reference's order — token ids exact, scores to 1e-4 — from the device-resident engine AND the host loop."""
    g = load_golden("decode_tiny.npz")
    r = load_golden("decode_recipe_tiny.npz")
    model, task, args = build_from_golden(g, "chimera", torch.float32)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    kw = dict(RECIPE_SETTINGS[name])
    kw.setdefault("min_len", 1)
    gen = SG([model], task.target_dictionary, max_len_a=0, max_len_b=int(r["meta/max_len_b"]), fused=fused, **kw)
    for tag in ("a", "b"):
        sample = {"net_input": {"src_tokens": torch.from_numpy(r["in/%s/src_tokens" % tag]).cuda(),
                                "src_lengths": torch.from_numpy(r["in/%s/src_lengths" % tag]).cuda()}}
        hyps = gen.generate([model], sample)
        for b in range(len(hyps)):
            n = int(r["gen/%s/%s/b%d/n" % (name, tag, b)])
            assert len(hyps[b]) == n, (tag, b)
            for k in range(n):
                key = "gen/%s/%s/b%d/r%d/" % (name, tag, b, k)
                assert hyps[b][k]["tokens"].tolist() == r[key + "tokens"].tolist(), key
                assert abs(float(hyps[b][k]["score"]) - float(r[key + "score"])) < 1e-4, key
                assert_close(hyps[b][k]["positional_scores"], r[key + "pos_scores"], 1e-3, key + "pos_scores")


def test_decode_text_input_and_incremental_equals_full():
    g = load_golden("decode_tiny.npz")
    model, task, args = build_from_golden(g, "chimera", torch.float32)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    gen = SG([model], task.target_dictionary, beam_size=1, max_len_a=0, max_len_b=12)
    s = to_cuda(golden_sample(g))
    txt = {"net_input": {"src_tokens": s["src_text"], "src_lengths": s["src_text_lengths"]}}
    hyps = gen.generate([model], txt)
    for b in range(len(hyps)):
        assert hyps[b][0]["tokens"].tolist() == g["gen/text_beam1/b%d/r0/tokens" % b].tolist()
    # incremental-state decoder == full decoder on the same prefix (1e-6 in the reference)
    model.eval()
    with torch.no_grad():
        enc = model.encoder(s["net_input"]["src_tokens"], s["net_input"]["src_lengths"])
        prev = s["net_input"]["prev_output_tokens"]
        full, _ = model.decoder(prev, encoder_out=enc)
        inc = {}
        outs = []
        for t in range(prev.size(1)):
            o, _ = model.decoder(prev[:, :t + 1], encoder_out=enc, incremental_state=inc)
            outs.append(o[:, -1])
        inc_logits = torch.stack(outs, 1)
    # rows after a pad token differ by construction (the full pass masks pad keys); compare non-pad prefixes
    m = prev.ne(1)
    assert float((full - inc_logits)[m].abs().max()) < 1e-4


def test_two_updates_match_reference_optimizer():
    """multiply_grads(1/sample_size) -> clip 0.05 -> Adam(wd 0.01) -> inverse_sqrt(warmup 4), twice (optim_tiny.npz)."""
    g0 = load_golden("chimera_tiny.npz")
    g = load_golden("optim_tiny.npz")
    model, task, args = build_from_golden(g0, "chimera", torch.float32)
    crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    targs = Namespace(bf16=False, lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.01, clip_norm=0.05,
                      warmup_updates=4, warmup_init_lr=1e-7, seed=1)
    tr = Trainer(targs, task, model, crit, device="cuda")
    assert tr.optimizer.get_lr() == pytest.approx(float(g["lr/0"]), rel=1e-9)
    sample = golden_sample(g0)
    for step in range(2):
        out = tr.train_step([sample])
        assert out["loss"] == pytest.approx(float(g["loss/%d" % step]), rel=1e-4)
        assert out["gnorm"] == pytest.approx(float(g["gnorm/%d" % step]), rel=1e-3)
        assert out["lr"] == pytest.approx(float(g["lr/%d" % (step + 1)]), rel=1e-9)
    for name, p in tr.get_model().named_parameters():
        assert_close(p, g["param_after/" + name], 1e-4, "param " + name)


@pytest.mark.parametrize("kind,fixture", [("chimera", "chimera_tiny.npz"), ("s2t", "s2t_w2v2_tiny.npz")])
def test_gradient_accumulation_equals_one_batch(kind, fixture):
    """update_freq = 2 on the same micro-batch twice (trainer.py:479-500: no_sync on the first, accumulate, normalise by the summed
    sample size) against one micro-batch: every gradient doubles exactly, the sample size doubles, so the update must be THE SAME BITS
    (fp32).  Exercises autograd's accumulation into the gradients the backward kernels handed over (views of the packed q | k | v
    gradient among them) and the flat-buffer gather behind it."""
    g0 = load_golden(fixture)
    crit_mod = import_module("chimera-st_amd.criterions")
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    targs = Namespace(bf16=False, lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.01, clip_norm=0.05,
                      warmup_updates=4, warmup_init_lr=1e-7, seed=1)
    results = []
    for copies in (1, 2):
        model, task, args = build_from_golden(g0, kind, torch.float32)
        crit = (crit_mod.TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1) if kind == "chimera"
                else crit_mod.LabelSmoothedCrossEntropyCriterion(task, False, 0.1))
        tr = Trainer(targs, task, model, crit, device="cuda")
        sample = golden_sample(g0)
        outs = [tr.train_step([sample] * copies) for _ in range(2)]
        results.append((outs, {n: p.detach().clone() for n, p in tr.get_model().named_parameters()}))
    (o1, p1), (o2, p2) = results
    for a, b in zip(o1, o2):
        assert b["sample_size"] == 2 * a["sample_size"]
        assert a["gnorm"] == pytest.approx(b["gnorm"], rel=1e-6)
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n


def test_bf16_trainer_runs_and_decreases_loss():
    g0 = load_golden("chimera_tiny.npz")
    model, task, args = build_from_golden(g0, "chimera", torch.float32)
    crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    targs = Namespace(bf16=True, lr=[2e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=1.0,
                      warmup_updates=1, warmup_init_lr=2e-3, seed=1)
    tr = Trainer(targs, task, model, crit, device="cuda")
    sample = golden_sample(g0)
    losses = [tr.train_step([sample])["loss"] for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0]


def test_nonfinite_gradient_skips_the_update_and_leaves_the_optimizer_untouched():
    """A batch that produces Inf / NaN gradients (trainer.py:629-646, optim/dynamic_loss_scaler.py:42-70): the update is dropped — fp32
    master weights, both Adam moments, the bf16 parameters and the update counter are bit-unchanged — the next clean batch trains on,
    and a persistent condition raises FloatingPointError."""
    g0 = load_golden("chimera_tiny.npz")
    model, task, args = build_from_golden(g0, "chimera", torch.float32)
    crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    targs = Namespace(bf16=True, lr=[2e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=1.0,
                      warmup_updates=1, warmup_init_lr=2e-3, seed=1, nonfinite_tolerance=2)
    tr = Trainer(targs, task, model, crit, device="cuda")
    sample = golden_sample(g0)
    assert np.isfinite(tr.train_step([sample])["loss"])
    opt = tr.optimizer
    before = [t.clone() for t in (opt.master, opt.exp_avg, opt.exp_avg_sq, tr.buffers.flat_param)]
    bad = {k: (dict(v) if isinstance(v, dict) else v) for k, v in sample.items()}
    wav = bad["net_input"]["src_tokens"].clone().float()
    wav[0, 100] = float("inf")
    bad["net_input"]["src_tokens"] = wav
    out = tr.train_step([bad])
    assert out is not None and out["overflow"] == 1.0 and np.isnan(out["gnorm"]) and tr.num_updates == 1 and opt.num_updates == 1
    for a, b in zip(before, (opt.master, opt.exp_avg, opt.exp_avg_sq, tr.buffers.flat_param)):
        assert torch.equal(a, b)
    assert float(tr.buffers.flat_grad.float().abs().max()) == 0.0  # the poisoned gradient is gone
    good = tr.train_step([sample])
    assert np.isfinite(good["loss"]) and np.isfinite(good["gnorm"]) and tr.num_updates == 2
    with pytest.raises(FloatingPointError):
        for _ in range(4):
            tr.train_step([bad])
    assert tr.num_updates == 2


def _resume_trainer(path, arg_overrides=None):
    CU = import_module("chimera-st_amd.checkpoint_utils")
    (model,), args, task = CU.load_model_ensemble_and_task([path], arg_overrides=arg_overrides)
    crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    args.bf16 = False
    tr = Trainer(args, task, model, crit, device="cuda")
    extra = tr.load_checkpoint(path)
    return tr, extra


def test_resume_from_reference_checkpoint_reproduces_its_next_update(tmp_path):
    """tests/golden/ref_checkpoint_tiny.pt was written by the REFERENCE's checkpoint_utils.save_state after two updates; loading it
    (model + Adam moments + update counter -> lr schedule) and running update 3 must give the reference's update 3."""
    import os
    from conftest import GOLDEN
    g0 = load_golden("chimera_tiny.npz")
    g = load_golden("ref_checkpoint_tiny_next.npz")
    tr, extra = _resume_trainer(os.path.join(GOLDEN, "ref_checkpoint_tiny.pt"))
    assert extra["train_iterator"] == {"epoch": 1, "iterations_in_epoch": 2} and tr.num_updates == 2
    sample = golden_sample(g0)
    out = tr.train_step([sample])
    assert out["loss"] == pytest.approx(float(g["loss/2"]), rel=1e-4)
    assert out["gnorm"] == pytest.approx(float(g["gnorm/2"]), rel=1e-3)
    assert out["lr"] == pytest.approx(float(g["lr/3"]), rel=1e-9)
    for name, p in tr.get_model().named_parameters():
        assert_close(p, g["param_after3/" + name], 1e-4, "param " + name)
    # a checkpoint written by THIS build has the reference's structure and resumes to the same next update
    path = str(tmp_path / "checkpoint_last.pt")
    tr.save_checkpoint(path, {"train_iterator": {"epoch": 1, "iterations_in_epoch": 3}})
    state = torch.load(path, weights_only=False)
    assert list(state.keys()) == ["cfg", "args", "model", "optimizer_history", "extra_state", "last_optimizer_state"]
    assert state["optimizer_history"][-1]["num_updates"] == 3 and list(state["model"].keys()) == list(tr.get_model().state_dict().keys())
    nxt = tr.train_step([sample])
    tr2, extra2 = _resume_trainer(path)
    assert extra2["train_iterator"]["iterations_in_epoch"] == 3 and tr2.num_updates == 3
    nxt2 = tr2.train_step([sample])
    assert nxt2["loss"] == pytest.approx(nxt["loss"], rel=1e-6) and nxt2["gnorm"] == pytest.approx(nxt["gnorm"], rel=1e-5)
    for (n, a), (_, b) in zip(tr.get_model().named_parameters(), tr2.get_model().named_parameters()):
        assert float((a - b).abs().max()) < 1e-6, n


def test_resume_from_reference_checkpoint_built_on_a_quantize_targets_wav2vec2():
    """The published wav2vec_small is a quantize_targets=True pre-training model: the Chimera checkpoints built on it carry
    encoder.wav2vec_model.quantizer.* / project_q.* and their optimizer state counts those parameters.  A checkpoint the
    REFERENCE wrote for such a model (tests/golden/ref_checkpoint_quant_tiny.pt) resumes to the reference's update 3."""
    import os
    from conftest import GOLDEN
    g0 = load_golden("chimera_quant_tiny.npz")
    g = load_golden("ref_checkpoint_quant_tiny_next.npz")
    tr, extra = _resume_trainer(os.path.join(GOLDEN, "ref_checkpoint_quant_tiny.pt"),
                                {"w2v2_model_path": os.path.join(GOLDEN, "w2v_quant_tiny.pt")})
    assert tr.num_updates == 2
    out = tr.train_step([golden_sample(g0)])
    assert out["loss"] == pytest.approx(float(g["loss/2"]), rel=1e-4)
    assert out["gnorm"] == pytest.approx(float(g["gnorm/2"]), rel=1e-3)
    assert out["lr"] == pytest.approx(float(g["lr/3"]), rel=1e-9)
    for name, p in tr.get_model().named_parameters():
        assert_close(p, g["param_after3/" + name], 1e-4, "param " + name)
