#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by running the REAL reference
(/root/reference, imported through ref_import.py) on CPU fp32.

Run in the build container only:   python tools/ref_harness/make_goldens.py
The fixtures hold data (inputs, parameters, per-stage activations, logits, loss terms, grads,
decode results) — never reference source.  RNG-free settings: dropout 0, layerdrop 0.

Fixtures
  chimera_tiny.npz   s2t_transformer_w2v2_interlingua + triplet_st_mt_contrastive (config 4, tiny dims)
  s2t_w2v2_tiny.npz  s2t_transformer_w2v2 + label_smoothed_cross_entropy          (config 2/3, tiny dims)
  decode_tiny.npz    the chimera model after fitting the sample: greedy / beam-5 hypotheses from SequenceGenerator
  optim_tiny.npz     two trainer-equivalent updates on the chimera model (multiply_grads, clip, Adam, inverse_sqrt)
"""
import argparse
import math
import os
import sys
import tempfile
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")

W2V_TINY = dict(
    extractor_mode="default",
    conv_feature_layers="[(32, 10, 5)] + [(32, 3, 2)] * 2 + [(32, 2, 2)]",
    conv_bias=False,
    encoder_layers=2,
    encoder_embed_dim=64,
    encoder_ffn_embed_dim=128,
    encoder_attention_heads=2,
    activation_fn="gelu",
    dropout=0.0,
    attention_dropout=0.0,
    activation_dropout=0.0,
    encoder_layerdrop=0.0,
    dropout_input=0.0,
    dropout_features=0.0,
    layer_norm_first=False,
    feature_grad_mult=0.1,
    conv_pos=16,
    conv_pos_groups=4,
    final_dim=0,
    quantize_targets=False,
    quantize_input=False,
    same_quantizer=False,
    latent_vars=8,
    latent_groups=2,
    latent_dim=0,
    latent_temp="(2,0.5,0.999995)",
    logit_temp=0.1,
    mask_length=10,
    mask_prob=0.65,
    mask_selection="static",
    mask_other=0,
    no_mask_overlap=False,
    mask_min_space=1,
    mask_channel_length=10,
    mask_channel_prob=0,
    mask_channel_selection="static",
    mask_channel_other=0,
    no_mask_channel_overlap=False,
    mask_channel_min_space=1,
    num_negatives=4,
    negatives_from_everywhere=False,
    cross_sample_negatives=0,
    codebook_negatives=0,
    target_glu=False,
)

VOCAB = 60  # incl. 4 specials (bos=0 pad=1 eos=2 unk=3)


def make_dictionary():
    from fairseq.data import Dictionary

    d = Dictionary()
    for i in range(VOCAB - 4):
        d.add_symbol("w%d" % i)
    assert len(d) == VOCAB and d.pad() == 1 and d.eos() == 2 and d.unk() == 3 and d.bos() == 0
    return d


class TaskStub:
    def __init__(self, d):
        self.source_dictionary = d
        self.target_dictionary = d


def build_w2v_ckpt(path, seed):
    from fairseq.models.wav2vec.wav2vec2 import Wav2Vec2Model

    torch.manual_seed(seed)
    ns = Namespace(**W2V_TINY)
    w2v = Wav2Vec2Model.build_model(ns, task=None)
    # give the (zero-init) pos-conv bias / GroupNorm affine non-trivial values so parity sees them
    with torch.no_grad():
        for n, p in w2v.named_parameters():
            if n.endswith("bias") or "layer_norm.weight" in n or n.endswith("2.weight"):
                p.add_(0.1 * torch.randn_like(p))
    torch.save({"args": ns, "model": w2v.state_dict()}, path)
    return ns


def model_args(w2v_path, **over):
    ns = Namespace(
        w2v2_model_path=w2v_path,
        use_asr_finetune_w2v=False,
        conv_kernel_sizes="5,5",
        conv_channels=64,
        encoder_embed_dim=64,
        encoder_ffn_embed_dim=128,
        encoder_layers=2,
        encoder_attention_heads=2,
        decoder_attention_heads=2,
        decoder_layers=2,
        dropout=0.0,
        attention_dropout=0.0,
        activation_dropout=0.0,
        share_decoder_input_output_embed=True,
        max_source_positions=2000000,
        max_target_positions=1024,
        tie_adaptive_weights=False,
        quant_noise_pq_block_size=8,
        interlingua_length=8,
        interlingua_layers=2,
        interlingua_debug_options=[],
    )
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def randomize_small_params(model, seed):
    """LayerNorm affine / biases are 1/0 at init; perturb so the fixtures exercise them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("encoder.wav2vec_model"):
                continue
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def make_sample(d, seed, B=2, S=(4000, 3200), U=(7, 5), L=(6, 9)):
    """Mirror of TripletDataset.collater output (data/audio/triplet_dataset.py:165-235):
    sorted by audio length descending, right-padded."""
    from fairseq.data import data_utils

    g = torch.Generator().manual_seed(seed)
    Smax = max(S)
    audio = torch.zeros(B, Smax)
    for i, s in enumerate(S):
        audio[i, :s] = 0.1 * torch.randn(s, generator=g)
    tgt = [torch.cat([torch.randint(4, VOCAB, (u,), generator=g), torch.tensor([d.eos()])]) for u in U]
    src = [torch.cat([torch.randint(4, VOCAB, (l,), generator=g), torch.tensor([d.eos()])]) for l in L]
    target = data_utils.collate_tokens(tgt, d.pad(), d.eos(), left_pad=False, move_eos_to_beginning=False)
    prev = data_utils.collate_tokens(tgt, d.pad(), d.eos(), left_pad=False, move_eos_to_beginning=True)
    src_text = data_utils.collate_tokens(src, d.pad(), d.eos(), left_pad=False, move_eos_to_beginning=False)
    sample = {
        "id": torch.arange(B),
        "net_input": {
            "src_tokens": audio,
            "src_lengths": torch.tensor(S, dtype=torch.long),
            "prev_output_tokens": prev,
            "mask": False,
        },
        "target": target,
        "target_lengths": torch.tensor([len(t) for t in tgt]),
        "src_text": src_text,
        "src_text_lengths": torch.tensor([len(s) for s in src]),
        "ntokens": int(sum(len(t) for t in tgt)),
        "nsentences": B,
    }
    return sample


def np_state(model):
    return {"param/" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def capture(model, names):
    acts, hooks = {}, []
    mods = dict(model.named_modules())
    for tag, mname in names.items():

        def hook(m, i, o, tag=tag):
            if isinstance(o, tuple):
                o = o[0]
            if isinstance(o, dict):
                o = o["x"]
            acts.setdefault(tag, []).append(o.detach().clone().cpu().numpy())

        hooks.append(mods[mname].register_forward_hook(hook))
    return acts, hooks


def gen_chimera(tmp):
    from fairseq.criterions.triplet_st_mt_contrastive import TripletSTMTContrastiveCriterion
    from fairseq.models.chimera.w2v2_transformer_interlingua import S2TTransformerInterlinguaModelW2V2
    from fairseq.sequence_generator import SequenceGenerator

    d = make_dictionary()
    task = TaskStub(d)
    w2v_path = os.path.join(tmp, "w2v_tiny.pt")
    build_w2v_ckpt(w2v_path, seed=11)
    torch.manual_seed(12)
    args = model_args(w2v_path)
    model = S2TTransformerInterlinguaModelW2V2.build_model(args, task)
    randomize_small_params(model, 13)
    # CPU conv backward needs a contiguous input here; values unchanged.
    model.encoder.wav2vec_model.encoder.pos_conv.register_forward_pre_hook(lambda m, i: (i[0].contiguous(),))
    model.train()  # dropout is 0 everywhere: train == eval numerically
    crit = TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1, 0, False, None, [None, None])
    sample = make_sample(d, seed=14)

    names = {
        "w2v_cnn": "encoder.wav2vec_model.feature_extractor",
        "w2v_ln": "encoder.wav2vec_model.layer_norm",
        "w2v_proj": "encoder.wav2vec_model.post_extract_proj",
        "w2v_out": "encoder.wav2vec_model.encoder",
        "subsample": "encoder.subsample",
        "enc_layer_last": "encoder.transformer_layers.%d" % (args.encoder_layers - 1),
        "enc_ln": "encoder.layer_norm",
        "mem_layer0": "encoder.interlingua_layers.0",
        "dec_features_ln": "decoder.layer_norm",
    }
    acts, hooks = capture(model, names)
    out = {}
    model.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    for h in hooks:
        h.remove()
    # per-stage activations: index 0 = audio pass, 1 = text pass (only stages both passes reach)
    for tag, v in acts.items():
        out["act/" + tag + "/audio"] = v[0]
        if len(v) > 1 and tag in ("enc_layer_last", "enc_ln", "mem_layer0", "dec_features_ln"):
            out["act/" + tag + "/text"] = v[1]
    with torch.no_grad():
        (st_logits, _), mem_a = model.forward_with_internal(**sample["net_input"])
        (mt_logits, _), mem_t = model.forward_with_internal(
            src_tokens=sample["src_text"], src_lengths=sample["src_text_lengths"],
            prev_output_tokens=sample["net_input"]["prev_output_tokens"], mask=False)
    out.update({
        "out/st_logits": st_logits.numpy(), "out/mt_logits": mt_logits.numpy(),
        "out/memory_audio": mem_a.numpy(), "out/memory_text": mem_t.numpy(),
        "loss/loss": np.float64(loss.item()), "loss/sample_size": np.int64(sample_size),
    })
    for k in ("nll_loss", "st_loss", "st_nll_loss", "mt_loss", "mt_nll_loss", "contrastive_loss"):
        out["loss/" + k] = np.float64(float(log[k]))
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    out.update(np_state(model))
    for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
        out["in/" + k] = sample["net_input"][k].numpy()
    for k in ("target", "target_lengths", "src_text", "src_text_lengths"):
        out["in/" + k] = sample[k].numpy()
    out["in/ntokens"] = np.int64(sample["ntokens"])

    out["meta/w2v_args"] = np.array(repr(W2V_TINY))
    out["meta/model_args"] = np.array(repr({k: v for k, v in vars(args).items() if k != "w2v2_model_path"}))
    np.savez_compressed(os.path.join(OUT, "chimera_tiny.npz"), **out)
    print("chimera_tiny: loss", loss.item(), {k: float(v) for k, v in log.items() if "loss" in k})

    # ---- one trainer-equivalent update (trainer.py:601-627; optim/adam.py:146-226;
    #      lr_scheduler/inverse_square_root_schedule.py:52-94; utils.py:323-364) -----------------
    from fairseq.optim.adam import FairseqAdam
    from fairseq.optim.lr_scheduler.inverse_square_root_schedule import InverseSquareRootSchedule

    model.train()
    oargs = Namespace(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.01,
                      use_old_adam=True, warmup_updates=4, warmup_init_lr=1e-7, tpu=False, fp16_adam_stats=False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FairseqAdam(oargs, params)
    sched = InverseSquareRootSchedule(oargs, opt)
    sched.step_update(0)
    o2 = {"lr/0": np.float64(opt.get_lr())}
    for step in range(2):
        opt.zero_grad()
        loss, sample_size, log = crit(model, sample)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(sample_size))  # world_size 1
        gnorm = opt.clip_grad_norm(0.05)  # low threshold so clipping is exercised
        opt.step()
        sched.step_update(step + 1)
        o2["gnorm/%d" % step] = np.float64(float(gnorm))
        o2["loss/%d" % step] = np.float64(loss.item())
        o2["lr/%d" % (step + 1)] = np.float64(opt.get_lr())
    for n, p in model.named_parameters():
        o2["param_after/" + n] = p.detach().numpy().copy()
    o2["meta/optim_args"] = np.array(repr(vars(oargs)))
    np.savez_compressed(os.path.join(OUT, "optim_tiny.npz"), **o2)
    print("optim_tiny: gnorm", o2["gnorm/0"], o2["gnorm/1"], "lr", o2["lr/1"], o2["lr/2"])

    # ---- decode fixture: a randomly initialised tied-embedding model decodes degenerate repeats
    #      (SURVEY §8c), so first fit the tiny model to the sample with the reference's own
    #      criterion + Adam, then record greedy / beam-5 results from the reference generator. ----
    oargs2 = Namespace(lr=[4e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0,
                       use_old_adam=True, tpu=False, fp16_adam_stats=False)
    opt2 = FairseqAdam(oargs2, params)
    for step in range(120):
        opt2.zero_grad()
        loss, sample_size, log = crit(model, sample)
        opt2.backward(loss)
        opt2.multiply_grads(1.0 / float(sample_size))
        opt2.clip_grad_norm(1.0)
        opt2.step()
    print("decode_tiny: loss after fit", loss.item())
    out = {}
    # decode goldens (eval mode), audio input: beam 1 (greedy) and beam 5; per-step logits of the
    # full (non-incremental) decoder are already pinned by out/st_logits.
    model.eval()
    for beam in (1, 5):
        gen = SequenceGenerator([model], d, beam_size=beam, max_len_a=0, max_len_b=12, min_len=1)
        with torch.no_grad():
            hyps = gen.generate([model], sample)
        for b, h in enumerate(hyps):
            for r, hyp in enumerate(h[: min(beam, 3)]):
                out["gen/beam%d/b%d/r%d/tokens" % (beam, b, r)] = hyp["tokens"].numpy()
                out["gen/beam%d/b%d/r%d/score" % (beam, b, r)] = np.float64(float(hyp["score"]))
                out["gen/beam%d/b%d/r%d/pos_scores" % (beam, b, r)] = hyp["positional_scores"].numpy()
    # also greedy decode on the TEXT input path (MT direction)
    txt_sample = {"net_input": {"src_tokens": sample["src_text"], "src_lengths": sample["src_text_lengths"]}}
    gen = SequenceGenerator([model], d, beam_size=1, max_len_a=0, max_len_b=12, min_len=1)
    with torch.no_grad():
        hyps = gen.generate([model], txt_sample)
    for b, h in enumerate(hyps):
        out["gen/text_beam1/b%d/r0/tokens" % b] = h[0]["tokens"].numpy()
        out["gen/text_beam1/b%d/r0/score" % b] = np.float64(float(h[0]["score"]))

    with torch.no_grad():
        (st_logits, _), mem_a = model.forward_with_internal(**sample["net_input"])
    out["out/st_logits"] = st_logits.numpy()
    out["out/memory_audio"] = mem_a.numpy()
    out.update(np_state(model))
    for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
        out["in/" + k] = sample["net_input"][k].numpy()
    for k in ("target", "src_text", "src_text_lengths"):
        out["in/" + k] = sample[k].numpy()
    out["meta/w2v_args"] = np.array(repr(W2V_TINY))
    out["meta/model_args"] = np.array(repr({k: v for k, v in vars(args).items() if k != "w2v2_model_path"}))
    np.savez_compressed(os.path.join(OUT, "decode_tiny.npz"), **out)
    for k in sorted(out):
        if k.startswith("gen/") and k.endswith("tokens"):
            print(k, out[k])


def gen_s2t_w2v2(tmp):
    from fairseq.criterions.label_smoothed_cross_entropy import LabelSmoothedCrossEntropyCriterion
    from fairseq.models.chimera.w2v2_transformer import S2TTransformerModelW2V2

    d = make_dictionary()
    task = TaskStub(d)
    w2v_path = os.path.join(tmp, "w2v_tiny2.pt")
    build_w2v_ckpt(w2v_path, seed=21)
    torch.manual_seed(22)
    args = model_args(w2v_path, encoder_layers=3)
    model = S2TTransformerModelW2V2.build_model(args, task)
    randomize_small_params(model, 23)
    model.encoder.wav2vec_model.encoder.pos_conv.register_forward_pre_hook(lambda m, i: (i[0].contiguous(),))
    model.train()
    crit = LabelSmoothedCrossEntropyCriterion(task, False, 0.1)
    sample = make_sample(d, seed=24, B=3, S=(3600, 3000, 1700), U=(6, 9, 4), L=(3, 3, 3))
    model.zero_grad()
    loss, sample_size, log = crit(model, sample)  # model.forward(**net_input) swallows `mask`
    loss.backward()
    out = {}
    with torch.no_grad():
        logits, _ = model(**sample["net_input"])
        enc = model.encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
    out["out/logits"] = logits.numpy()
    out["out/encoder_out"] = enc.encoder_out.numpy()
    out["out/encoder_padding_mask"] = (
        enc.encoder_padding_mask.numpy() if enc.encoder_padding_mask is not None else np.zeros((0,), dtype=bool))
    out["loss/loss"] = np.float64(loss.item())
    out["loss/nll_loss"] = np.float64(float(log["nll_loss"]))
    out["loss/sample_size"] = np.int64(sample_size)
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    out.update(np_state(model))
    for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
        out["in/" + k] = sample["net_input"][k].numpy()
    for k in ("target", "target_lengths"):
        out["in/" + k] = sample[k].numpy()
    out["in/ntokens"] = np.int64(sample["ntokens"])
    out["meta/w2v_args"] = np.array(repr(W2V_TINY))
    out["meta/model_args"] = np.array(repr({k: v for k, v in vars(args).items() if k != "w2v2_model_path"}))
    np.savez_compressed(os.path.join(OUT, "s2t_w2v2_tiny.npz"), **out)
    print("s2t_w2v2_tiny: loss", loss.item())


def main():
    ap = argparse.ArgumentParser()
    ap.parse_args()
    import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    with tempfile.TemporaryDirectory() as tmp:
        gen_chimera(tmp)
        gen_s2t_w2v2(tmp)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
