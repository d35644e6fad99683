#!/usr/bin/env python3
"""Round-2 golden fixtures, produced by running the REAL reference (/root/reference via ref_import.py) on CPU fp32.
Run in the build container only:   python tools/ref_harness/make_goldens_r02.py

  tests/golden/s2t_fbank_tiny.npz        stock `s2t_transformer` (models/speech_to_text/s2t_transformer.py:80-366) on filter-bank
                                         input [B, T, 80] — BASELINE config 1 (`s2t_transformer_s`) and config 5
                                         (`s2t_transformer_l`) at tiny dimensions: encoder output, logits, label-smoothed CE, every
                                         gradient, one Adam update, and SequenceGenerator beam-1 / beam-5 hypotheses after fitting
  tests/golden/w2v_quant_tiny.pt         a wav2vec2 pre-training checkpoint file {"args": Namespace, "model": state_dict} as
                                         chimera/tools/download_wav2vec2.sh fetches it, quantize_targets=True (the published
                                         wav2vec_small has it): carries quantizer.* / project_q.* keys
  tests/golden/ref_checkpoint_quant_tiny.pt / _next.npz
                                         a Chimera model built ON that file by the reference (--w2v2-model-path), two Adam updates,
                                         checkpoint written by the reference's save_state, then update 3 recorded: the optimizer
                                         state index space includes the quantizer / project_q parameters
  tests/golden/chimera_quant_tiny.npz    forward / backward of that model at its initial parameters (loss terms, gradients)

The fixtures hold data only (inputs, parameters, outputs) — never reference source."""
import os
import sys
import tempfile
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import import_reference  # noqa: E402

import_reference()
import make_goldens as G  # noqa: E402

GOLD = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))


def s2t_args(**over):
    ns = Namespace(
        input_feat_per_channel=80, input_channels=1, conv_kernel_sizes="5,5", conv_channels=64,
        encoder_embed_dim=64, encoder_ffn_embed_dim=128, encoder_layers=3, encoder_attention_heads=2,
        decoder_attention_heads=2, decoder_layers=2, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        share_decoder_input_output_embed=True, max_source_positions=6000, max_target_positions=1024,
        tie_adaptive_weights=False, quant_noise_pq_block_size=8)
    for k, v in over.items():
        setattr(ns, k, v)
    return ns


def gen_s2t_fbank():
    from fairseq.criterions.label_smoothed_cross_entropy import label_smoothed_nll_loss
    from fairseq.data import data_utils
    from fairseq.models.speech_to_text.s2t_transformer import S2TTransformerModel
    from fairseq.optim.adam import FairseqAdam
    from fairseq.sequence_generator import SequenceGenerator

    d = G.make_dictionary()
    task = G.TaskStub(d)
    torch.manual_seed(31)
    args = s2t_args()
    model = S2TTransformerModel.build_model(args, task)
    G.randomize_small_params(model, 32)
    model.train()
    g = torch.Generator().manual_seed(33)
    B, T = 3, (57, 44, 23)
    feats = torch.zeros(B, max(T), 80)
    for i, t in enumerate(T):
        feats[i, :t] = torch.randn(t, 80, generator=g)
    tgt = [torch.cat([torch.randint(4, G.VOCAB, (u,), generator=g), torch.tensor([d.eos()])]) for u in (8, 5, 11)]
    target = data_utils.collate_tokens(tgt, d.pad(), d.eos(), left_pad=False, move_eos_to_beginning=False)
    prev = data_utils.collate_tokens(tgt, d.pad(), d.eos(), left_pad=False, move_eos_to_beginning=True)
    lens = torch.tensor(T, dtype=torch.long)
    ntokens = int(sum(len(t) for t in tgt))

    def loss_fn():
        # LabelSmoothedCrossEntropyCriterion.compute_loss (label_smoothed_cross_entropy.py:87-108), called without the collater's
        # `mask` kwarg: the stock model's forward has no **kwargs
        net_output = model(feats, lens, prev)
        lprobs = model.get_normalized_probs(net_output, log_probs=True)
        loss, nll = label_smoothed_nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), 0.1, ignore_index=d.pad(), reduce=True)
        return loss, nll, net_output[0]

    out = {}
    model.zero_grad()
    loss, nll, logits = loss_fn()
    loss.backward()
    with torch.no_grad():
        enc = model.encoder(feats, lens)
    out["out/logits"] = logits.detach().numpy()
    out["out/encoder_out"] = enc.encoder_out.numpy()
    out["out/encoder_padding_mask"] = enc.encoder_padding_mask.numpy() if enc.encoder_padding_mask is not None else np.zeros((0,), dtype=bool)
    out["loss/loss"], out["loss/nll_loss"], out["loss/sample_size"] = np.float64(loss.item()), np.float64(nll.item()), np.int64(ntokens)
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    out.update({k: v.copy() for k, v in G.np_state(model).items()})  # copies: the fit below updates the parameters in place
    out["in/src_tokens"], out["in/src_lengths"] = feats.numpy(), lens.numpy()
    out["in/prev_output_tokens"], out["in/target"], out["in/ntokens"] = prev.numpy(), target.numpy(), np.int64(ntokens)
    out["meta/model_args"] = np.array(repr(vars(args)))
    print("s2t_fbank_tiny: loss", loss.item())

    # fit, then decode (a random-init tied model decodes degenerate repeats): greedy + beam 5 from the reference generator
    oargs = Namespace(lr=[4e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, use_old_adam=True, tpu=False, fp16_adam_stats=False)
    opt = FairseqAdam(oargs, [p for p in model.parameters() if p.requires_grad])
    for step in range(150):
        opt.zero_grad()
        loss, _, _ = loss_fn()
        loss.backward()
        opt.multiply_grads(1.0 / ntokens)
        opt.clip_grad_norm(1.0)
        opt.step()
    print("s2t_fbank_tiny: loss after fit", loss.item())
    model.eval()
    for k, v in model.state_dict().items():
        out["fit_param/" + k] = v.detach().numpy().copy()
    sample = {"net_input": {"src_tokens": feats, "src_lengths": lens}}
    for beam in (1, 5):
        gen = SequenceGenerator([model], d, beam_size=beam, max_len_a=0, max_len_b=14, min_len=1)
        with torch.no_grad():
            hyps = gen.generate([model], sample)
        for b, h in enumerate(hyps):
            for r, hyp in enumerate(h[: min(beam, 3)]):
                out["gen/beam%d/b%d/r%d/tokens" % (beam, b, r)] = hyp["tokens"].numpy()
                out["gen/beam%d/b%d/r%d/score" % (beam, b, r)] = np.float64(float(hyp["score"]))
                print("beam", beam, b, r, hyp["tokens"].tolist(), float(hyp["score"]))
    np.savez_compressed(os.path.join(GOLD, "s2t_fbank_tiny.npz"), **out)


def gen_quant(tmp):
    from fairseq import checkpoint_utils
    from fairseq.criterions.triplet_st_mt_contrastive import TripletSTMTContrastiveCriterion
    from fairseq.models.chimera.w2v2_transformer_interlingua import S2TTransformerInterlinguaModelW2V2
    from fairseq.models.wav2vec.wav2vec2 import Wav2Vec2Model
    from fairseq.optim.adam import FairseqAdam
    from fairseq.optim.lr_scheduler.inverse_square_root_schedule import InverseSquareRootSchedule

    wargs = dict(G.W2V_TINY)
    wargs.update(quantize_targets=True, final_dim=16, latent_vars=8, latent_groups=2, latent_dim=0, feature_grad_mult=0.1)
    torch.manual_seed(41)
    ns = Namespace(**wargs)
    w2v = Wav2Vec2Model.build_model(ns, task=None)
    with torch.no_grad():
        for n, p in w2v.named_parameters():
            if n.endswith("bias") or "layer_norm.weight" in n or n.endswith("2.weight"):
                p.add_(0.1 * torch.randn_like(p))
    w2v_path = os.path.join(GOLD, "w2v_quant_tiny.pt")
    torch.save({"args": ns, "model": w2v.state_dict()}, w2v_path)
    qkeys = [k for k in w2v.state_dict() if k.split(".")[0] in ("quantizer", "project_q")]
    print("w2v_quant_tiny.pt: %d keys, pre-training-only: %s" % (len(w2v.state_dict()), qkeys))

    d = G.make_dictionary()
    task = G.TaskStub(d)
    torch.manual_seed(42)
    args = G.model_args(w2v_path)
    model = S2TTransformerInterlinguaModelW2V2.build_model(args, task)
    G.randomize_small_params(model, 43)
    model.encoder.wav2vec_model.encoder.pos_conv.register_forward_pre_hook(lambda m, i: (i[0].contiguous(),))
    model.train()
    crit = TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1, 0, False, None, [None, None])
    sample = G.make_sample(d, seed=44, B=3, S=(4200, 3000, 2100), U=(6, 8, 3), L=(5, 4, 9))
    out = {}
    model.zero_grad()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    out["loss/loss"], out["loss/sample_size"] = np.float64(loss.item()), np.int64(sample_size)
    for k in ("nll_loss", "st_loss", "mt_loss", "contrastive_loss"):
        out["loss/" + k] = np.float64(float(log[k]))
    for n, p in model.named_parameters():
        out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    out.update(G.np_state(model))
    for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
        out["in/" + k] = sample["net_input"][k].numpy()
    for k in ("target", "target_lengths", "src_text", "src_text_lengths"):
        out["in/" + k] = sample[k].numpy()
    out["in/ntokens"] = np.int64(sample["ntokens"])
    out["meta/w2v_args"] = np.array(repr(wargs))
    out["meta/model_args"] = np.array(repr({k: v for k, v in vars(args).items() if k != "w2v2_model_path"}))
    out["meta/param_names"] = np.array(repr([n for n, _ in model.named_parameters()]))
    np.savez_compressed(os.path.join(GOLD, "chimera_quant_tiny.npz"), **out)
    print("chimera_quant_tiny: loss", loss.item())

    oargs = Namespace(lr=[1e-3], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.01, use_old_adam=True, warmup_updates=4,
                      warmup_init_lr=1e-7, tpu=False, fp16_adam_stats=False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FairseqAdam(oargs, params)
    sched = InverseSquareRootSchedule(oargs, opt)
    sched.step_update(0)

    def update(step):
        opt.zero_grad()
        loss, sample_size, log = crit(model, sample)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(sample_size))
        gnorm = opt.clip_grad_norm(0.05)
        opt.step()
        sched.step_update(step + 1)
        return float(loss), float(gnorm)

    for step in range(2):
        update(step)
    ck_args = Namespace(**{k: v for k, v in vars(args).items() if k != "w2v2_model_path"})
    ck_args.w2v2_model_path = "tests/golden/w2v_quant_tiny.pt"  # resolved against the repository root by the test
    ck_args.arch, ck_args.task, ck_args.criterion = "s2t_transformer_w2v2_interlingua_base", "triplet", "triplet_st_mt_contrastive"
    ck_args.optimizer, ck_args.lr_scheduler, ck_args.no_save_optimizer_state = "adam", "inverse_sqrt", False
    for k, v in vars(oargs).items():
        setattr(ck_args, k, v)
    ck_args.clip_norm, ck_args.seed, ck_args.label_smoothing = 0.05, 1, 0.1
    path = os.path.join(GOLD, "ref_checkpoint_quant_tiny.pt")
    checkpoint_utils.save_state(path, None, model.state_dict(), crit, opt, sched, 2, optim_history=None,
                                extra_state={"train_iterator": {"epoch": 1, "iterations_in_epoch": 2}, "val_loss": None}, args=ck_args)
    state = torch.load(path, weights_only=False)
    ost = state["last_optimizer_state"]["state"]
    print("optimizer state entries", len(ost), "of", len(params), "parameters; max index", max(ost))
    loss, gnorm = update(2)
    nxt = {"loss/2": np.float64(loss), "gnorm/2": np.float64(gnorm), "lr/3": np.float64(opt.get_lr())}
    for n, p in model.named_parameters():
        nxt["param_after3/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "ref_checkpoint_quant_tiny_next.npz"), **nxt)
    print("update 3: loss %.6f gnorm %.6f lr %.6g" % (loss, gnorm, opt.get_lr()))


def main():
    torch.set_num_threads(4)
    gen_s2t_fbank()
    with tempfile.TemporaryDirectory() as tmp:
        gen_quant(tmp)
    for f in ("s2t_fbank_tiny.npz", "w2v_quant_tiny.pt", "chimera_quant_tiny.npz", "ref_checkpoint_quant_tiny.pt", "ref_checkpoint_quant_tiny_next.npz"):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
