#!/usr/bin/env python3
"""Time the REAL reference (imported from /root/reference through ref_import.py) beside the CPU oracle on the same
full-size model, weights and sample — SURVEY §8(d) "CPU baseline": shows that the oracle `bench.py` times as
`cpu_baseline` on the GPU box (where the reference cannot travel) runs at the reference's own CPU speed, and that the two
agree on the loss of a full-size model.

Run in the build container only:   python tools/ref_harness/time_ref_vs_oracle.py [--seconds 10] [--tokens 64]
One update = forward + backward + Adam (reference: fairseq/optim/adam.py Adam; oracle: adam_step), fp32, all host
threads, 1 warm-up on a 1 s utterance + 1 timed update each.  Model: `s2t_transformer_w2v2` with the bench dimensions
(wav2vec2-small front end + d512/ffn2048/8h 12+6 layers, V = 10 000), dropout 0."""
import argparse
import os
import sys
import tempfile
import time
from argparse import Namespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from ref_import import import_reference  # noqa: E402
import make_goldens as G  # noqa: E402

V = 10000


def dictionary():
    from fairseq.data import Dictionary
    d = Dictionary()
    for i in range(V - 4):
        d.add_symbol("w%d" % i)
    return d


def sample_for(d, seconds, tokens, seed):
    from fairseq.data import data_utils
    g = torch.Generator().manual_seed(seed)
    S = int(seconds * 16000)
    audio = 0.1 * torch.randn(1, S, generator=g)
    tgt = [torch.cat([torch.randint(4, V, (tokens,), generator=g), torch.tensor([d.eos()])])]
    return {
        "id": torch.arange(1),
        "net_input": {"src_tokens": audio, "src_lengths": torch.tensor([S]), "mask": False,
                      "prev_output_tokens": data_utils.collate_tokens(tgt, d.pad(), d.eos(), False, True)},
        "target": data_utils.collate_tokens(tgt, d.pad(), d.eos(), False, False),
        "target_lengths": torch.tensor([tokens + 1]), "ntokens": tokens + 1, "nsentences": 1,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--tokens", type=int, default=64)
    a = ap.parse_args()
    import_reference()
    from fairseq.criterions.label_smoothed_cross_entropy import LabelSmoothedCrossEntropyCriterion
    from fairseq.models.chimera.w2v2_transformer import S2TTransformerModelW2V2, base_architecture as s2t_transformer_w2v2
    from fairseq.models.wav2vec.wav2vec2 import Wav2Vec2Model
    from fairseq.optim.adam import Adam
    from oracle import chimera_oracle as O

    w2v = dict(G.W2V_TINY)
    w2v.update(conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2", encoder_layers=12,
               encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12, conv_pos=128, conv_pos_groups=16,
               final_dim=256, latent_vars=320, quantize_targets=True)
    d = dictionary()
    task = G.TaskStub(d)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "w2v_small_random.pt")
        torch.manual_seed(1)
        wns = Namespace(**w2v)
        torch.save({"args": wns, "model": Wav2Vec2Model.build_model(wns, task=None).state_dict()}, path)
        ns = Namespace(w2v2_model_path=path, use_asr_finetune_w2v=False, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                       share_decoder_input_output_embed=True, max_source_positions=2000000, max_target_positions=1024,
                       tie_adaptive_weights=False, quant_noise_pq_block_size=8)
        s2t_transformer_w2v2(ns)  # the arch preset bench.py uses: d512 / ffn2048 / 8 heads / 12 + 6 layers
        ns.dropout = ns.attention_dropout = ns.activation_dropout = 0.0
        model = S2TTransformerModelW2V2.build_model(ns, task)
    model.encoder.wav2vec_model.encoder.pos_conv.register_forward_pre_hook(lambda m, i: (i[0].contiguous(),))
    model.train()
    crit = LabelSmoothedCrossEntropyCriterion(task, False, 0.1)
    nparam = sum(p.numel() for p in model.parameters())
    cores = torch.get_num_threads()

    cfg = dict(conv_layers=eval(wns.conv_feature_layers), conv_pos=wns.conv_pos, conv_pos_groups=wns.conv_pos_groups,
               w2v_layers=wns.encoder_layers, w2v_heads=wns.encoder_attention_heads, feature_grad_mult=wns.feature_grad_mult,
               d=ns.encoder_embed_dim, heads=ns.encoder_attention_heads, dec_heads=ns.decoder_attention_heads,
               enc_layers=ns.encoder_layers, dec_layers=ns.decoder_layers, mem_layers=0)
    def fresh_p():
        p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "_float_tensor" not in k and k != "decoder.version")
             for k, v in model.state_dict().items()}
        p["decoder.output_projection.weight"] = p["decoder.embed_tokens.weight"]
        return p

    opt = Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0)

    def ref_update(s):
        t0 = time.time()
        opt.zero_grad()
        loss, _, _ = crit(model, s)
        loss.backward()
        opt.step()
        return time.time() - t0, float(loss)

    def oracle_update(p, s):
        t0 = time.time()
        out = O.lsce_criterion(p, s, cfg)
        out["loss"].backward()
        leaves = list({id(t): t for t in p.values() if t.requires_grad and t.grad is not None}.values())
        with torch.no_grad():
            for t in leaves:
                O.adam_step(t, t.grad, torch.zeros_like(t), torch.zeros_like(t), 1, 2e-4)
                t.grad = None
        return time.time() - t0, float(out["loss"])

    warm, s = sample_for(d, 1.0, 8, 5), sample_for(d, a.seconds, a.tokens, 6)
    # both sides run their timed update from the same, un-updated weights (the warm-up updates are discarded)
    oracle_update(fresh_p(), warm)
    t_or, l_or = oracle_update(fresh_p(), s)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref_update(warm)
    model.load_state_dict(sd)
    t_ref, l_ref = ref_update(s)
    print("model: s2t_transformer_w2v2 full size, %d parameters; sample: 1 utterance x %.0f s + %d target tokens; %d host threads, fp32"
          % (nparam, a.seconds, a.tokens + 1, cores))
    print("reference (fairseq, /root/reference): %.2f s per update = %.4f utterances/s   loss %.6f" % (t_ref, 1 / t_ref, l_ref))
    print("oracle    (oracle/chimera_oracle.py): %.2f s per update = %.4f utterances/s   loss %.6f" % (t_or, 1 / t_or, l_or))
    print("oracle / reference time: %.2f   |loss difference| %.2e (relative %.1e)" % (t_or / t_ref, abs(l_or - l_ref), abs(l_or - l_ref) / abs(l_ref)))


if __name__ == "__main__":
    main()
