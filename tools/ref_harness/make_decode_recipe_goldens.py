#!/usr/bin/env python3
"""tests/golden/decode_recipe_tiny.npz — the FINAL decoding recipe of the reference (chimera/generate/generate-mustc-final.sh:5-8:
`--beam 10 --lenpen 1.5`) run by the REAL reference's SequenceGenerator (imported from /root/reference through ref_import.py) on the
fitted tiny Chimera model of decode_tiny.npz (its parameters are loaded from that fixture, so decode_tiny.npz itself is not touched).

Build container only:   python tools/ref_harness/make_decode_recipe_goldens.py
Holds data only — inputs and the generator's outputs (token ids, length-normalised scores, positional scores), never reference source.

Settings recorded (sequence_generator.py:179-541, len_penalty normalisation in finalize_hypos :623-624, unk penalty :321,
min_len :327-329; search.py:109-144):
  recipe        beam 10, len_penalty 1.5                       <- the final recipe
  recipe_unk    beam 10, len_penalty 1.5, unk_penalty 0.5, min_len 4
  short         beam 5,  len_penalty 0.6                       (a penalty < 1 prefers SHORT hypotheses: the other side of the branch)
  nonorm        beam 4,  len_penalty 1.5, normalize_scores off (the penalty must then be ignored)
each on the fixture's own two utterances ("a") and on three fresh utterances of other lengths ("b": audio the model was not fitted to,
so its hypotheses end at different steps and the length normalisation decides the order)."""
import ast
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import import_reference  # noqa: E402

import_reference()
import make_goldens as MG  # noqa: E402

SETTINGS = {
    "recipe": dict(beam_size=10, len_penalty=1.5),
    "recipe_unk": dict(beam_size=10, len_penalty=1.5, unk_penalty=0.5, min_len=4),
    "short": dict(beam_size=5, len_penalty=0.6),
    "nonorm": dict(beam_size=4, len_penalty=1.5, normalize_scores=False),
}


def main():
    from fairseq.models.chimera.w2v2_transformer_interlingua import S2TTransformerInterlinguaModelW2V2
    from fairseq.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(MG.OUT, "decode_tiny.npz"), allow_pickle=False)
    d = MG.make_dictionary()
    task = MG.TaskStub(d)
    with tempfile.TemporaryDirectory() as tmp:
        w2v_path = os.path.join(tmp, "w2v_tiny.pt")
        MG.build_w2v_ckpt(w2v_path, seed=11)
        torch.manual_seed(12)
        model = S2TTransformerInterlinguaModelW2V2.build_model(MG.model_args(w2v_path), task)
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("_float_tensor" in k or k == "decoder.version" for k in missing), (missing, unexpected)
    model.eval()
    # the loaded model must BE the one decode_tiny.npz was made with: its stored logits come back
    with torch.no_grad():
        (logits, _), _ = model.forward_with_internal(torch.from_numpy(g["in/src_tokens"]), torch.from_numpy(g["in/src_lengths"]),
                                                     torch.from_numpy(g["in/prev_output_tokens"]))
    assert float((logits - torch.from_numpy(g["out/st_logits"])).abs().max()) < 1e-5

    inputs = {"a": (torch.from_numpy(g["in/src_tokens"]), torch.from_numpy(g["in/src_lengths"]))}
    gen_b = torch.Generator().manual_seed(31)
    S = (4800, 3520, 2560)
    audio = torch.zeros(len(S), max(S))
    for i, s in enumerate(S):
        audio[i, :s] = 0.1 * torch.randn(s, generator=gen_b)
    inputs["b"] = (audio, torch.tensor(S, dtype=torch.long))

    out = {"meta/settings": np.array(repr(SETTINGS)), "meta/max_len_b": np.int64(12)}
    for tag, (src, lens) in inputs.items():
        out["in/%s/src_tokens" % tag] = src.numpy()
        out["in/%s/src_lengths" % tag] = lens.numpy()
        for name, kw in SETTINGS.items():
            kw = dict(kw)
            kw.setdefault("min_len", 1)
            gen = SequenceGenerator([model], d, max_len_a=0, max_len_b=12, **kw)
            with torch.no_grad():
                hyps = gen.generate([model], {"net_input": {"src_tokens": src, "src_lengths": lens}})
            for b, h in enumerate(hyps):
                out["gen/%s/%s/b%d/n" % (name, tag, b)] = np.int64(len(h))
                for r, hyp in enumerate(h):
                    key = "gen/%s/%s/b%d/r%d/" % (name, tag, b, r)
                    out[key + "tokens"] = hyp["tokens"].numpy()
                    out[key + "score"] = np.float64(float(hyp["score"]))
                    out[key + "pos_scores"] = hyp["positional_scores"].numpy()
                lens_ = [len(x["tokens"]) for x in h]
                print(name, tag, b, "n", len(h), "lengths", lens_, "best", h[0]["tokens"].tolist(), "%.4f" % float(h[0]["score"]))
    np.savez_compressed(os.path.join(MG.OUT, "decode_recipe_tiny.npz"), **out)


if __name__ == "__main__":
    main()
