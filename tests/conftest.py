import ast
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# deferred reductions (kernels.DEFER): every deferred gradient holds NaN until the flush has written it, in every test of the suite
# (and in the processes the tests spawn) — a gradient that is read too early cannot pass for a stale but plausible value
os.environ.setdefault("CST_DEFER_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite builds full-size models and runs the oracle on the CPU: keep torch's intra-op pool inside the container's CPU quota
    # (hostcfg.py — on a 256-thread GPU box with a 16-CPU quota the default pool gets the process throttled)
    importlib.import_module("chimera-st_amd.hostcfg").limit_host_threads()


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: g[k] for k in g.files}


def golden_cfg(g):
    """Oracle / host config dict from the fixture's recorded reference args."""
    w = ast.literal_eval(str(g["meta/w2v_args"]))
    m = ast.literal_eval(str(g["meta/model_args"]))
    return dict(
        conv_layers=eval(w["conv_feature_layers"]),
        conv_pos=w["conv_pos"], conv_pos_groups=w["conv_pos_groups"],
        w2v_layers=w["encoder_layers"], w2v_heads=w["encoder_attention_heads"],
        w2v_dim=w["encoder_embed_dim"], w2v_ffn=w["encoder_ffn_embed_dim"],
        feature_grad_mult=w["feature_grad_mult"],
        d=m["encoder_embed_dim"], heads=m["encoder_attention_heads"], dec_heads=m["decoder_attention_heads"],
        ffn=m["encoder_ffn_embed_dim"], enc_layers=m["encoder_layers"], mem_layers=m["interlingua_layers"],
        mem_len=m["interlingua_length"], dec_layers=m["decoder_layers"], conv_channels=m["conv_channels"],
    )


def golden_params(g, requires_grad=False):
    p = {}
    for k, v in g.items():
        if k.startswith("param/"):
            t = torch.from_numpy(np.array(v))
            if requires_grad and t.is_floating_point() and "_float_tensor" not in k and "version" not in k:
                t.requires_grad_(True)
            p[k[len("param/"):]] = t
    return p


def golden_sample(g):
    s = {
        "net_input": {
            "src_tokens": torch.from_numpy(g["in/src_tokens"]),
            "src_lengths": torch.from_numpy(g["in/src_lengths"]),
            "prev_output_tokens": torch.from_numpy(g["in/prev_output_tokens"]),
            "mask": False,
        },
        "target": torch.from_numpy(g["in/target"]),
    }
    if "in/ntokens" in g:
        s["ntokens"] = int(g["in/ntokens"])
    for k in ("src_text", "src_text_lengths", "target_lengths"):
        if "in/" + k in g:
            s[k] = torch.from_numpy(g["in/" + k])
    s["nsentences"] = s["target"].size(0)
    return s


def load_pkg():
    """The package directory is `chimera-st_amd` (hyphenated, as the brief names it)."""
    return importlib.import_module("chimera-st_amd")
