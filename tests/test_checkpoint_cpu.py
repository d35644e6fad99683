"""Checkpoint interop, CPU part: a checkpoint WRITTEN BY THE REFERENCE (tests/golden/ref_checkpoint_tiny.pt,
tools/ref_harness/make_ckpt_goldens.py) is read, upgraded and loaded into a freshly built model; legacy key layouts are
upgraded; files this build writes have the reference's structure."""
import os
from argparse import Namespace
from importlib import import_module

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, load_pkg

CKPT = os.path.join(GOLDEN, "ref_checkpoint_tiny.pt")


@pytest.fixture(scope="module")
def CU():
    load_pkg()
    return import_module("chimera-st_amd.checkpoint_utils")


def test_reference_checkpoint_loads_and_matches_recorded_parameters(CU):
    state = CU.load_checkpoint_to_cpu(CKPT)
    assert set(state.keys()) >= {"args", "model", "optimizer_history", "extra_state", "last_optimizer_state"}
    assert state["optimizer_history"][-1]["num_updates"] == 2 and state["optimizer_history"][-1]["optimizer_name"] == "FairseqAdam"
    assert state["extra_state"]["train_iterator"] == {"epoch": 1, "iterations_in_epoch": 2}
    (model,), args, task = CU.load_model_ensemble_and_task([CKPT])
    assert args.arch == "s2t_transformer_w2v2_interlingua_base" and type(task).__name__ == "TripletTask"
    g = load_golden("optim_tiny.npz")  # the reference's parameters after the same two updates
    own = dict(model.named_parameters())
    ref_names = [k[len("param_after/"):] for k in g if k.startswith("param_after/")]
    assert list(own.keys()) == ref_names  # same parameters in the same ORDER: flat optimizer state maps 1:1
    for n in ref_names:
        assert np.allclose(own[n].detach().numpy(), g["param_after/" + n], rtol=0, atol=3e-4), n
    # every optimizer-state entry lines up with its parameter
    ost = state["last_optimizer_state"]["state"]
    assert 0 < len(ost) <= len(ref_names)  # torch.optim creates state only for parameters that received a gradient
    for i, e in ost.items():
        assert tuple(e["exp_avg"].shape) == tuple(own[ref_names[i]].shape) and int(e["step"]) == 2, ref_names[i]


def test_legacy_layouts_are_upgraded(CU):
    """in_proj_weight/bias (multihead_attention.py:459-488), encoder layer_norms.N (transformer_layer.py:60-78), a missing
    positional-embedding buffer and pre-history optimizer fields (checkpoint_utils.py:395-430)."""
    state = CU.load_checkpoint_to_cpu(CKPT)
    sd = state["model"]
    pre = "decoder.layers.0.self_attn."
    legacy = dict(sd)
    legacy[pre + "in_proj_weight"] = torch.cat([sd[pre + "q_proj.weight"], sd[pre + "k_proj.weight"], sd[pre + "v_proj.weight"]], 0)
    legacy[pre + "in_proj_bias"] = torch.cat([sd[pre + "q_proj.bias"], sd[pre + "k_proj.bias"], sd[pre + "v_proj.bias"]], 0)
    for k in ("q_proj", "k_proj", "v_proj"):
        del legacy[pre + k + ".weight"], legacy[pre + k + ".bias"]
    enc = "encoder.transformer_layers.0."
    for old, new in (("0", "self_attn_layer_norm"), ("1", "final_layer_norm")):
        for m in ("weight", "bias"):
            legacy[enc + "layer_norms.%s.%s" % (old, m)] = legacy.pop(enc + "%s.%s" % (new, m))
    (model,), _, _ = CU.load_model_ensemble_and_task([CKPT])
    model.upgrade_state_dict(legacy)
    model.load_state_dict(legacy, strict=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    old = {"model": {}, "best_loss": 1.5, "optimizer": {"state": {}}, "epoch": 3, "batch_offset": 7, "val_loss": 2.0}
    up = CU._upgrade_state_dict(dict(old))
    assert up["last_optimizer_state"] == {"state": {}} and up["optimizer_history"][-1]["lr_scheduler_state"] == {"best": 1.5}
    assert up["extra_state"]["train_iterator"] == {"epoch": 3, "iterations_in_epoch": 7} and up["optimizer_history"][-1]["num_updates"] == 0


def test_cfg_only_pickles_do_not_need_omegaconf(CU, tmp_path):
    """A `cfg` entry pickled from a class of a package that is not installed is read as an inert placeholder."""
    import pickle
    import sys
    import types
    mod = types.ModuleType("omegaconf")
    cls = type("DictConfig", (), {"__module__": "omegaconf", "__init__": lambda self: setattr(self, "_content", {"a": 1})})
    mod.DictConfig = cls
    sys.modules["omegaconf"] = mod
    try:
        state = torch.load(CKPT, weights_only=False)
        state["cfg"] = cls()
        p = str(tmp_path / "with_cfg.pt")
        torch.save(state, p)
    finally:
        del sys.modules["omegaconf"]
    loaded = CU.load_checkpoint_to_cpu(p)
    assert type(loaded["cfg"]).__name__ == "DictConfig" and loaded["args"].arch == state["args"].arch
    with pytest.raises(IOError):
        CU.load_checkpoint_to_cpu(str(tmp_path / "missing.pt"))


def test_average_checkpoints(CU, tmp_path):
    """tools/average_checkpoints.py: fp32 mean of the floating-point entries, directory + --num-epoch-checkpoints selection."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("avg", os.path.join(os.path.dirname(GOLDEN), "..", "tools", "average_checkpoints.py"))
    avg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(avg)
    base = torch.load(CKPT, weights_only=False)
    for e, scale in ((3, 1.0), (4, 3.0), (5, 5.0), (6, 100.0)):
        st = dict(base)
        st["model"] = {k: (v * scale if v.is_floating_point() else v) for k, v in base["model"].items()}
        torch.save(st, str(tmp_path / ("checkpoint%d.pt" % e)))
    out = str(tmp_path / "avg.pt")
    avg.main(["--inputs", str(tmp_path), "--num-epoch-checkpoints", "3", "--checkpoint-upper-bound", "5", "--output", out])
    got = CU.load_checkpoint_to_cpu(out)
    for k, v in base["model"].items():
        if v.is_floating_point():
            assert torch.allclose(got["model"][k], v * 3.0, rtol=1e-6, atol=0), k
        else:
            assert torch.equal(got["model"][k], v), k
    (model,), _, _ = CU.load_model_ensemble_and_task([out])  # the averaged file is a loadable checkpoint
    assert list(model.state_dict().keys()) == list(base["model"].keys())


QCKPT = os.path.join(GOLDEN, "ref_checkpoint_quant_tiny.pt")
W2VQ = os.path.join(GOLDEN, "w2v_quant_tiny.pt")


def test_real_wav2vec2_file_with_quantizer_loads_through_the_model_path_flag(CU):
    """--w2v2-model-path <file> (models/chimera/w2v2_transformer.py:255-267: torchHLoad of {"args", "model"}, build from args,
    load_state_dict of ckpt["model"]) with a quantize_targets=True pre-training checkpoint written by the reference: the
    quantizer.* / project_q.* keys load strictly, through the enclosing model too, and state_dict() re-emits them."""
    import ast
    ref = torch.load(W2VQ, weights_only=False)
    assert ref["args"].quantize_targets and "quantizer.vars" in ref["model"] and "project_q.weight" in ref["model"]
    state = CU.load_checkpoint_to_cpu(QCKPT)
    (model,), args, task = CU.load_model_ensemble_and_task([QCKPT], arg_overrides={"w2v2_model_path": W2VQ})  # strict=True
    g = load_golden("chimera_quant_tiny.npz")
    names = ast.literal_eval(str(g["meta/param_names"]))
    assert [n for n, _ in model.named_parameters()] == names  # the reference's parameter ORDER: optimizer-state index space
    assert list(model.state_dict().keys()) == list(state["model"].keys())
    for k, v in state["model"].items():
        assert torch.equal(model.state_dict()[k].float(), v.float()) or "_float_tensor" in k, k
    # the optimizer state the reference wrote lines up entry by entry (incl. the indices after the quantizer block)
    ost = state["last_optimizer_state"]["state"]
    own = dict(model.named_parameters())
    assert max(ost) == len(names) - 1
    for i, e in ost.items():
        assert tuple(e["exp_avg"].shape) == tuple(own[names[i]].shape), names[i]
    # a model built FROM the file alone carries the file's wav2vec2 weights
    w2t = import_module("chimera-st_amd.w2v2_transformer")
    margs = Namespace(**{k: v for k, v in vars(args).items()})
    margs.w2v2_model_path = W2VQ
    fresh = task.build_model(margs)
    fsd = fresh.state_dict()
    for k, v in ref["model"].items():
        assert torch.equal(fsd["encoder.wav2vec_model." + k], v), k
