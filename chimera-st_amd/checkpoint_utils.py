"""Checkpoint interop with the reference — mirror of fairseq/checkpoint_utils.py:
load_checkpoint_to_cpu (:225-241), _upgrade_state_dict (:395-476), load_model_ensemble_and_task (:269-309), save_state (:344-392)
and of the trainer's save/load (fairseq/trainer.py:270-395).

File format (kept, so either side can read the other's files): a torch pickle of
  {"args": argparse.Namespace, "cfg": DictConfig | None, "model": state_dict, "optimizer_history": [{criterion_name,
   optimizer_name, lr_scheduler_state, num_updates}], "extra_state": {train_iterator, val_loss, ...}, "last_optimizer_state":
   torch-optimizer state dict {"state": {i: {step, exp_avg, exp_avg_sq}}, "param_groups": [...]}}.
The reference's optimizer state is per parameter (optim/adam.py, fp32 training) or ONE flat fp32 tensor (FP16Optimizer,
optim/fp16_optimizer.py:33-60); both map onto this build's flat master / moment buffers in model.parameters() order.
Files written by newer fairseq builds pickle an omegaconf DictConfig under "cfg"; omegaconf is not required here: unknown
classes are unpickled as inert placeholders and the Namespace under "args" (always written by this fork, trainer.py:283-300)
is what the model is rebuilt from."""
import argparse
import collections
import io
import os
import pickle
from argparse import Namespace

import torch

from . import registry


class _Placeholder:
    """Stands in for a class whose module is not installed (omegaconf.*): keeps the pickled state, does nothing."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        self._state = state

    def __call__(self, *a, **k):
        return _Placeholder(*a, **k)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if module.split(".")[0] in ("omegaconf", "hydra", "fairseq"):
                return type(name, (_Placeholder,), {"__module__": module})
            raise


class _TolerantPickle:
    """pickle_module for torch.load(weights_only=False)."""
    __name__ = "chimera_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _TolerantUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps = pickle.dump, pickle.dumps
    Pickler, HIGHEST_PROTOCOL, PickleError, UnpicklingError = pickle.Pickler, pickle.HIGHEST_PROTOCOL, pickle.PickleError, pickle.UnpicklingError


def load_checkpoint_to_cpu(path, arg_overrides=None):
    """checkpoint_utils.py:225-241.  Only load files you trust: like the reference, this unpickles arbitrary objects."""
    if not os.path.exists(path):
        raise IOError("Model file not found: {}".format(path))
    state = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    if state.get("args") is not None and arg_overrides is not None:
        for k, v in arg_overrides.items():
            setattr(state["args"], k, v)
    return _upgrade_state_dict(state)


def _upgrade_state_dict(state):
    """checkpoint_utils.py:395-476 (the entries that apply to files this fork can have written)."""
    if "optimizer_history" not in state:
        state["optimizer_history"] = [{"criterion_name": "CrossEntropyCriterion", "best_loss": state["best_loss"]}]
        state["last_optimizer_state"] = state["optimizer"]
        del state["optimizer"], state["best_loss"]
    if "epoch" in state and "extra_state" not in state:
        state["extra_state"] = {"epoch": state["epoch"], "batch_offset": state["batch_offset"], "val_loss": state["val_loss"]}
        del state["epoch"], state["batch_offset"], state["val_loss"]
    last = state["optimizer_history"][-1]
    if "optimizer" in last:
        state["last_optimizer_state"] = last["optimizer"]
        for h in state["optimizer_history"]:
            del h["optimizer"]
    last.setdefault("optimizer_name", "FairseqNAG")
    if "lr_scheduler_state" not in last:
        last["lr_scheduler_state"] = {"best": last.pop("best_loss", None)}
    last.setdefault("num_updates", 0)
    state.setdefault("extra_state", {})
    if "train_iterator" not in state["extra_state"]:
        state["extra_state"]["train_iterator"] = {"epoch": state["extra_state"].get("epoch", 1),
                                                  "iterations_in_epoch": state["extra_state"].get("batch_offset", 0)}
    if state.get("args") is not None:
        a = state["args"]
        if not hasattr(a, "task"):
            a.task = "translation"
        ti = state["extra_state"]["train_iterator"]
        if ti is not None:
            ti["epoch"] = max(ti.get("epoch", 1), 1)
        if hasattr(a, "remove_bpe"):
            a.post_process = a.remove_bpe
    return state


def _register_embedded_w2v(args):
    """The reference rebuilds the wav2vec2 frontend from a SEPARATE file (args.w2v2_model_path -> wav2vec_small.pt,
    w2v2_transformer.py:255-267) even though its weights are also inside the Chimera checkpoint.  When that file is not around
    (or the checkpoint carries `w2v_args`), build the frontend from the recorded hyper-parameters; the weights come from the
    checkpoint's own model state."""
    from . import w2v2_transformer as W
    path = getattr(args, "w2v2_model_path", None)
    w2v_args = getattr(args, "w2v_args", None)
    if isinstance(path, str) and path.startswith("synthetic:"):
        name = path.split(":", 1)[1]
        if w2v_args is not None and name not in W.SYNTHETIC_W2V:
            W.SYNTHETIC_W2V[name] = Namespace(**w2v_args) if isinstance(w2v_args, dict) else w2v_args
    elif path is not None and not os.path.exists(str(path)):
        name = "from_checkpoint_%x" % (hash(str(path)) & 0xFFFFFF)
        W.SYNTHETIC_W2V[name] = (Namespace(**w2v_args) if isinstance(w2v_args, dict) else w2v_args) if w2v_args is not None \
            else W.wav2vec_small_args()  # the published Chimera checkpoints use wav2vec_small
        args.w2v2_model_path = "synthetic:" + name


def load_model_ensemble_and_task(filenames, arg_overrides=None, task=None, strict=True):
    """checkpoint_utils.py:269-309: [(model built from the checkpoint's args, state loaded)], args, task."""
    from . import criterions, s2t_transformer, tasks, w2v2_transformer, w2v2_transformer_interlingua, wav2vec2  # noqa: F401 (registries)
    ensemble, args = [], None
    for filename in filenames:
        state = load_checkpoint_to_cpu(filename, arg_overrides)
        args = state.get("args")
        if args is None:
            raise RuntimeError("checkpoint %s has no `args` Namespace (keys: %s); a `cfg`-only checkpoint needs omegaconf to be "
                               "decoded" % (filename, list(state.keys())))
        _register_embedded_w2v(args)
        data = getattr(args, "data", None)
        if task is None and not (data and os.path.isdir(str(data))) and "decoder.embed_tokens.weight" in state["model"]:
            # the dictionary file of the training run is not reachable: a placeholder dictionary of the checkpoint's vocabulary
            # size keeps the model loadable (token ids are what the model consumes; symbols are needed only to print text)
            args.data, args.synthetic_vocab_size = None, int(state["model"]["decoder.embed_tokens.weight"].shape[0])
        if task is None:
            task = registry.setup_task(args)
        registry.ARCH_CONFIG_REGISTRY[args.arch](args)
        model = task.build_model(args)
        sd = state["model"]
        model.upgrade_state_dict(sd)
        model.load_state_dict(sd, strict=strict)
        ensemble.append(model)
    return ensemble, args, task


def save_state(filename, args, model_state_dict, criterion, optimizer, num_updates, optim_history=None, extra_state=None):
    """checkpoint_utils.py:344-392 — same keys, `cfg` left None (the Namespace under `args` is authoritative)."""
    state = {
        "cfg": None,
        "args": args,
        "model": collections.OrderedDict((k, v.detach().cpu()) for k, v in (model_state_dict or {}).items()),
        "optimizer_history": (optim_history or []) + [{
            "criterion_name": criterion.__class__.__name__,
            "optimizer_name": "FairseqAdam",
            "lr_scheduler_state": {"best": None},
            "num_updates": num_updates,
        }],
        "extra_state": extra_state or {},
    }
    if not getattr(args, "no_save_optimizer_state", False):
        state["last_optimizer_state"] = optimizer.fairseq_state_dict()
    tmp = filename + ".tmp"
    with open(tmp, "wb") as f:
        torch.save(state, f)
    os.replace(tmp, filename)
    return state
