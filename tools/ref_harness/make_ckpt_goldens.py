#!/usr/bin/env python3
"""Checkpoint-interop fixture: the REAL reference writes a checkpoint with its own
fairseq/checkpoint_utils.py:save_state (:344-392) after the two optimizer updates of optim_tiny.npz — model state dict, args
Namespace, optimizer_history, extra_state and last_optimizer_state (torch Adam state: per-parameter step/exp_avg/exp_avg_sq) —
then performs a THIRD update, whose loss / grad norm / lr / parameters are stored next to it.  The product must load the .pt,
resume, and reproduce update 3.

  tests/golden/ref_checkpoint_tiny.pt       written by the reference (pickle of dicts / tensors / argparse.Namespace only)
  tests/golden/ref_checkpoint_tiny_next.npz update 3 as the reference computed it

Run in the build container only:  python tools/ref_harness/make_ckpt_goldens.py"""
import ast
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


def main():
    from fairseq import checkpoint_utils
    from fairseq.criterions.triplet_st_mt_contrastive import TripletSTMTContrastiveCriterion
    from fairseq.models.chimera.w2v2_transformer_interlingua import S2TTransformerInterlinguaModelW2V2
    from fairseq.optim.adam import FairseqAdam
    from fairseq.optim.lr_scheduler.inverse_square_root_schedule import InverseSquareRootSchedule

    g0 = dict(np.load(os.path.join(GOLD, "chimera_tiny.npz")))
    g1 = dict(np.load(os.path.join(GOLD, "optim_tiny.npz")))
    tmp = tempfile.mkdtemp()
    d = G.make_dictionary()
    task = G.TaskStub(d)
    w2v_path = os.path.join(tmp, "w2v_tiny.pt")
    G.build_w2v_ckpt(w2v_path, seed=11)
    args = G.model_args(w2v_path)
    model = S2TTransformerInterlinguaModelW2V2.build_model(args, task)
    model.load_state_dict({k[len("param/"):]: torch.from_numpy(v) for k, v in g0.items() if k.startswith("param/")})
    model.encoder.wav2vec_model.encoder.pos_conv.register_forward_pre_hook(lambda m, i: (i[0].contiguous(),))
    model.train()
    crit = TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1, 0, False, None, [None, None])
    sample = {"net_input": {k: torch.from_numpy(g0["in/" + k]) for k in ("src_tokens", "src_lengths", "prev_output_tokens")},
              "target": torch.from_numpy(g0["in/target"]), "target_lengths": torch.from_numpy(g0["in/target_lengths"]),
              "src_text": torch.from_numpy(g0["in/src_text"]), "src_text_lengths": torch.from_numpy(g0["in/src_text_lengths"]),
              "ntokens": int(g0["in/ntokens"])}
    sample["net_input"]["mask"] = False
    oargs = Namespace(**ast.literal_eval(str(g1["meta/optim_args"])))
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
        loss, gnorm = update(step)
        assert abs(loss - float(g1["loss/%d" % step])) < 1e-6 * abs(loss), "this script must retrace optim_tiny.npz"
    for n, p in model.named_parameters():
        assert np.allclose(p.detach().numpy(), g1["param_after/" + n], rtol=0, atol=3e-4), n  # Adam's m/sqrt(v) amplifies the run-to-run CPU reduction-order noise of tiny gradients

    ck_args = Namespace(**{k: v for k, v in vars(args).items() if k != "w2v2_model_path"})
    ck_args.w2v2_model_path = "synthetic:golden_tiny"  # the wav2vec2 weights are inside this checkpoint's model state
    ck_args.arch, ck_args.task, ck_args.criterion = "s2t_transformer_w2v2_interlingua_base", "triplet", "triplet_st_mt_contrastive"
    ck_args.optimizer, ck_args.lr_scheduler, ck_args.no_save_optimizer_state = "adam", "inverse_sqrt", False
    for k, v in vars(oargs).items():
        setattr(ck_args, k, v)
    ck_args.clip_norm, ck_args.seed, ck_args.label_smoothing = 0.05, 1, 0.1
    ck_args.w2v_args = G.W2V_TINY  # not a reference field: lets a loader rebuild the frontend without the separate wav2vec_small.pt
    path = os.path.join(GOLD, "ref_checkpoint_tiny.pt")
    checkpoint_utils.save_state(path, None, model.state_dict(), crit, opt, sched, 2, optim_history=None,
                                extra_state={"train_iterator": {"epoch": 1, "iterations_in_epoch": 2}, "val_loss": None}, args=ck_args)
    state = torch.load(path, weights_only=False)
    print("checkpoint keys", list(state.keys()), "| optimizer state entries", len(state["last_optimizer_state"]["state"]),
          "| param_groups", {k: v for k, v in state["last_optimizer_state"]["param_groups"][0].items() if k != "params"})

    loss, gnorm = update(2)
    out = {"loss/2": np.float64(loss), "gnorm/2": np.float64(gnorm), "lr/3": np.float64(opt.get_lr())}
    for n, p in model.named_parameters():
        out["param_after3/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "ref_checkpoint_tiny_next.npz"), **out)
    print("update 3: loss %.6f gnorm %.6f lr %.6g; wrote %d bytes" % (loss, gnorm, opt.get_lr(), os.path.getsize(path)))


if __name__ == "__main__":
    main()
