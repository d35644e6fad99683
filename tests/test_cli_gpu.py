"""End-to-end drivers on a real MI355X: fairseq-train style flags -> manifests -> sharded batches -> HIP training
updates -> reference-format checkpoints -> resume -> fairseq-generate style decoding of the dev split, on the tiny manifest root
of tests/golden/data_tiny."""
import ast
import json
import os
import shutil
from argparse import Namespace
from importlib import import_module

import pytest
import torch

from conftest import GOLDEN, load_golden, load_pkg

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLDEN, "data_tiny")


def _events(capsys):
    return [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_train_resume_generate(tmp_path, capsys):
    load_pkg()
    cli = import_module("chimera-st_amd.cli")
    w2t = import_module("chimera-st_amd.w2v2_transformer")
    g = load_golden("chimera_tiny.npz")
    w2t.SYNTHETIC_W2V["golden_tiny"] = Namespace(**ast.literal_eval(str(g["meta/w2v_args"])))
    root = tmp_path / "data"
    root.mkdir()
    for f in os.listdir(DATA):
        if not f.endswith(".wav"):
            shutil.copy(os.path.join(DATA, f), root / f)
    (root / "config_wave.yaml").write_text((root / "config_wave.yaml").read_text().replace("AUDIO_ROOT", DATA))
    save = str(tmp_path / "ckpt")
    common = [str(root), "--task", "triplet", "--train-subset", "train_st", "--valid-subset", "dev_st", "--config-yaml", "config_wave.yaml",
              "--max-tokens", "12000", "--max-source-positions", "2000000", "--save-dir", save,
              "--criterion", "triplet_st_mt_contrastive", "--label-smoothing", "0.1",
              "--arch", "s2t_transformer_w2v2_interlingua_base", "--share-decoder-input-output-embed",
              "--w2v2-model-path", "synthetic:golden_tiny", "--encoder-layers", "2", "--encoder-embed-dim", "64",
              "--encoder-ffn-embed-dim", "128", "--encoder-attention-heads", "2", "--decoder-attention-heads", "2", "--decoder-layers", "2",
              "--conv-channels", "64", "--interlingua-length", "8", "--interlingua-layers", "2", "--dropout", "0.1",
              "--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--clip-norm", "0.0", "--lr", "2e-3", "--lr-scheduler", "inverse_sqrt",
              "--weight-decay", "0.0001", "--warmup-updates", "2", "--fp16", "--update-freq", "2", "--num-workers", "1",
              "--ddp-backend", "no_c10d", "--best-checkpoint-metric", "st_loss", "--seed", "1", "--log-interval", "1"]
    tr = cli.train_main(common + ["--max-epoch", "3"])
    ev = _events(capsys)
    assert ev[0]["event"] == "start" and ev[0]["dtype"] == "torch.bfloat16" and ev[0]["train_examples"] == 11
    train = [e for e in ev if e["event"] == "train"]
    valid = [e for e in ev if e["event"] == "valid"]
    assert len(train) == 3 and len(valid) == 3 and train[-1]["loss"] < train[0]["loss"]
    assert all(e["nsentences"] == 11 for e in train)  # every manifest row was consumed once per epoch
    for f in ("checkpoint1.pt", "checkpoint3.pt", "checkpoint_last.pt", "checkpoint_best.pt"):
        assert os.path.exists(os.path.join(save, f)), f
    n3 = tr.num_updates
    state = torch.load(os.path.join(save, "checkpoint_last.pt"), weights_only=False)
    assert state["extra_state"]["train_iterator"] == {"epoch": 4, "iterations_in_epoch": 0} and state["optimizer_history"][-1]["num_updates"] == n3
    # resume: picks up checkpoint_last.pt, continues the update counter and the epoch numbering
    tr2 = cli.train_main(common + ["--max-epoch", "4"])
    ev = _events(capsys)
    loaded = [e for e in ev if e["event"] == "loaded_checkpoint"]
    assert loaded and loaded[0]["num_updates"] == n3 and loaded[0]["epoch"] == 4
    assert [e["epoch"] for e in ev if e["event"] == "train"] == [4] and tr2.num_updates > n3
    # decode the dev split from the checkpoint
    summary = cli.generate_main([str(root), "--task", "triplet", "--config-yaml", "config_wave.yaml", "--path", os.path.join(save, "checkpoint_last.pt"),
                                 "--gen-subset", "dev_st", "--max-tokens", "12000", "--beam", "3", "--max-len-b", "10", "--lenpen", "1.5",
                                 "--max-source-positions", "2000000", "--fp16"])
    lines = capsys.readouterr().out.splitlines()
    assert summary["sentences"] == 4 and summary["tokens"] >= 4
    assert sum(l.startswith("H-") for l in lines) == 4 and sum(l.startswith("T-") for l in lines) == 4


def _tiny_cli_flags(root, save, w2v):
    return [str(root), "--task", "triplet", "--train-subset", "train_st", "--valid-subset", "dev_st", "--config-yaml", "config_wave.yaml",
            "--max-tokens", "12000", "--max-source-positions", "2000000", "--save-dir", save,
            "--criterion", "triplet_st_mt_contrastive", "--label-smoothing", "0.1",
            "--arch", "s2t_transformer_w2v2_interlingua_base", "--share-decoder-input-output-embed",
            "--w2v2-model-path", w2v, "--encoder-layers", "2", "--encoder-embed-dim", "64",
            "--encoder-ffn-embed-dim", "128", "--encoder-attention-heads", "2", "--decoder-attention-heads", "2", "--decoder-layers", "2",
            "--conv-channels", "64", "--interlingua-length", "8", "--interlingua-layers", "2", "--dropout", "0.1",
            "--optimizer", "adam", "--adam-betas", "(0.9, 0.98)", "--clip-norm", "0.0", "--lr", "2e-3", "--lr-scheduler", "inverse_sqrt",
            "--weight-decay", "0.0001", "--warmup-updates", "2", "--fp16", "--update-freq", "1", "--num-workers", "1",
            "--ddp-backend", "no_c10d", "--best-checkpoint-metric", "st_loss", "--seed", "1", "--log-interval", "1"]


def test_zero_sharded_training_saves_through_the_cli_on_two_ranks(tmp_path, capsys):
    """`fairseq_train.py … --distributed-world-size 2 --zero-sharding os` started plainly (the script launches its two ranks; they
    share this box's GPU over gloo): with the optimizer's moments sharded, writing a checkpoint is a collective — every rank has to
    enter Trainer.save_checkpoint, rank 0 alone writes.  The job must finish (it used to hang at the first save: cli.save returned
    early on rank != 0), leave mid-epoch and end-of-epoch checkpoints whose optimizer state is the FULL one, and a 1-rank run must
    resume from it."""
    import subprocess
    import sys
    load_pkg()
    g = load_golden("chimera_tiny.npz")
    w2v = str(tmp_path / "w2v_tiny_random.pt")
    torch.save({"args": Namespace(**ast.literal_eval(str(g["meta/w2v_args"]))), "model": None}, w2v)
    root = tmp_path / "data"
    root.mkdir()
    for f in os.listdir(DATA):
        if not f.endswith(".wav"):
            shutil.copy(os.path.join(DATA, f), root / f)
    (root / "config_wave.yaml").write_text((root / "config_wave.yaml").read_text().replace("AUDIO_ROOT", DATA))
    save = str(tmp_path / "ckpt")
    flags = _tiny_cli_flags(root, save, w2v)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CST_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "fairseq_train.py")] + flags +
                       ["--max-epoch", "1", "--distributed-world-size", "2", "--zero-sharding", "os", "--save-interval-updates", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=repo)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ev = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert ev[0]["event"] == "start" and ev[0]["world_size"] == 2 and ev[-1]["event"] == "done" and ev[-1]["num_updates"] >= 1
    assert os.path.exists(os.path.join(save, "checkpoint1.pt")) and os.path.exists(os.path.join(save, "checkpoint_1_1.pt"))
    state = torch.load(os.path.join(save, "checkpoint_last.pt"), weights_only=False)
    moments = state["last_optimizer_state"]["state"]
    # a single process resumes from the two-rank checkpoint (programmatic call: no self-launch, the Trainer comes back)
    cli = import_module("chimera-st_amd.cli")
    capsys.readouterr()
    tr = cli.train_main(flags + ["--max-epoch", "2"])
    ev2 = _events(capsys)
    loaded = [e for e in ev2 if e["event"] == "loaded_checkpoint"]
    assert tr is not None and loaded and loaded[0]["num_updates"] == ev[-1]["num_updates"] and tr.num_updates > ev[-1]["num_updates"]
    nparam = sum(p.numel() for p in tr.get_model().parameters())
    assert sum(m["exp_avg"].numel() for m in moments.values()) == nparam, "the checkpoint must carry every rank's moments, not a shard"
