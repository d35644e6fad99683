#!/usr/bin/env python3
"""Golden fixture for the input pipeline: a tiny MuST-C-style manifest root (tests/golden/data_tiny/: two 16-bit
PCM WAV files, train/dev TSV manifests whose `audio` column is "<wav>:<offset>:<length>", the data-config YAML, a fairseq
dictionary file — all synthetic, generated HERE) is read by the REAL reference (TripletDatasetCreator.from_tsv, ordered_indices,
filter_indices_by_size, the Cython batch_by_size_fast compiled from the reference's .pyx into a temp dir, the collater, the
epoch shuffling + sharding of EpochBatchIterator) and the results are stored in tests/golden/data_tiny.npz.

Run in the build container only:  python tools/ref_harness/make_data_goldens.py"""
import importlib.util
import os
import subprocess
import sys
import tempfile
import wave
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_import import REF, import_reference  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "data_tiny"))
OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "data_tiny.npz"))
WORDS = ["▁the", "▁a", "▁cat", "▁dog", "s", "▁sat", "▁on", "▁mat", "ing", "▁run", "▁und", "▁der", "▁die", "▁haus", "▁ist", "en", "▁zu"]


def make_dataset():
    os.makedirs(ROOT, exist_ok=True)
    rng = np.random.RandomState(7)
    wav_len = {"talk_a.wav": 26000, "talk_b.wav": 19000}
    for name, n in wav_len.items():
        x = (rng.randn(n) * 3000).clip(-32768, 32767).astype("<i2")
        with wave.open(os.path.join(ROOT, name), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(x.tobytes())
    with open(os.path.join(ROOT, "dict.txt"), "w", encoding="utf-8") as f:
        for i, wd in enumerate(WORDS):
            f.write("%s %d\n" % (wd, 100 - i))
    with open(os.path.join(ROOT, "config_wave.yaml"), "w") as f:
        f.write("audio_root: %s\nbpe_tokenizer:\n  bpe: null\nsrc_bpe_tokenizer:\n  bpe: null\ninput_channels: 1\n"
                "input_feat_per_channel: 80\nsampling_alpha: 1.0\nsrc_vocab_filename: dict.txt\nuse_audio_input: true\n"
                "vocab_filename: dict.txt\n" % "AUDIO_ROOT")

    def rows(n, seed, files):
        r = np.random.RandomState(seed)
        out = []
        for i in range(n):
            fn = files[i % len(files)]
            length = int(r.randint(1600, 6400))
            off = int(r.randint(0, wav_len[fn] - length))
            tgt = " ".join(r.choice(WORDS[10:] + ["▁unknownword"], size=r.randint(2, 9)))
            src = " ".join(r.choice(WORDS[:10], size=r.randint(2, 9)))
            out.append(("utt_%d_%d" % (seed, i), "%s:%d:%d" % (fn, off, length), str(length), tgt, src, "spk%d" % (i % 3)))
        return out

    for split, n, seed in (("train_st", 11, 3), ("dev_st", 4, 5)):
        with open(os.path.join(ROOT, split + ".tsv"), "w", encoding="utf-8") as f:
            f.write("id\taudio\tn_frames\ttgt_text\tsrc_text\tspeaker\n")
            for r_ in rows(n, seed, list(wav_len)):
                f.write("\t".join(r_) + "\n")


def build_cython_batcher(tmp):
    """Compile the reference's data_utils_fast.pyx (unmodified, read in place) into `tmp` and register it under its package name."""
    src = os.path.join(REF, "fairseq", "data", "data_utils_fast.pyx")
    cpp = os.path.join(tmp, "data_utils_fast.cpp")
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", src, "-o", cpp])
    import sysconfig
    so = os.path.join(tmp, "data_utils_fast" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-w", "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
                           cpp, "-o", so])
    spec = importlib.util.spec_from_file_location("fairseq.data.data_utils_fast", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["fairseq.data.data_utils_fast"] = mod


def main():
    make_dataset()
    import_reference()
    tmp = tempfile.mkdtemp()
    build_cython_batcher(tmp)
    from fairseq.data import Dictionary, data_utils, iterators
    from fairseq.data.audio.triplet_dataset import TripletDataConfig, TripletDatasetCreator

    # the committed YAML carries a placeholder audio_root (the fixture is relocatable); resolve it in a temp copy
    cfg_path = os.path.join(tmp, "config_wave.yaml")
    open(cfg_path, "w").write(open(os.path.join(ROOT, "config_wave.yaml")).read().replace("AUDIO_ROOT", ROOT))
    cfg = TripletDataConfig(cfg_path)
    d = Dictionary.load(os.path.join(ROOT, "dict.txt"))
    out = {"dict_len": np.array(len(d))}
    for split, train in (("train_st", True), ("dev_st", False)):
        ds = TripletDatasetCreator.from_tsv(ROOT, cfg, split, d, d, None, None, None, is_train_split=train, epoch=1, seed=1)
        with data_utils.numpy_seed(1):
            idx = ds.ordered_indices()
        out[split + "/ordered"] = np.asarray(idx)
        idx_f, ignored = ds.filter_indices_by_size(idx, (6000, 7))
        out[split + "/filtered"] = np.asarray(idx_f)
        out[split + "/ignored"] = np.asarray(ignored, dtype=np.int64)
        for tag, kw in (("tok12000", dict(max_tokens=12000)), ("sent3", dict(max_sentences=3)),
                        ("tok16000_mult2", dict(max_tokens=16000, required_batch_size_multiple=2))):
            batches = ds.batch_by_size(idx, **kw)
            out["%s/batches/%s/sizes" % (split, tag)] = np.array([len(b) for b in batches])
            out["%s/batches/%s/flat" % (split, tag)] = np.concatenate([np.asarray(b) for b in batches])
        batches = ds.batch_by_size(idx, max_tokens=12000)
        for num_shards in (1, 2):
            for shard in range(num_shards):
                it = iterators.EpochBatchIterator(ds, ds.collater, batches, seed=1, num_shards=num_shards, shard_id=shard, epoch=1)
                for ep in (1, 2):
                    itr = it.next_epoch_itr(shuffle=train)
                    ids = [s["id"].numpy() if len(s) else np.zeros(0, dtype=np.int64) for s in itr]
                    out["%s/epoch%d/shards%d/%d/sizes" % (split, ep, num_shards, shard)] = np.array([len(i) for i in ids])
                    out["%s/epoch%d/shards%d/%d/ids" % (split, ep, num_shards, shard)] = np.concatenate(ids) if ids else np.zeros(0)
        sample = ds.collater([ds[int(i)] for i in batches[0]])
        out[split + "/sample/batch"] = np.asarray(batches[0])
        for k in ("id", "target", "target_lengths", "src_text", "src_text_lengths"):
            out[split + "/sample/" + k] = sample[k].numpy()
        for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
            out[split + "/sample/net_input/" + k] = sample["net_input"][k].numpy()
        out[split + "/sample/ntokens"] = np.array(sample["ntokens"])
        out[split + "/sample/mask"] = np.array(bool(sample["net_input"]["mask"]))
        out[split + "/sizes"] = np.array([ds.size(i) for i in range(len(ds))])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
