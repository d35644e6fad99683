"""Input pipeline against fixtures produced by the REAL reference (tools/ref_harness/make_data_goldens.py):
manifest parsing, WAV slice decoding (native cst_wav_read_f32), tokenisation, ordering, size filtering, the native
cst_batch_by_size (vs the reference's Cython batch_by_size_fast), epoch shuffling + sharding, and the collater's sample dict."""
import os
import shutil
from argparse import Namespace
from importlib import import_module

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, load_pkg

DATA = os.path.join(GOLDEN, "data_tiny")


@pytest.fixture(scope="module")
def task(tmp_path_factory):
    import __graft_entry__ as ge
    load_pkg()
    lib = import_module("chimera-st_amd.lib")
    if not os.path.exists(lib.LIB_PATH):
        ge.build()
    root = tmp_path_factory.mktemp("data_root")
    for f in os.listdir(DATA):
        if not f.endswith(".wav"):
            shutil.copy(os.path.join(DATA, f), root / f)
    cfg = (root / "config_wave.yaml").read_text().replace("AUDIO_ROOT", DATA)  # the committed YAML is relocatable
    (root / "config_wave.yaml").write_text(cfg)
    tasks = import_module("chimera-st_amd.tasks")
    t = tasks.TripletTask(Namespace(data=str(root), config_yaml="config_wave.yaml", seed=1, max_source_positions=6000, max_target_positions=7))
    return t


def test_dictionaries_and_config(task):
    g = load_golden("data_tiny.npz")
    assert len(task.target_dictionary) == int(g["dict_len"]) == len(task.source_dictionary)
    assert task.data_cfg.use_audio_input and task.data_cfg.vocab_filename == "dict.txt"


@pytest.mark.parametrize("split", ["train_st", "dev_st"])
def test_dataset_matches_reference(task, split):
    D = import_module("chimera-st_amd.data")
    g = load_golden("data_tiny.npz")
    ds = task.load_dataset(split)
    assert np.array_equal(np.array([ds.size(i) for i in range(len(ds))]), g[split + "/sizes"])
    with D.numpy_seed(1):
        idx = ds.ordered_indices()
    assert np.array_equal(idx, g[split + "/ordered"])
    kept, ignored = ds.filter_indices_by_size(idx, task.max_positions())
    assert np.array_equal(kept, g[split + "/filtered"]) and ignored == g[split + "/ignored"].tolist()
    for tag, kw in (("tok12000", dict(max_tokens=12000)), ("sent3", dict(max_sentences=3)),
                    ("tok16000_mult2", dict(max_tokens=16000, required_batch_size_multiple=2))):
        batches = ds.batch_by_size(idx, **kw)
        assert [len(b) for b in batches] == g["%s/batches/%s/sizes" % (split, tag)].tolist(), tag
        assert np.concatenate([np.asarray(b) for b in batches]).tolist() == g["%s/batches/%s/flat" % (split, tag)].tolist()
    # the collater's sample dict, bit for bit (waveform decoded natively from "<wav>:<offset>:<length>")
    batch = g[split + "/sample/batch"].tolist()
    s = ds.collater([ds[i] for i in batch])
    assert list(s.keys()) == ["id", "net_input", "target", "target_lengths", "src_text", "src_text_lengths", "ntokens", "nsentences"]
    for k in ("id", "target", "target_lengths", "src_text", "src_text_lengths"):
        assert np.array_equal(s[k].numpy(), g[split + "/sample/" + k]), k
    for k in ("src_tokens", "src_lengths", "prev_output_tokens"):
        got = s["net_input"][k].numpy()
        assert got.dtype == g[split + "/sample/net_input/" + k].dtype and np.array_equal(got, g[split + "/sample/net_input/" + k]), k
    assert s["ntokens"] == int(g[split + "/sample/ntokens"]) and s["net_input"]["mask"] == bool(g[split + "/sample/mask"])
    assert s["nsentences"] == len(batch)


@pytest.mark.parametrize("split,shuffle", [("train_st", True), ("dev_st", False)])
def test_epoch_iterator_shuffle_and_shards(task, split, shuffle):
    g = load_golden("data_tiny.npz")
    ds = task.load_dataset(split)
    for num_shards in (1, 2):
        for shard in range(num_shards):
            it = task.get_batch_iterator(ds, max_tokens=12000, seed=1, num_shards=num_shards, shard_id=shard, epoch=1)
            for ep in (1, 2):
                ids = [s["id"].numpy() if s else np.zeros(0, dtype=np.int64) for s in it.next_epoch_itr(shuffle=shuffle)]
                key = "%s/epoch%d/shards%d/%d/" % (split, ep, num_shards, shard)
                assert [len(i) for i in ids] == g[key + "sizes"].tolist(), key
                assert np.concatenate(ids).tolist() == g[key + "ids"].tolist(), key


def test_buffered_iterator_equals_plain(task):
    ds = task.load_dataset("train_st")
    it = task.get_batch_iterator(ds, max_tokens=12000, seed=1, epoch=1)
    plain = [s["id"].tolist() for s in it.next_epoch_itr(shuffle=True)]
    it = task.get_batch_iterator(ds, max_tokens=12000, seed=1, epoch=1)
    buffered = [s["id"].tolist() for s in it.next_epoch_itr(shuffle=True, buffer_size=2)]
    assert plain == buffered and len(plain) > 1

    class Boom:
        def collater(self, x):
            raise ValueError("decode failed")

        def __getitem__(self, i):
            return i
    D = import_module("chimera-st_amd.data")
    bad = D.EpochBatchIterator(Boom(), [[0], [1]])
    with pytest.raises(ValueError, match="decode failed"):
        list(bad.next_epoch_itr(shuffle=False, buffer_size=2))


def test_batch_by_size_native_properties():
    """Edge cases of the native batcher: empty input, one oversized sample (error), multiples."""
    D = import_module("chimera-st_amd.data")
    assert D.batch_by_size(np.zeros(0, dtype=np.int64), lambda i: 1, max_tokens=10) == []
    with pytest.raises(AssertionError, match="exceeds max_tokens"):
        D.batch_by_size(np.arange(3), lambda i: [5, 50, 5][i], max_tokens=20)
    sizes = [9, 9, 8, 8, 7, 3, 3, 2, 1, 1, 1]
    b = D.batch_by_size(np.arange(len(sizes)), lambda i: sizes[i], max_tokens=27, required_batch_size_multiple=2)
    assert [i for x in b for i in x] == list(range(len(sizes)))          # order preserved, nothing lost
    assert all(len(x) * max(sizes[i] for i in x) <= 27 for x in b)        # padded size within the budget
    rng = np.random.RandomState(0)
    big = np.sort(rng.randint(160000, 480000, size=5000))[::-1].copy()
    b = D.batch_by_size(np.arange(5000), lambda i: int(big[i]), max_tokens=480000 * 32, max_sentences=32, required_batch_size_multiple=8)
    assert sum(len(x) for x in b) == 5000 and max(len(x) for x in b) <= 32


def test_wav_reader_edges(tmp_path):
    import wave
    D = import_module("chimera-st_amd.data")
    x = (np.arange(-500, 500) * 60).astype("<i2")
    p = str(tmp_path / "t.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000); w.writeframes(x.tobytes())
    full, sr = D.get_waveform(p)
    assert sr == 8000 and np.array_equal(full, x.astype(np.float32) / 32768.0)
    part, _ = D.get_waveform(p, 990, 100)            # slice running past the end is truncated, like libsndfile
    assert np.array_equal(part, full[990:])
    assert len(D.get_waveform(p, 1000, 5)[0]) == 0   # empty slice
    assert np.array_equal(D.get_features_or_waveform("%s:10:20" % p, need_waveform=True, sample_rate=8000), full[10:30])
    with pytest.raises(FileNotFoundError):
        D.get_features_or_waveform(str(tmp_path / "missing.wav"), need_waveform=True)
    (tmp_path / "bad.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(ValueError):
        D.get_waveform(str(tmp_path / "bad.wav"))
