"""Input pipeline of the Chimera training path — mirror of
  fairseq/data/audio/speech_to_text_dataset.py  S2TDataConfig :30-115, get_features_or_waveform :165-204, _collate_frames :207-225,
                                                SpeechToTextDataset :228-420, SpeechToTextDatasetCreator :423-557
  fairseq/data/audio/triplet_dataset.py         TripletDataConfig :34-48, TripletDataset :51-244, TripletDatasetCreator :247-370
  fairseq/data/audio/audio_utils.py             get_waveform / get_waveform_chi :7-55 (soundfile -> cst_wav_read_f32)
  fairseq/data/data_utils.py                    collate_tokens :34-64, numpy_seed :115-128, _filter_by_size_dynamic :148-184,
                                                batch_by_size :276-337 (Cython batch_by_size_fast -> cst_batch_by_size)
  fairseq/data/iterators.py                     EpochBatchIterator._get_iterator_for_epoch :384-435, ShardedIterator :470-500
  fairseq/data/encoders/sentencepiece_bpe.py    SentencepieceBPE :20-47
Wire formats kept: the MuST-C TSV manifest (id, audio, n_frames, tgt_text, src_text, speaker[, src_lang, tgt_lang]) with
`audio` = "<wav path>[:<offset>:<length>]", the data config YAML of chimera/tools/hand-make-config.py, the fairseq dictionary
file, and the collater's `sample` dict.  Host-only code: decoded batches are handed to the trainer as CPU tensors (pinned when
asked) and moved to the GPU by it."""
import contextlib
import csv
import ctypes
import math
import os.path as op
import re
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import lib as L
from .tasks import collate_tokens


# --------------------------------------------------------------------------------------------------------------------
class S2TDataConfig:
    """speech_to_text_dataset.py:30-115."""

    def __init__(self, yaml_path):
        import yaml
        self.config = {}
        if op.isfile(yaml_path):
            with open(yaml_path) as f:
                self.config = yaml.load(f, Loader=yaml.FullLoader) or {}

    vocab_filename = property(lambda self: self.config.get("vocab_filename", "dict.txt"))
    shuffle = property(lambda self: self.config.get("shuffle", False))
    pre_tokenizer = property(lambda self: self.config.get("pre_tokenizer", {"tokenizer": None}))
    bpe_tokenizer = property(lambda self: self.config.get("bpe_tokenizer", {"bpe": None}))
    prepend_tgt_lang_tag = property(lambda self: self.config.get("prepend_tgt_lang_tag", False))
    input_feat_per_channel = property(lambda self: self.config.get("input_feat_per_channel", 80))
    input_channels = property(lambda self: self.config.get("input_channels", 1))
    sampling_alpha = property(lambda self: self.config.get("sampling_alpha", 1.0))
    use_audio_input = property(lambda self: self.config.get("use_audio_input", False))
    audio_root = property(lambda self: self.config.get("audio_root", ""))

    def get_feature_transforms(self, split, is_train):
        cur = (self.config.get("transforms") or {})
        t = cur.get(split)
        t = cur.get("_train") if t is None and is_train else t
        t = cur.get("_eval") if t is None and not is_train else t
        return cur.get("*") if t is None else t


class TripletDataConfig(S2TDataConfig):
    """triplet_dataset.py:34-48."""
    src_bpe_tokenizer = property(lambda self: self.config.get("src_bpe_tokenizer", {"bpe": None}))
    src_vocab_filename = property(lambda self: self.config.get("src_vocab_filename", "dict.txt"))


class SentencepieceBPE:
    """encoders/sentencepiece_bpe.py:20-47."""

    def __init__(self, sentencepiece_model):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor()
        self.sp.Load(sentencepiece_model)

    def encode(self, x: str) -> str:
        return " ".join(self.sp.EncodeAsPieces(x))

    def decode(self, x: str) -> str:
        return x.replace(" ", "").replace("▁", " ").strip()


def build_bpe(cfg: Dict):
    """encoders.build_bpe for the two values the Chimera configs use."""
    name = (cfg or {}).get("bpe")
    if name in (None, "none"):
        return None
    if name == "sentencepiece":
        return SentencepieceBPE(cfg["sentencepiece_model"])
    raise NotImplementedError("bpe tokenizer %r is not on the Chimera path (sentencepiece only)" % name)


def build_tokenizer(cfg: Dict):
    name = (cfg or {}).get("tokenizer")
    if name in (None, "none"):
        return None
    raise NotImplementedError("pre-tokenizer %r is not on the Chimera path" % name)


# --------------------------------------------------------------------------------------------------------------------
def get_waveform(path, offset=0, length=-1):
    """audio_utils.py:7-55 — (float32 waveform in [-1, 1), sample rate) of a 16-bit PCM WAV (slice), decoded natively."""
    lib = L.load()
    sr, ch, bits = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    frames = ctypes.c_int64()
    rc = lib.cst_wav_info(path.encode(), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(frames), ctypes.byref(bits))
    if rc != 0:
        raise ValueError("Unsupported audio file %s: %s" % (path, lib.cst_last_error().decode()))
    n = frames.value - offset if length < 0 else min(length, max(frames.value - offset, 0))
    out = np.empty(max(n, 0) * ch.value, dtype=np.float32)
    got = lib.cst_wav_read_f32(path.encode(), offset, n, out.ctypes.data_as(ctypes.c_void_p), out.size)
    if got < 0:
        raise ValueError("cannot read %s: %s" % (path, lib.cst_last_error().decode()))
    out = out[:got * ch.value]
    return (out if ch.value == 1 else out.reshape(-1, ch.value)), sr.value


def get_features_or_waveform(path: str, need_waveform=False, sample_rate=16000):
    """speech_to_text_dataset.py:165-204: "<.npy/.wav path>" or "<wav path>:<sample offset>:<sample count>"."""
    _path, *extra = path.split(":")
    if not op.exists(_path):
        raise FileNotFoundError("File not found: %s" % _path)
    if len(extra) == 0:
        if need_waveform:
            wave, sr = get_waveform(_path)
            return wave[::(sr // sample_rate)] if sr != sample_rate else wave
        if op.splitext(_path)[1] == ".npy":
            return np.load(_path)
        raise NotImplementedError("filter-bank extraction from audio (torchaudio/kaldi) is not on the Chimera wave path; "
                                  "precomputed .npy features or use_audio_input: true")
    if len(extra) == 2:
        if _path.endswith(".zip"):
            raise NotImplementedError("zip-packed feature archives are not on the Chimera wave path")
        assert need_waveform, "path %s and not need_waveform conflict" % _path
        wave, sr = get_waveform(_path, int(extra[0]), int(extra[1]))
        return wave[::(sr // sample_rate)] if sr != sample_rate else wave
    raise ValueError("Invalid path: %s" % path)


def _collate_frames(frames: List[torch.Tensor], is_audio_input: bool = False) -> torch.Tensor:
    """speech_to_text_dataset.py:207-225."""
    max_len = max(f.size(0) for f in frames)
    out = frames[0].new_zeros((len(frames), max_len) if is_audio_input else (len(frames), max_len, frames[0].size(1)))
    for i, v in enumerate(frames):
        out[i, :v.size(0)] = v
    return out


@contextlib.contextmanager
def numpy_seed(seed, *addl_seeds):
    """data_utils.py:115-128."""
    if seed is None:
        yield
        return
    if len(addl_seeds) > 0:
        seed = int(hash((seed, *addl_seeds)) % 1e6)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def batch_by_size(indices, num_tokens_fn, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
    """data_utils.py:276-337 over the native cst_batch_by_size (the reference's Cython batch_by_size_fast)."""
    indices = np.asarray(indices, dtype=np.int64)
    sizes = np.fromiter((num_tokens_fn(int(i)) for i in indices), dtype=np.int64, count=len(indices))
    out = np.empty(max(len(indices), 1), dtype=np.int64)
    lib = L.load()
    nb = lib.cst_batch_by_size(sizes.ctypes.data_as(ctypes.c_void_p), len(indices), -1 if max_tokens is None else int(max_tokens),
                               -1 if max_sentences is None else int(max_sentences), int(required_batch_size_multiple),
                               out.ctypes.data_as(ctypes.c_void_p))
    if nb < 0:
        raise AssertionError(lib.cst_last_error().decode())
    ends = np.cumsum(out[:nb])
    return [indices[e - n:e].tolist() for n, e in zip(out[:nb], ends)]


# --------------------------------------------------------------------------------------------------------------------
class SpeechToTextDataset(torch.utils.data.Dataset):
    """speech_to_text_dataset.py:228-420."""
    LANG_TAG_TEMPLATE = "<lang:{}>"

    def __init__(self, split, is_train_split, data_cfg, audio_paths, n_frames, src_texts=None, tgt_texts=None, speakers=None,
                 src_langs=None, tgt_langs=None, ids=None, tgt_dict=None, pre_tokenizer=None, bpe_tokenizer=None, normalize=False,
                 mask=True, sample_rate=16000):
        self.split, self.is_train_split = split, is_train_split
        self.data_cfg = data_cfg
        self.audio_paths, self.n_frames = audio_paths, n_frames
        self.n_samples = len(audio_paths)
        assert len(n_frames) == self.n_samples > 0
        for lst in (src_texts, tgt_texts, speakers, src_langs, tgt_langs, ids):
            assert lst is None or len(lst) == self.n_samples
        assert (tgt_dict is None and tgt_texts is None) or (tgt_dict is not None and tgt_texts is not None)
        self.src_texts, self.tgt_texts = src_texts, tgt_texts
        self.src_langs, self.tgt_langs = src_langs, tgt_langs
        self.tgt_dict = tgt_dict
        self.check_tgt_lang_tag()
        self.ids = ids
        self.shuffle = data_cfg.shuffle if is_train_split else False
        if data_cfg.get_feature_transforms(split, is_train_split):
            raise NotImplementedError("feature transforms (specaugment / cmvn on filter banks) are not on the wave-input path")
        self.pre_tokenizer, self.bpe_tokenizer = pre_tokenizer, bpe_tokenizer
        self.normalize, self.mask, self.sample_rate = normalize, mask, sample_rate

    @classmethod
    def is_lang_tag(cls, token):
        pattern = cls.LANG_TAG_TEMPLATE.replace("{}", "(.*)")
        return re.match(pattern, token)

    def check_tgt_lang_tag(self):
        if self.data_cfg.prepend_tgt_lang_tag:
            assert self.tgt_langs is not None and self.tgt_dict is not None
            tags = [self.LANG_TAG_TEMPLATE.format(t) for t in set(self.tgt_langs)]
            assert all(t in self.tgt_dict.indices for t in tags)

    def tokenize_text(self, text: str, side="target"):
        if self.pre_tokenizer is not None:
            text = self.pre_tokenizer.encode(text)
        if self.bpe_tokenizer is not None:
            text = self.bpe_tokenizer.encode(text)
        return text

    def _source(self, index):
        source = get_features_or_waveform(self.audio_paths[index], need_waveform=self.data_cfg.use_audio_input,
                                          sample_rate=self.sample_rate)
        source = torch.from_numpy(np.ascontiguousarray(source)).float()
        if self.normalize:
            with torch.no_grad():
                source = F.layer_norm(source, source.shape)
        return source

    def _target(self, index):
        if self.tgt_texts is None:
            return None
        tokenized = self.tokenize_text(self.tgt_texts[index], "target")
        target = self.tgt_dict.encode_line(tokenized, add_if_not_exist=False, append_eos=True).long()
        if self.data_cfg.prepend_tgt_lang_tag:
            tag = self.tgt_dict.index(self.LANG_TAG_TEMPLATE.format(self.tgt_langs[index]))
            target = torch.cat((torch.LongTensor([tag]), target), 0)
        return target

    def __getitem__(self, index):
        return index, self._source(index), self._target(index)

    def __len__(self):
        return self.n_samples

    def _collate_common(self, samples):
        indices = torch.tensor([s[0] for s in samples], dtype=torch.long)
        frames = _collate_frames([s[1] for s in samples], self.data_cfg.use_audio_input)
        n_frames = torch.tensor([s[1].size(0) for s in samples], dtype=torch.long)
        n_frames, order = n_frames.sort(descending=True)  # sort samples by descending number of frames
        indices, frames = indices.index_select(0, order), frames.index_select(0, order)
        target = target_lengths = prev_output_tokens = ntokens = None
        if self.tgt_texts is not None:
            tg = [s[2] for s in samples]
            target = collate_tokens(tg, self.tgt_dict.pad(), self.tgt_dict.eos()).index_select(0, order)
            target_lengths = torch.tensor([t.size(0) for t in tg], dtype=torch.long).index_select(0, order)
            prev_output_tokens = collate_tokens(tg, self.tgt_dict.pad(), self.tgt_dict.eos(),
                                                move_eos_to_beginning=True).index_select(0, order)
            ntokens = sum(t.size(0) for t in tg)
        out = {"id": indices,
               "net_input": {"src_tokens": frames, "src_lengths": n_frames, "prev_output_tokens": prev_output_tokens, "mask": self.mask},
               "target": target, "target_lengths": target_lengths, "ntokens": ntokens, "nsentences": len(samples)}
        return out, order

    def collater(self, samples):
        if len(samples) == 0:
            return {}
        return self._collate_common(samples)[0]

    def num_tokens(self, index):
        return self.n_frames[index]

    def size(self, index):
        t_len = 0
        if self.tgt_texts is not None:
            t_len = len(self.tokenize_text(self.tgt_texts[index]).split(" "))
        return self.n_frames[index], t_len

    @property
    def sizes(self):
        return np.array(self.n_frames)

    def ordered_indices(self):
        order = [np.random.permutation(len(self))] if self.shuffle else [np.arange(len(self))]
        order.append([-n for n in self.n_frames])  # first by descending number of frames, then by original / random order
        return np.lexsort(order)

    def filter_indices_by_size(self, indices, max_sizes):
        """fairseq_dataset.py:144-180 / data_utils._filter_by_size_dynamic :148-184."""
        if isinstance(max_sizes, (int, float)):
            keep = self.sizes[indices] <= max_sizes
            return indices[keep], indices[~keep].tolist()
        ok = np.fromiter((all(a is None or b is None or a <= b for a, b in zip(self.size(int(i)), max_sizes)) for i in indices),
                         dtype=bool, count=len(indices))
        return indices[ok], indices[~ok].tolist()

    def batch_by_size(self, indices, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
        return batch_by_size(indices, self.num_tokens, max_tokens, max_sentences, required_batch_size_multiple)


class TripletDataset(SpeechToTextDataset):
    """triplet_dataset.py:51-244 — (audio, source text, target text)."""

    def __init__(self, split, is_train_split, data_cfg, audio_paths, n_frames, src_texts=None, tgt_texts=None, speakers=None,
                 src_langs=None, tgt_langs=None, ids=None, tgt_dict=None, src_dict=None, pre_tokenizer=None, bpe_tokenizer=None,
                 src_bpe_tokenizer=None, normalize=False, mask=True, sample_rate=16000):
        super().__init__(split, is_train_split, data_cfg, audio_paths, n_frames, src_texts, tgt_texts, speakers, src_langs,
                         tgt_langs, ids, tgt_dict, pre_tokenizer, bpe_tokenizer, normalize, mask, sample_rate)
        assert (src_dict is None and src_texts is None) or (src_dict is not None and src_texts is not None)
        self.src_dict, self.src_bpe_tokenizer = src_dict, src_bpe_tokenizer

    def tokenize_text(self, text: str, side="target"):
        if self.pre_tokenizer is not None:
            text = self.pre_tokenizer.encode(text)
        tok = self.bpe_tokenizer if side == "target" else (self.src_bpe_tokenizer if side == "source" else None)
        return tok.encode(text) if tok is not None else text

    def __getitem__(self, index):
        src_text = None
        if self.src_texts is not None:
            src_text = self.src_dict.encode_line(self.tokenize_text(self.src_texts[index], "source"), add_if_not_exist=False,
                                                 append_eos=True).long()
        return index, self._source(index), self._target(index), src_text

    def collater(self, samples):
        if len(samples) == 0:
            return {}
        out, order = self._collate_common(samples)
        src_text = src_text_lengths = None
        if self.src_texts is not None:
            st = [s[3] for s in samples]
            src_text = collate_tokens(st, self.src_dict.pad(), self.src_dict.eos()).index_select(0, order)
            src_text_lengths = torch.tensor([s.size(0) for s in st], dtype=torch.long).index_select(0, order)
        # key order of triplet_dataset.py:220-234
        return {"id": out["id"], "net_input": out["net_input"], "target": out["target"], "target_lengths": out["target_lengths"],
                "src_text": src_text, "src_text_lengths": src_text_lengths, "ntokens": out["ntokens"], "nsentences": out["nsentences"]}


class ConcatDataset(torch.utils.data.Dataset):
    """fairseq/data/concat_dataset.py:14-125 (sample_ratios = 1).  The reference wraps EVERY manifest list in one — also a
    single split (triplet_dataset.py:370) — so its batching order is this class's: ascending by size (np.argsort(sizes), :87-110;
    the collater re-sorts each batch by descending length), and num_tokens = max(size) (:68-69)."""

    def __init__(self, datasets):
        assert len(datasets) > 0, "datasets should not be an empty iterable"
        self.datasets = list(datasets)
        self.cum = np.cumsum([len(d) for d in self.datasets])

    def _loc(self, idx):
        k = int(np.searchsorted(self.cum, idx, side="right"))
        return k, int(idx - (self.cum[k - 1] if k > 0 else 0))

    def __len__(self):
        return int(self.cum[-1])

    def __getitem__(self, idx):
        k, j = self._loc(idx)
        return self.datasets[k][j]  # the item carries the index WITHIN its manifest (concat_dataset.py:37-39)

    def collater(self, samples):
        return self.datasets[0].collater(samples)

    def size(self, idx):
        k, j = self._loc(idx)
        return self.datasets[k].size(j)

    def num_tokens(self, idx):
        return int(np.max(self.size(idx)))

    @property
    def sizes(self):
        return np.concatenate([d.sizes for d in self.datasets])

    def ordered_indices(self):
        return np.argsort(self.sizes)

    def filter_indices_by_size(self, indices, max_sizes):
        return SpeechToTextDataset.filter_indices_by_size(self, indices, max_sizes)

    def batch_by_size(self, indices, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
        return batch_by_size(indices, self.num_tokens, max_tokens, max_sentences, required_batch_size_multiple)


class TripletDatasetCreator:
    """speech_to_text_dataset.py:423-557 / triplet_dataset.py:247-370: datasets from MuST-C style TSV manifests."""
    KEY_ID, KEY_AUDIO, KEY_N_FRAMES, KEY_TGT_TEXT = "id", "audio", "n_frames", "tgt_text"
    KEY_SPEAKER, KEY_SRC_TEXT, KEY_SRC_LANG, KEY_TGT_LANG = "speaker", "src_text", "src_lang", "tgt_lang"
    DEFAULT_SPEAKER = DEFAULT_SRC_TEXT = DEFAULT_LANG = ""

    @staticmethod
    def read_tsv(path):
        if not op.isfile(path):
            raise FileNotFoundError("Dataset not found: %s" % path)
        with open(path) as f:
            reader = csv.DictReader(f, delimiter="\t", quotechar=None, doublequote=False, lineterminator="\n", quoting=csv.QUOTE_NONE)
            return [dict(e) for e in reader]

    @classmethod
    def _from_list(cls, split_name, is_train_split, samples, data_cfg, tgt_dict, src_dict, pre_tokenizer, bpe_tokenizer,
                   src_bpe_tokenizer, normalize, mask, sample_rate, triplet=True):
        g = lambda key, default=None: [s[key] if default is None else s.get(key, default) for s in samples]
        audio_paths = [op.join(data_cfg.audio_root, s[cls.KEY_AUDIO]) for s in samples]
        n_frames = [int(s[cls.KEY_N_FRAMES]) for s in samples]
        common = dict(speakers=g(cls.KEY_SPEAKER, cls.DEFAULT_SPEAKER), src_langs=g(cls.KEY_SRC_LANG, cls.DEFAULT_LANG),
                      tgt_langs=g(cls.KEY_TGT_LANG, cls.DEFAULT_LANG), ids=g(cls.KEY_ID), tgt_dict=tgt_dict,
                      pre_tokenizer=pre_tokenizer, bpe_tokenizer=bpe_tokenizer, normalize=normalize, mask=mask, sample_rate=sample_rate)
        if triplet:
            return TripletDataset(split_name, is_train_split, data_cfg, audio_paths, n_frames, g(cls.KEY_SRC_TEXT, cls.DEFAULT_SRC_TEXT),
                                  g(cls.KEY_TGT_TEXT), src_dict=src_dict, src_bpe_tokenizer=src_bpe_tokenizer, **common)
        return SpeechToTextDataset(split_name, is_train_split, data_cfg, audio_paths, n_frames, g(cls.KEY_SRC_TEXT, cls.DEFAULT_SRC_TEXT),
                                   g(cls.KEY_TGT_TEXT), **common)

    @classmethod
    def from_tsv(cls, root, data_cfg, splits, tgt_dict, src_dict, pre_tokenizer, bpe_tokenizer, src_bpe_tokenizer, is_train_split,
                 epoch=1, seed=1, normalize=False, mask=True, sample_rate=16000, triplet=True):
        names = splits.split(",")
        datasets = [cls._from_list(n, is_train_split, cls.read_tsv(op.join(root, n + ".tsv")), data_cfg, tgt_dict, src_dict,
                                   pre_tokenizer, bpe_tokenizer, src_bpe_tokenizer, normalize, mask, sample_rate, triplet) for n in names]
        if is_train_split and len(names) > 1 and data_cfg.sampling_alpha != 1.0:
            raise NotImplementedError("temperature-based resampling of several train manifests (sampling_alpha != 1)")
        return ConcatDataset(datasets)


# --------------------------------------------------------------------------------------------------------------------
class EpochBatchIterator:
    """iterators.py:200-435, the part the trainer consumes: frozen batches -> per-epoch shuffle (numpy_seed(seed + epoch)) ->
    this rank's shard (ShardedIterator :470-500: batches[shard_id::num_shards], padded with empty batches) -> collater."""

    def __init__(self, dataset, batch_sampler, seed=1, num_shards=1, shard_id=0, epoch=1, pin_memory=False):
        self.dataset, self.frozen_batches = dataset, [list(b) for b in batch_sampler]
        self.seed, self.num_shards, self.shard_id, self.epoch, self.pin_memory = seed, num_shards, shard_id, max(epoch, 1), pin_memory

    def __len__(self):
        return int(math.ceil(len(self.frozen_batches) / float(self.num_shards)))

    def epoch_batches(self, epoch, shuffle=True):
        batches = list(self.frozen_batches)
        if shuffle:
            with numpy_seed(self.seed + epoch):
                np.random.shuffle(batches)
        mine = batches[self.shard_id::self.num_shards]
        return mine + [[] for _ in range(len(self) - len(mine))]

    def next_epoch_itr(self, shuffle=True, buffer_size=0):
        """buffer_size > 0: batches are decoded / collated (and pinned) by a background thread that stays that many batches ahead
        of the consumer (iterators.py BufferedIterator :503-570): WAV decoding and padding overlap the GPU update."""
        batches = self.epoch_batches(self.epoch, shuffle)
        self.epoch += 1

        def gen():
            for b in batches:
                sample = self.dataset.collater([self.dataset[i] for i in b])
                if self.pin_memory and sample:
                    sample = _pin(sample)
                yield sample
        return _buffered(gen(), buffer_size) if buffer_size > 0 else gen()


def _buffered(source, size):
    import queue
    import threading
    q, done = queue.Queue(maxsize=size), object()

    def work():
        try:
            for item in source:
                q.put(item)
            q.put(done)
        except BaseException as e:  # surface loader errors in the consumer
            q.put(e)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is done:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


def _pin(x):
    if torch.is_tensor(x):
        return x.pin_memory() if torch.cuda.is_available() else x
    if isinstance(x, dict):
        return {k: _pin(v) for k, v in x.items()}
    return x


def get_batch_iterator(dataset, max_tokens=None, max_sentences=None, max_positions=None, ignore_invalid_inputs=False,
                       required_batch_size_multiple=1, seed=1, num_shards=1, shard_id=0, epoch=1, pin_memory=False):
    """tasks/fairseq_task.py:162-275."""
    with numpy_seed(seed):
        indices = dataset.ordered_indices()
    if max_positions is not None:
        indices, ignored = dataset.filter_indices_by_size(indices, max_positions)
        if len(ignored) > 0 and not ignore_invalid_inputs:
            raise Exception("Size of sample #{} is invalid (={}) since max_positions={}, skip this example with "
                            "--skip-invalid-size-inputs-valid-test".format(ignored[0], dataset.size(ignored[0]), max_positions))
    batch_sampler = dataset.batch_by_size(indices, max_tokens=max_tokens, max_sentences=max_sentences,
                                          required_batch_size_multiple=required_batch_size_multiple)
    return EpochBatchIterator(dataset, batch_sampler, seed=seed, num_shards=num_shards, shard_id=shard_id, epoch=epoch,
                              pin_memory=pin_memory)
