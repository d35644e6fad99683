"""Mirror of the task surface the hot path is called through: fairseq/tasks/fairseq_task.py (build_model :265-281,
build_criterion :283-298, train_step :414-445, valid_step :447-451, inference_step :453-459, build_generator
:309-412), fairseq/tasks/speech_to_text.py (:27-154) and fairseq/tasks/triplet.py (:23-241).

Datasets / TSV manifests / audio decoding are a "next" row; this build provides the collater's
`sample` dict layout (data/audio/triplet_dataset.py:220-234) through `synthetic_sample`, the same shapes the
reference batches have: audio sorted by length descending, right-padded; targets with eos; prev_output_tokens
= eos-shifted targets (data/data_utils.py:34-64)."""
import os

import torch

from . import registry
from .dictionary import Dictionary
from .registry import register_task
from .profiling import scope


def collate_tokens(values, pad_idx, eos_idx, left_pad=False, move_eos_to_beginning=False):
    """data/data_utils.py:34-64."""
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        dst = res[i][size - len(v):] if left_pad else res[i][:len(v)]
        if move_eos_to_beginning:
            dst[0] = eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res


class FairseqTask:
    def __init__(self, args):
        self.args = args

    @staticmethod
    def add_args(parser):
        pass

    @classmethod
    def setup_task(cls, args, **kwargs):
        return cls(args, **kwargs)

    @property
    def source_dictionary(self):
        return None

    @property
    def target_dictionary(self):
        return None

    def build_model(self, args):
        return registry.build_model(args, self)

    def build_criterion(self, args):
        return registry.build_criterion(args, self)

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """fairseq_task.py:414-445."""
        model.train()
        if hasattr(model, "set_num_updates"):
            model.set_num_updates(update_num)
        if hasattr(criterion, "set_num_updates"):
            criterion.set_num_updates(update_num)
        with scope("forward"):
            loss, sample_size, logging_output = criterion(model, sample)
        if ignore_grad:
            loss = loss * 0
        with scope("backward"):
            optimizer.backward(loss)
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output

    def inference_step(self, generator, models, sample, prefix_tokens=None, constraints=None):
        with torch.no_grad():
            return generator.generate(models, sample, prefix_tokens=prefix_tokens)

    # ---- datasets (data.py; tasks/fairseq_task.py:162-275) --------------------------------------------------------
    def dataset(self, split):
        if split not in getattr(self, "datasets", {}):
            raise KeyError("Dataset not loaded: " + split)
        return self.datasets[split]

    def max_positions(self):
        return getattr(self.args, "max_source_positions", 6000), getattr(self.args, "max_target_positions", 1024)

    def get_batch_iterator(self, dataset, max_tokens=None, max_sentences=None, max_positions=None, ignore_invalid_inputs=False,
                           required_batch_size_multiple=1, seed=1, num_shards=1, shard_id=0, num_workers=0, epoch=1, **unused):
        from . import data as D
        return D.get_batch_iterator(dataset, max_tokens, max_sentences, max_positions, ignore_invalid_inputs,
                                    required_batch_size_multiple, seed, num_shards, shard_id, epoch)

    def build_generator(self, models, args, seq_gen_cls=None, extra_gen_cls_kwargs=None):
        """fairseq_task.py:309-412, beam-search branch."""
        from .sequence_generator import SequenceGenerator

        return SequenceGenerator(
            models, self.target_dictionary, beam_size=getattr(args, "beam", 5), max_len_a=getattr(args, "max_len_a", 0),
            max_len_b=getattr(args, "max_len_b", 200), min_len=getattr(args, "min_len", 1),
            normalize_scores=(not getattr(args, "unnormalized", False)), len_penalty=getattr(args, "lenpen", 1),
            unk_penalty=getattr(args, "unkpen", 0), temperature=getattr(args, "temperature", 1.0))


def _load_dict(args, default_size):
    data = getattr(args, "data", None)
    if data:
        for name in (getattr(args, "dict_file", None), "spm_unigram10000_wave_joint.txt", "dict.txt"):
            if name and os.path.exists(os.path.join(data, name)):
                return Dictionary.load(os.path.join(data, name))
    return Dictionary.synthetic(getattr(args, "synthetic_vocab_size", default_size))


@register_task("speech_to_text")
class SpeechToTextTask(FairseqTask):
    """tasks/speech_to_text.py:27-154."""

    @staticmethod
    def add_args(parser):
        parser.add_argument("data", nargs="?", default=None, help="manifest root path")  # tasks/speech_to_text.py:27 (optional here: synthetic data)
        parser.add_argument("--config-yaml", type=str, default="config.yaml")
        parser.add_argument("--normalize", action="store_true")
        parser.add_argument("--max-source-positions", default=6000, type=int, metavar="N")
        parser.add_argument("--max-target-positions", default=1024, type=int, metavar="N")
        parser.add_argument("--synthetic-vocab-size", default=10000, type=int)

    TRIPLET = False

    def __init__(self, args, tgt_dict=None, data_cfg=None):
        super().__init__(args)
        self.data_cfg = data_cfg if data_cfg is not None else self._load_data_cfg(args)
        if tgt_dict is None and self.data_cfg is not None and getattr(args, "data", None) and \
                os.path.isfile(os.path.join(args.data, self.data_cfg.vocab_filename)):
            tgt_dict = Dictionary.load(os.path.join(args.data, self.data_cfg.vocab_filename))  # speech_to_text.py:60-70
        self.tgt_dict = tgt_dict if tgt_dict is not None else _load_dict(args, 10000)
        self.datasets = {}

    @staticmethod
    def _load_data_cfg(args):
        data, name = getattr(args, "data", None), getattr(args, "config_yaml", "config.yaml")
        if data and os.path.isfile(os.path.join(data, name)):
            from . import data as D
            return D.TripletDataConfig(os.path.join(data, name))
        return None

    def load_dataset(self, split, epoch=1, combine=False, **kwargs):
        """tasks/speech_to_text.py:95-111 / tasks/triplet.py:112-132: <data>/<split>.tsv through the data config YAML."""
        from . import data as D
        assert self.data_cfg is not None, "load_dataset needs --data <manifest root> with the --config-yaml file in it"
        self.datasets[split] = D.TripletDatasetCreator.from_tsv(
            self.args.data, self.data_cfg, split, self.tgt_dict, self.source_dictionary if self.TRIPLET else None,
            D.build_tokenizer(self.data_cfg.pre_tokenizer), D.build_bpe(self.data_cfg.bpe_tokenizer),
            D.build_bpe(self.data_cfg.src_bpe_tokenizer) if self.TRIPLET else None, is_train_split=split.startswith("train"),
            epoch=epoch, seed=getattr(self.args, "seed", 1), normalize=getattr(self.args, "normalize", False),
            sample_rate=getattr(self.args, "sample_rate", 16000), triplet=self.TRIPLET)
        return self.datasets[split]

    def build_model(self, args):
        if self.data_cfg is not None:  # speech_to_text.py:119-122
            args.input_feat_per_channel = self.data_cfg.input_feat_per_channel
            args.input_channels = self.data_cfg.input_channels
        return super().build_model(args)

    @property
    def target_dictionary(self):
        return self.tgt_dict

    @property
    def source_dictionary(self):
        return None


@register_task("triplet")
class TripletTask(SpeechToTextTask):
    """tasks/triplet.py:23-241 — audio + source text + target text triplets; joint dictionary."""

    @staticmethod
    def add_args(parser):
        SpeechToTextTask.add_args(parser)
        parser.add_argument("--dump-feature-to-file", type=str, default=None)
        parser.add_argument("--sample-rate", type=int, default=16000)

    TRIPLET = True

    def __init__(self, args, tgt_dict=None, src_dict=None, data_cfg=None):
        super().__init__(args, tgt_dict, data_cfg)
        if src_dict is None and self.data_cfg is not None and getattr(args, "data", None) and \
                os.path.isfile(os.path.join(args.data, self.data_cfg.src_vocab_filename)):
            src_dict = Dictionary.load(os.path.join(args.data, self.data_cfg.src_vocab_filename))  # triplet.py:80-95
        self.src_dict = src_dict if src_dict is not None else self.tgt_dict

    @property
    def source_dictionary(self):
        return self.src_dict


def synthetic_sample(dictionary, batch_size, audio_lengths, target_lengths, src_text_lengths=None, seed=1, device="cpu",
                     sort=True):
    """A collater-shaped batch (triplet_dataset.py:165-235) of seeded synthetic data (SURVEY §8d "Synthetic inputs")."""
    g = torch.Generator().manual_seed(seed)
    V, pad, eos = len(dictionary), dictionary.pad(), dictionary.eos()
    order = sorted(range(batch_size), key=lambda i: -audio_lengths[i]) if sort else list(range(batch_size))
    audio_lengths = [audio_lengths[i] for i in order]
    target_lengths = [target_lengths[i] for i in order]
    smax = max(audio_lengths)
    audio = torch.zeros(batch_size, smax)
    for i, s in enumerate(audio_lengths):
        audio[i, :s] = 0.1 * torch.randn(s, generator=g)
    tgt = [torch.cat([torch.randint(4, V, (u,), generator=g), torch.tensor([eos])]) for u in target_lengths]
    sample = {
        "id": torch.arange(batch_size),
        "net_input": {
            "src_tokens": audio.to(device),
            "src_lengths": torch.tensor(audio_lengths, dtype=torch.long, device=device),
            "prev_output_tokens": collate_tokens(tgt, pad, eos, move_eos_to_beginning=True).to(device),
            "mask": False,
        },
        "target": collate_tokens(tgt, pad, eos).to(device),
        "target_lengths": torch.tensor([len(t) for t in tgt], device=device),
        "ntokens": int(sum(len(t) for t in tgt)),
        "nsentences": batch_size,
    }
    if src_text_lengths is not None:
        src_text_lengths = [src_text_lengths[i] for i in order]
        src = [torch.cat([torch.randint(4, V, (l,), generator=g), torch.tensor([eos])]) for l in src_text_lengths]
        sample["src_text"] = collate_tokens(src, pad, eos).to(device)
        sample["src_text_lengths"] = torch.tensor([len(s) for s in src], device=device)
    return sample
