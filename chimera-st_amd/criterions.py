"""Mirror of fairseq/criterions/label_smoothed_cross_entropy.py (:33-160) and
fairseq/criterions/triplet_st_mt_contrastive.py (:18-212).  The log-softmax + NLL + smoothing runs in the fused
cst_ls_ce kernels (logits read once each way; no fp32 [B*U,V] log-prob tensor)."""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as CF
from .distributed import notify_unused_parameters
from .registry import register_criterion


class FairseqCriterion(nn.Module):
    def __init__(self, task):
        super().__init__()
        self.task = task
        self.padding_idx = task.target_dictionary.pad() if hasattr(task, "target_dictionary") and task.target_dictionary else -100

    @staticmethod
    def add_args(parser):
        pass

    @classmethod
    def build_criterion(cls, args, task):
        raise NotImplementedError

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return False


@register_criterion("label_smoothed_cross_entropy")
class LabelSmoothedCrossEntropyCriterion(FairseqCriterion):
    single_pass = True  # one forward pass per sample: every parameter receives one gradient per backward pass (trainer.py)

    def __init__(self, task, sentence_avg, label_smoothing, ignore_prefix_size=0, report_accuracy=False):
        super().__init__(task)
        self.sentence_avg = sentence_avg
        self.eps = label_smoothing
        self.ignore_prefix_size = ignore_prefix_size
        self.report_accuracy = report_accuracy
        assert ignore_prefix_size == 0

    @staticmethod
    def add_args(parser):
        parser.add_argument("--label-smoothing", default=0.0, type=float, metavar="D")
        parser.add_argument("--report-accuracy", action="store_true")
        parser.add_argument("--ignore-prefix-size", default=0, type=int)

    @classmethod
    def build_criterion(cls, args, task):
        return cls(task, getattr(args, "sentence_avg", False), args.label_smoothing,
                   getattr(args, "ignore_prefix_size", 0), getattr(args, "report_accuracy", False))

    def forward(self, model, sample, reduce=True):
        """:56-86.  Note Q6: net_input carries the collater's `mask`; the models here swallow it."""
        src = sample["net_input"].get("src_tokens")
        if torch.is_tensor(src) and not src.dtype.is_floating_point and model.training and torch.is_grad_enabled():
            # a text-only update (the MT pre-training recipe, chimera/scripts/train-en2any-MT.sh:38-60, runs this criterion over
            # token ids on the Chimera arch): the audio front end takes no part, its ~96 M parameters never fire a gradient hook and
            # are all-reduced as zeros (legacy_distributed_data_parallel.py:155-156).  Reported here — where it is known that NO
            # pass of this micro-batch is an audio pass — their buckets leave during backward instead of at finish().
            enc = getattr(model, "encoder", None)
            unused = [p for name in ("wav2vec_model", "subsample") if getattr(enc, name, None) is not None
                      for p in getattr(enc, name).parameters()]
            if unused:
                notify_unused_parameters(unused)
        net_output = model(**sample["net_input"])
        loss, nll_loss = self.compute_loss(model, net_output, sample, reduce=reduce)
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        logging_output = {"loss": loss.data, "nll_loss": nll_loss.data, "ntokens": sample["ntokens"],
                          "nsentences": sample["target"].size(0), "sample_size": sample_size}
        return loss, sample_size, logging_output

    def compute_loss(self, model, net_output, sample, reduce=True):
        assert reduce, "reduce=False (kd_ratio path) is not on the chimera/scripts path"
        logits = net_output[0]
        target = model.get_targets(sample, net_output)
        return CF.label_smoothed_nll_loss(logits, target, self.eps, self.padding_idx)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True

    @staticmethod
    def logging_keys():
        """The keys of logging_output, known without running a batch: a rank that runs out of memory in its FIRST update still has to
        build the all-reduced statistics vector in the other ranks' layout (trainer.py:564-570)."""
        return ("loss", "nll_loss", "ntokens", "nsentences", "sample_size")


@register_criterion("triplet_st_mt_contrastive")
class TripletSTMTContrastiveCriterion(LabelSmoothedCrossEntropyCriterion):
    single_pass = False  # an audio pass and a text pass per sample: the shared layers receive two gradients per backward pass

    def __init__(self, task, sentence_avg, label_smoothing, loss_ratio, contrastive_temp=0.1, ignore_prefix_size=0,
                 report_accuracy=False, contrastive_increase_until=None, kd_ratio=None):
        super().__init__(task, sentence_avg, label_smoothing, ignore_prefix_size, report_accuracy)
        self.loss_ratio = list(loss_ratio)
        self.contrastive_temp = contrastive_temp
        self.contrastive_increase_until = contrastive_increase_until
        self.kd_ratio = kd_ratio if kd_ratio is not None else [None, None]
        self.num_updates = 0
        assert list(self.kd_ratio) == [None, None], "kd_ratio is not used by chimera/scripts"

    @staticmethod
    def add_args(parser):
        """:47-66."""
        parser.add_argument("--label-smoothing", default=0.0, type=float, metavar="D")
        parser.add_argument("--report-accuracy", action="store_true")
        parser.add_argument("--ignore-prefix-size", default=0, type=int)
        parser.add_argument("--loss-ratio", default=[1, 1, 1], type=float, nargs=3)
        parser.add_argument("--contrastive-temp", default=0.1, type=float)
        parser.add_argument("--contrastive-increase-until", type=int, default=None)
        parser.add_argument("--kd-ratio", default=[None, None], type=float, nargs=2)

    @classmethod
    def build_criterion(cls, args, task):
        return cls(task, getattr(args, "sentence_avg", False), args.label_smoothing, args.loss_ratio,
                   args.contrastive_temp, getattr(args, "ignore_prefix_size", 0), getattr(args, "report_accuracy", False),
                   args.contrastive_increase_until, list(args.kd_ratio))

    def set_num_updates(self, n):
        """The reference reads metrics.get_smoothed_values("train")["num_updates"] (:43-45); the trainer pushes it here."""
        self.num_updates = n

    @staticmethod
    def one_decoder_pass(model):
        """Can the audio pass and the text pass share ONE decoder call?  Both feed the decoder the same prev_output_tokens and an
        M-slot memory of the same shape (w2v2_transformer_interlingua.py:137-146, :301-304: no padding in the memory), and every
        decoder row depends on its own batch element only — so decoder(cat(prev, prev), cat(memory_audio, memory_text)) yields the
        two passes' logits as its two batch halves: half the decoder launches at twice the rows, and every decoder parameter
        receives one gradient per backward pass instead of two."""
        import os
        return (not os.environ.get("CST_NO_PAIR_DECODER") and hasattr(model, "encoder") and hasattr(model, "decoder")
                and hasattr(model, "forward_with_internal") and not getattr(getattr(model, "encoder", None), "no_interlingua", True))

    @staticmethod
    def twice_used(model):
        """The parameters BOTH passes of an update run through when the decoder is shared (one_decoder_pass): the encoder layers behind
        the modality-specific front ends and the memory module.  They receive two gradients per backward pass (trainer.py marks them:
        their reductions are not deferred); everything else — wav2vec2, subsampler, text embedding, decoder — receives one."""
        if not TripletSTMTContrastiveCriterion.one_decoder_pass(model):
            return None
        import os
        if not hasattr(model.encoder, "forward_pair") or os.environ.get("CST_NO_PAIR_ENCODER") or os.environ.get("CST_NO_PACK"):
            return [p for n, p in model.encoder.named_parameters() if not n.startswith(("wav2vec_model.", "subsample.", "text_embed_tokens."))]
        if os.environ.get("CST_NO_PAIR_MEMORY"):  # forward_pair walks the shared encoder layers once, the memory layers per modality
            return [p for n, p in model.encoder.named_parameters() if n.startswith(("interlingua_layers.", "interlingua_embedding."))]
        return []  # ... and, by default, the memory layers once as well: every parameter receives one gradient per backward pass

    def _two_passes_one_decoder(self, model, sample, reduce):
        from .fairseq_model import EncoderOut
        from .modules import to_batch_major, to_time_major_view
        ni = sample["net_input"]
        import os
        if hasattr(model.encoder, "forward_pair") and not os.environ.get("CST_NO_PAIR_ENCODER") and not os.environ.get("CST_NO_PACK"):
            # ... and ONE walk through the shared encoder layers (S2T_W2V2_TransformerInterlinguaEncoder.forward_pair)
            enc_a, enc_t, enc_both = model.encoder.forward_pair(ni["src_tokens"], ni["src_lengths"], sample["src_text"], sample["src_text_lengths"])
        else:
            enc_a = model.encoder(src_tokens=ni["src_tokens"], src_lengths=ni["src_lengths"])
            enc_t = model.encoder(src_tokens=sample["src_text"], src_lengths=sample["src_text_lengths"])
            enc_both = None
        if enc_both is None:
            mem = to_time_major_view(torch.cat((to_batch_major(enc_a.encoder_out), to_batch_major(enc_t.encoder_out)), 0))  # [M, 2B, C]
            pm = torch.cat((enc_a.encoder_padding_mask, enc_t.encoder_padding_mask), 0)
            enc_both = EncoderOut(encoder_out=mem, encoder_padding_mask=pm, encoder_embedding=None, encoder_states=None, src_tokens=None,
                                  src_lengths=None)
        prev = ni["prev_output_tokens"]
        logits, extra = model.decoder(prev_output_tokens=torch.cat((prev, prev), 0), encoder_out=enc_both)
        B = prev.size(0)
        st_loss, st_nll_loss = self.compute_loss(model, (logits[:B], extra), sample, reduce=reduce)
        mt_loss, mt_nll_loss = self.compute_loss(model, (logits[B:], extra), sample, reduce=reduce)
        return st_loss, st_nll_loss, mt_loss, mt_nll_loss, enc_a.encoder_out, enc_t.encoder_out

    def forward(self, model, sample, reduce=True):
        """:68-146 — audio pass, text pass, contrastive term."""
        if self.loss_ratio[1] != 0 and self.one_decoder_pass(model):
            st_loss, st_nll_loss, mt_loss, mt_nll_loss, audio_internal, text_internal = self._two_passes_one_decoder(model, sample, reduce)
        else:
            st_net_output, audio_internal = model.forward_with_internal(**sample["net_input"])
            st_loss, st_nll_loss = self.compute_loss(model, st_net_output, sample, reduce=reduce)
            if self.loss_ratio[1] != 0:
                mt_input = {"src_tokens": sample["src_text"], "src_lengths": sample["src_text_lengths"],
                            "prev_output_tokens": sample["net_input"]["prev_output_tokens"], "mask": sample["net_input"]["mask"]}
                mt_net_output, text_internal = model.forward_with_internal(**mt_input)
                mt_loss, mt_nll_loss = self.compute_loss(model, mt_net_output, sample, reduce=reduce)
            else:
                mt_loss, mt_nll_loss = 0, 0
        if self.loss_ratio[2] != 0:
            contrastive_loss = self.compute_contrastive(audio_internal, text_internal, reduce)
        else:
            contrastive_loss = 0
        loss_ratio = deepcopy(self.loss_ratio)
        if self.contrastive_increase_until is not None:
            loss_ratio[2] *= min(1, (self.num_updates or 0) / self.contrastive_increase_until)
        loss = sum(loss_ratio[i] * l for i, l in enumerate((st_loss, mt_loss, contrastive_loss)))
        nll_loss = sum(loss_ratio[i] * l for i, l in enumerate((st_nll_loss, mt_nll_loss)))
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        logging_output = {
            "loss": loss.data, "nll_loss": nll_loss.data, "st_loss": st_loss.data, "st_nll_loss": st_nll_loss.data,
            "mt_loss": mt_loss.data if self.loss_ratio[1] != 0 else 0,
            "mt_nll_loss": mt_nll_loss.data if self.loss_ratio[1] != 0 else 0,
            "contrastive_loss": contrastive_loss.data if self.loss_ratio[2] != 0 else 0,
            "ntokens": sample["ntokens"], "nsentences": sample["target"].size(0), "sample_size": sample_size,
        }
        return loss, sample_size, logging_output

    def compute_contrastive(self, input1, input2, reduce):
        """:154-169 — per-utterance M x M cosine-similarity CE (class dim = audio slot): cst_contrastive_fwd/bwd."""
        assert input1.shape == input2.shape
        if not reduce:
            raise NotImplementedError("compute_contrastive(reduce=False) is not on the training path")
        from . import functional as CF
        return CF.contrastive_loss(input1.transpose(0, 1), input2.transpose(0, 1), self.contrastive_temp)

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True

    @staticmethod
    def logging_keys():
        return ("loss", "nll_loss", "st_loss", "st_nll_loss", "mt_loss", "mt_nll_loss", "contrastive_loss", "ntokens", "nsentences",
                "sample_size")
