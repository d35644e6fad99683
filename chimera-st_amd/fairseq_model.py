"""Base classes of the reference's model API that the hot path's callers rely on
(fairseq/models/fairseq_encoder.py:13-23 EncoderOut; fairseq_model.py:286 FairseqEncoderDecoderModel;
fairseq_decoder.py:58-79 get_normalized_probs; fairseq_incremental_decoder.py reorder hooks)."""
from typing import Dict, List, NamedTuple, Optional

import torch
import torch.nn as nn
from torch import Tensor

EncoderOut = NamedTuple(
    "EncoderOut",
    [
        ("encoder_out", Tensor),  # T x B x C
        ("encoder_padding_mask", Optional[Tensor]),  # B x T
        ("encoder_embedding", Optional[Tensor]),  # B x T x C
        ("encoder_states", Optional[List[Tensor]]),  # List[T x B x C]
        ("src_tokens", Optional[Tensor]),  # B x T
        ("src_lengths", Optional[Tensor]),  # B x 1
    ],
)


def lengths_to_padding_mask(lens: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """fairseq/data/data_utils.py:491-495."""
    bsz = lens.size(0)
    m = int(lens.max().item()) if max_len is None else max_len
    mask = torch.arange(m, device=lens.device).view(1, m).expand(bsz, -1) >= lens.view(bsz, 1).expand(-1, m)
    return mask


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary

    def forward_torchscript(self, net_input: Dict[str, Tensor]):
        """fairseq_encoder.py:43-62: everything in net_input except prev_output_tokens is passed to forward()."""
        encoder_input = {k: v for k, v in net_input.items() if k != "prev_output_tokens"}
        return self.forward(**encoder_input)

    def reorder_encoder_out(self, encoder_out, new_order):
        raise NotImplementedError

    def max_positions(self):
        return 1e6

    def upgrade_state_dict_named(self, state_dict, name):
        return state_dict


class FairseqDecoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary
        self.onnx_trace = False

    def get_normalized_probs(self, net_output, log_probs: bool, sample=None):
        """fairseq_decoder.py:58-79 -> utils.log_softmax in fp32 (utils.py:469-473)."""
        logits = net_output[0]
        if log_probs:
            return torch.log_softmax(logits.float(), dim=-1)
        return torch.softmax(logits.float(), dim=-1)

    def max_positions(self):
        return 1e6

    def upgrade_state_dict_named(self, state_dict, name):
        return state_dict


class FairseqIncrementalDecoder(FairseqDecoder):
    def reorder_incremental_state(self, incremental_state, new_order):
        pass

    def reorder_incremental_state_scripting(self, incremental_state, new_order):
        """fairseq_incremental_decoder.py:89-104: every sub-module that caches state reorders it."""
        for module in self.modules():
            if hasattr(module, "reorder_incremental_state") and module is not self:
                result = module.reorder_incremental_state(incremental_state, new_order)
                if result is not None:
                    incremental_state = result


class BaseFairseqModel(nn.Module):
    @staticmethod
    def add_args(parser):
        pass

    @classmethod
    def build_model(cls, args, task):
        raise NotImplementedError

    def get_targets(self, sample, net_output):
        return sample["target"]

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        if hasattr(self, "decoder"):
            return self.decoder.get_normalized_probs(net_output, log_probs, sample)
        raise NotImplementedError

    def max_positions(self):
        return None

    def upgrade_state_dict(self, state_dict):
        self.upgrade_state_dict_named(state_dict, "")

    def upgrade_state_dict_named(self, state_dict, name):
        """fairseq_model.py:117-141: recurse over children."""

        def do_upgrade(m, prefix):
            if len(prefix) > 0:
                prefix += "."
            for n, c in m.named_children():
                name = prefix + n
                if hasattr(c, "upgrade_state_dict_named"):
                    c.upgrade_state_dict_named(state_dict, name)
                elif hasattr(c, "upgrade_state_dict"):
                    c.upgrade_state_dict(state_dict)
                do_upgrade(c, name)

        do_upgrade(self, name)

    def set_num_updates(self, num_updates):
        pass


class FairseqEncoderDecoderModel(BaseFairseqModel):
    """fairseq_model.py:286-356."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **kwargs):
        encoder_out = self.encoder(src_tokens, src_lengths=src_lengths, **kwargs)
        return self.decoder(prev_output_tokens, encoder_out=encoder_out, **kwargs)

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def max_decoder_positions(self):
        return self.decoder.max_positions()
