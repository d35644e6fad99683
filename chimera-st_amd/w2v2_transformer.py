"""Mirror of fairseq/models/chimera/w2v2_transformer.py: S2TTransformerModelW2V2 (:42-236),
S2T_W2V2_TransformerEncoder (:239-432) and the arch presets (:435-506).
This is BASELINE config 2/3 ("s2t_transformer_m dims + wav2vec2 frontend")."""
import argparse
import logging
import math
import os

import torch
import torch.nn as nn

from . import functional as CF
from .fairseq_model import EncoderOut, FairseqEncoder, FairseqEncoderDecoderModel, lengths_to_padding_mask
from .modules import FairseqDropout, LayerNorm, PositionalEmbedding, TransformerEncoderLayer, to_batch_major, to_time_major_view
from .registry import register_model, register_model_architecture
from .s2t_transformer import (Conv1dSubsampler, TransformerDecoderScriptable, add_transformer_args, build_embedding,
                              reorder_encoder_out)
from .wav2vec2 import Wav2Vec2Model, wav2vec_small_args

logger = logging.getLogger(__name__)

SYNTHETIC_W2V = {}  # name -> argparse.Namespace, for checkpoint-free (random-init) builds


def load_w2v_checkpoint(path):
    """torchHLoad (models/chimera/hdfs_utils.py:38-45) -> {"args": Namespace, "model": state_dict}.
    `synthetic:<name>` builds random-init weights from a registered Namespace (no network in the build/bench boxes)."""
    if isinstance(path, str) and path.startswith("synthetic:"):
        name = path.split(":", 1)[1]
        if name == "wav2vec_small" and name not in SYNTHETIC_W2V:
            SYNTHETIC_W2V[name] = wav2vec_small_args()
        return {"args": SYNTHETIC_W2V[name], "model": None}
    torch.serialization.add_safe_globals([argparse.Namespace])
    return torch.load(path, map_location="cpu")


@register_model("s2t_transformer_w2v2")
class S2TTransformerModelW2V2(FairseqEncoderDecoderModel):
    single_use_parameters = True  # one forward pass uses every parameter once (trainer.py: deferred reductions); tied tables are detected by name

    @staticmethod
    def add_args(parser):
        """w2v2_transformer.py:53-172."""
        parser.add_argument("--w2v2-model-path", type=str, metavar="N")
        parser.add_argument("--reset-w2v", action="store_true")
        parser.add_argument("--use-asr-finetune-w2v", action="store_true")
        parser.add_argument("--conv-kernel-sizes", type=str, metavar="N")
        parser.add_argument("--conv-channels", type=int, metavar="N")
        add_transformer_args(parser)
        parser.add_argument("--load-pretrained-encoder-from", type=str, metavar="STR")

    @classmethod
    def build_encoder(cls, args):
        return S2T_W2V2_TransformerEncoder(args)

    @classmethod
    def build_decoder(cls, args, tgt_dict, embed_tokens):
        return TransformerDecoderScriptable(args, tgt_dict, embed_tokens)

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        decoder_embed_tokens = build_embedding(task.target_dictionary, args.decoder_embed_dim)
        encoder = cls.build_encoder(args)
        decoder = cls.build_decoder(args, task.target_dictionary, decoder_embed_tokens)
        return cls(encoder, decoder)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        lprobs = self.decoder.get_normalized_probs(net_output, log_probs, sample)
        lprobs.batch_first = True
        return lprobs

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **extra_args):
        encoder_out = self.encoder(src_tokens=src_tokens, src_lengths=src_lengths)
        return self.decoder(prev_output_tokens=prev_output_tokens, encoder_out=encoder_out)


class S2T_W2V2_TransformerEncoder(FairseqEncoder):
    """wav2vec2 (full features_only path) -> Conv1dSubsampler -> sqrt(d) scale (+ sinusoidal positions) -> N pre-norm layers."""

    def __init__(self, args):
        super().__init__(None)
        assert args.w2v2_model_path is not None
        self.w2v2_model_path = args.w2v2_model_path
        self.use_asr_finetune_w2v = args.use_asr_finetune_w2v
        assert not self.use_asr_finetune_w2v, "wav2vec-CTC checkpoints are not on the Chimera script path"
        self.reset_w2v = getattr(args, "reset_w2v", False)
        self.max_source_positions = args.max_source_positions
        ckpt = load_w2v_checkpoint(self.w2v2_model_path)
        self.w2v_args = ckpt["args"]
        self.wav2vec_model = Wav2Vec2Model.build_model(ckpt["args"], task=None)
        if not self.reset_w2v and ckpt["model"] is not None:
            self.wav2vec_model.load_state_dict(ckpt["model"])
        # this encoder reads the wav2vec2 output only through the subsampler (a few frames past each utterance's end) and masks
        # padding from there on: the padding frames of the wav2vec2 layer stack are never consumed (wav2vec2.TransformerEncoder)
        self.wav2vec_model.encoder.padding_rows_consumed = False
        self.dropout_module = FairseqDropout(p=args.dropout, module_name=self.__class__.__name__)
        self.embed_scale = 1.0 if args.no_scale_embedding else math.sqrt(args.encoder_embed_dim)
        self.padding_idx = 1
        self.subsample = Conv1dSubsampler(self.w2v_args.encoder_embed_dim, args.conv_channels, args.encoder_embed_dim,
                                          [int(k) for k in args.conv_kernel_sizes.split(",")])
        if not self.wav2vec_model.encoder.padding_rows_consumed:
            self.wav2vec_model.encoder.padding_rows_read = self.subsample.input_reach()
        self.embed_positions = PositionalEmbedding(args.max_source_positions, args.encoder_embed_dim, self.padding_idx)
        self.transformer_layers = nn.ModuleList([TransformerEncoderLayer(args) for _ in range(args.encoder_layers)])
        self.layer_norm = LayerNorm(args.encoder_embed_dim) if args.encoder_normalize_before else None

    def _get_w2v_feature(self, src_tokens, src_lengths):
        """:319-336.  Returns batch-major features [B,T1,C], frame padding mask, frame lengths."""
        padding_mask = lengths_to_padding_mask(src_lengths, max_len=src_tokens.size(1))
        w2v_feature, padding_mask = self.wav2vec_model.extract_features(src_tokens, padding_mask)
        output_length = (1 - padding_mask.int()).sum(dim=1)
        return w2v_feature, padding_mask, output_length

    def _packing_plan(self, input_lengths, encoder_padding_mask):
        w2v_plan = getattr(self.wav2vec_model, "last_plan", None)
        if w2v_plan is None or w2v_plan.host_lens is None or os.environ.get("CST_NO_PACK") or os.environ.get("CST_NO_PACK_S2T"):
            return None
        hl = list(w2v_plan.host_lens)
        for _ in range(self.subsample.n_layers):  # Conv1dSubsampler.get_out_seq_lens_tensor on host integers
            hl = [int(math.floor((v - 1) / 2 + 1)) for v in hl]
        T = encoder_padding_mask.size(1)
        if sum(min(v, T) for v in hl) >= len(hl) * T:
            return None  # nothing to drop
        return CF.plan_from_lengths(input_lengths, hl, T)

    def forward(self, src_tokens, src_lengths, **extra_args):
        self.wav2vec_model.last_plan = None
        w2v_feature, _, input_lengths = self._get_w2v_feature(src_tokens, src_lengths)
        # (the stack below runs on the real frames only whenever the wav2vec2 plan's host lengths are there: the subsampler then need not
        #  compute — nor reduce its weight gradients over — the frames behind each utterance's end)
        w2v_plan = getattr(self.wav2vec_model, "last_plan", None)
        will_pack = (w2v_plan is not None and w2v_plan.host_lens is not None and not os.environ.get("CST_NO_PACK")
                     and not os.environ.get("CST_NO_PACK_S2T"))
        x, input_lengths = self.subsample(w2v_feature, input_lengths, padding_unread=will_pack)
        encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=x.size(0))
        # embed_scale * x + sinusoidal positions (audio DOES get positions here, :356-358) + dropout: one kernel
        x = to_time_major_view(CF.embed_positions(pad_mask=encoder_padding_mask, x=to_batch_major(x),
                                                  pos_table=self.embed_positions.table(encoder_padding_mask.size(1), x.device),
                                                  scale=self.embed_scale, pad_idx=self.padding_idx,
                                                  dropout_p=self.dropout_module.p if self.training else 0.0))
        # padding-free layer stack: a padded frame is read by nobody (every attention masks it as a key, the decoder included) and
        # its gradient is exactly zero, so the layers run on the real frames only and the padded rows of the output are zeros.
        # The plan costs no host read of its own: the frame counts follow from the wav2vec2 plan's by the subsampler's formula.
        seq = self._packing_plan(input_lengths, encoder_padding_mask)
        if seq is not None:
            x = to_time_major_view(CF.pack_rows(to_batch_major(x), seq))
        for layer in self.transformer_layers:
            x = layer(x, encoder_padding_mask, seq=seq)
        if seq is not None:
            if self.layer_norm is not None:
                x = self.layer_norm(x)
            x = to_time_major_view(CF.unpack_rows(to_batch_major(x), seq, broadcast=False))
            return EncoderOut(encoder_out=x, encoder_padding_mask=encoder_padding_mask, encoder_embedding=None,
                              encoder_states=None, src_tokens=None, src_lengths=None)
        # (the reference drops an all-False mask here, w2v2_transformer.py:377-378: `.any()` is a host sync in the middle of the
        #  forward pass — the queue drains and the decoder's small kernels are then launch-bound.  An all-False mask gives the
        #  same results in the fused attention kernels, so it is kept.)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return EncoderOut(encoder_out=x, encoder_padding_mask=encoder_padding_mask, encoder_embedding=None,
                          encoder_states=None, src_tokens=None, src_lengths=None)

    def reorder_encoder_out(self, encoder_out, new_order):
        return reorder_encoder_out(encoder_out, new_order)

    def max_positions(self):
        return self.max_source_positions


@register_model_architecture(model_name="s2t_transformer_w2v2", arch_name="s2t_transformer_w2v2")
def base_architecture(args):
    """w2v2_transformer.py:435-477."""
    args.w2v2_model_path = getattr(args, "w2v2_model_path", "./wav2vec_small_100h.pt")
    args.use_asr_finetune_w2v = getattr(args, "use_asr_finetune_w2v", False)
    args.conv_kernel_sizes = getattr(args, "conv_kernel_sizes", "5,5")
    args.conv_channels = getattr(args, "conv_channels", 1024)
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", 512)
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", 2048)
    args.encoder_layers = getattr(args, "encoder_layers", 12)
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", 8)
    args.encoder_normalize_before = getattr(args, "encoder_normalize_before", True)
    args.decoder_embed_dim = getattr(args, "decoder_embed_dim", args.encoder_embed_dim)
    args.decoder_ffn_embed_dim = getattr(args, "decoder_ffn_embed_dim", args.encoder_ffn_embed_dim)
    args.decoder_layers = getattr(args, "decoder_layers", 6)
    args.decoder_attention_heads = getattr(args, "decoder_attention_heads", 8)
    args.decoder_normalize_before = getattr(args, "decoder_normalize_before", True)
    args.decoder_learned_pos = getattr(args, "decoder_learned_pos", False)
    args.dropout = getattr(args, "dropout", 0.1)
    args.attention_dropout = getattr(args, "attention_dropout", args.dropout)
    args.activation_dropout = getattr(args, "activation_dropout", args.dropout)
    args.activation_fn = getattr(args, "activation_fn", "relu")
    args.adaptive_softmax_cutoff = getattr(args, "adaptive_softmax_cutoff", None)
    args.adaptive_softmax_dropout = getattr(args, "adaptive_softmax_dropout", 0)
    args.share_decoder_input_output_embed = getattr(args, "share_decoder_input_output_embed", True)
    args.no_token_positional_embeddings = getattr(args, "no_token_positional_embeddings", False)
    args.adaptive_input = getattr(args, "adaptive_input", False)
    args.decoder_layerdrop = getattr(args, "decoder_layerdrop", 0.0)
    args.decoder_output_dim = getattr(args, "decoder_output_dim", args.decoder_embed_dim)
    args.decoder_input_dim = getattr(args, "decoder_input_dim", args.decoder_embed_dim)
    args.no_scale_embedding = getattr(args, "no_scale_embedding", False)
    args.quant_noise_pq = getattr(args, "quant_noise_pq", 0)
    args.max_source_positions = getattr(args, "max_source_positions", 1000000)
    args.max_target_positions = getattr(args, "max_target_positions", 1024)


@register_model_architecture("s2t_transformer_w2v2", "s2t_transformer_w2v2_s")
def s2t_transformer_w2v2_s(args):
    args.use_asr_finetune_w2v = getattr(args, "use_asr_finetune_w2v", False)
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", 256)
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", 256 * 8)
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", 4)
    args.decoder_attention_heads = getattr(args, "decoder_attention_heads", 4)
    args.dropout = getattr(args, "dropout", 0.1)
    base_architecture(args)


@register_model_architecture("s2t_transformer_w2v2", "s2t_transformer_w2v2yr_s")
def s2t_transformer_w2v2yr_s(args):
    s2t_transformer_w2v2_s(args)


@register_model_architecture("s2t_transformer_w2v2", "s2t_transformer_w2v2_sp")
def s2t_transformer_w2v2_sp(args):
    args.use_asr_finetune_w2v = getattr(args, "use_asr_finetune_w2v", False)
    args.encoder_layers = getattr(args, "encoder_layers", 16)
    s2t_transformer_w2v2_s(args)


@register_model_architecture("s2t_transformer_w2v2", "s2t_transformer_w2v2asr_s")
def s2t_transformer_w2v2asr_s(args):
    args.use_asr_finetune_w2v = getattr(args, "use_asr_finetune_w2v", True)
    s2t_transformer_w2v2_s(args)
