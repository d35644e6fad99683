"""Mirror of fairseq/models/speech_to_text/s2t_transformer.py (Conv1dSubsampler :31-77, S2TTransformerModel :80-266,
S2TTransformerEncoder :269-366, TransformerDecoderScriptable :369-389, arch presets :392-478) and of the decoder in
fairseq/models/transformer.py:530-903."""
import math
from typing import Dict, List, Optional

import os

import torch
import torch.nn as nn

from . import functional as CF
from .fairseq_model import (EncoderOut, FairseqEncoder, FairseqEncoderDecoderModel, FairseqIncrementalDecoder,
                            lengths_to_padding_mask)
from .modules import (Embedding, FairseqDropout, LayerNorm, Linear, PositionalEmbedding, TransformerDecoderLayer,
                      TransformerEncoderLayer, causal_mask, to_batch_major, to_time_major_view)
from .registry import register_model, register_model_architecture


class Conv1dSubsampler(nn.Module):
    """s2t_transformer.py:31-77: n x (Conv1d(k, stride 2, pad k//2) + GLU).  Channels-last implicit GEMMs (bias in the
    epilogue) + a GLU kernel; input [B,T,C] is consumed as is (the reference's transpose(1,2) is a layout no-op here)."""

    def __init__(self, in_channels, mid_channels, out_channels, kernel_sizes=(3, 3)):
        super().__init__()
        self.n_layers = len(kernel_sizes)
        self.conv_layers = nn.ModuleList()
        for i, k in enumerate(kernel_sizes):
            cin = in_channels if i == 0 else mid_channels // 2
            cout = mid_channels if i < self.n_layers - 1 else out_channels * 2
            conv = nn.Module()
            ref = nn.Conv1d(cin, cout, k, stride=2, padding=k // 2)  # same default init as the reference
            conv.weight = nn.Parameter(ref.weight.detach().clone())
            conv.bias = nn.Parameter(ref.bias.detach().clone())
            conv.kernel_size = k
            self.conv_layers.append(conv)

    def input_reach(self):
        """How many input frames past the last real one the last real OUTPUT frame reads: r <- r * stride + (k - 1 - pad) from the
        last layer to the first (k = 5, stride 2, pad 2, two layers: 6)."""
        r = 0
        for conv in reversed(self.conv_layers):
            r = 2 * r + (conv.kernel_size - 1 - conv.kernel_size // 2)
        return r

    def get_out_seq_lens_tensor(self, in_seq_lens_tensor):
        out = in_seq_lens_tensor.clone()
        for _ in range(self.n_layers):
            out = ((out.float() - 1) / 2 + 1).floor().long()
        return out

    def forward(self, src_tokens, src_lengths, padding_unread=False):
        """padding_unread: the caller drops every output frame behind an utterance's subsampled length (the s2t encoder's padding-free
        layer stack) — their gradient is exactly zero and nobody reads their values, so each conv computes only the frames the live
        outputs of the layer behind it read (cst_gemm_desc.m_len) and its weight-gradient GEMM stops there (k_len)."""
        x = src_tokens  # B x T x C
        lens = [src_lengths]
        for _ in range(self.n_layers):
            lens.append(((lens[-1].float() - 1) / 2 + 1).floor().long())
        limits = [None] * self.n_layers
        if padding_unread and not os.environ.get("CST_NO_MLEN"):
            need = lens[-1]  # frames of the last layer's output that are read
            for i in reversed(range(self.n_layers)):
                limits[i] = need.to(torch.int32).contiguous()
                k = self.conv_layers[i].kernel_size
                need = (need - 1) * 2 + (k - k // 2)  # the input frames those read: up to 2 (need - 1) + k - 1 - pad
        for i, conv in enumerate(self.conv_layers):
            z, _ = CF.conv1d_cl(x, conv.weight, conv.bias, 2, pad=conv.kernel_size // 2, nz_out=limits[i], unread_ok=limits[i] is not None)
            x = CF.glu(z)
        return to_time_major_view(x), lens[-1]  # T x B x C view of batch-major storage


class TransformerDecoder(FairseqIncrementalDecoder):
    """models/transformer.py:530-903 restricted to the Chimera configuration (sinusoidal positions, no adaptive softmax,
    no layerdrop, pre-norm, tied or untied output projection)."""

    def __init__(self, args, dictionary, embed_tokens, no_encoder_attn=False):
        self.args = args
        super().__init__(dictionary)
        self.register_buffer("version", torch.Tensor([3]))
        self._future_mask = None
        self.dropout_module = FairseqDropout(args.dropout, module_name=self.__class__.__name__)
        self.decoder_layerdrop = getattr(args, "decoder_layerdrop", 0.0)
        assert self.decoder_layerdrop == 0.0
        self.share_input_output_embed = args.share_decoder_input_output_embed
        input_embed_dim = embed_tokens.embedding_dim
        embed_dim = args.decoder_embed_dim
        self.embed_dim = embed_dim
        self.output_embed_dim = getattr(args, "decoder_output_dim", embed_dim)
        assert input_embed_dim == embed_dim and self.output_embed_dim == embed_dim
        self.padding_idx = embed_tokens.padding_idx
        self.max_target_positions = args.max_target_positions
        self.embed_tokens = embed_tokens
        self.embed_scale = 1.0 if getattr(args, "no_scale_embedding", False) else math.sqrt(embed_dim)
        self.project_in_dim = None
        self.embed_positions = (PositionalEmbedding(args.max_target_positions, embed_dim, self.padding_idx,
                                                    learned=getattr(args, "decoder_learned_pos", False))
                                if not getattr(args, "no_token_positional_embeddings", False) else None)
        self.layernorm_embedding = None
        self.cross_self_attention = False
        self.layers = nn.ModuleList([TransformerDecoderLayer(args, no_encoder_attn) for _ in range(args.decoder_layers)])
        self.num_layers = len(self.layers)
        if args.decoder_normalize_before and not getattr(args, "no_decoder_final_norm", False):
            self.layer_norm = LayerNorm(embed_dim)
        else:
            self.layer_norm = None
        self.project_out_dim = None
        self.adaptive_softmax = None
        if self.share_input_output_embed:
            self.output_projection = Linear(self.embed_tokens.weight.shape[1], self.embed_tokens.weight.shape[0], bias=False)
            self.output_projection.weight = self.embed_tokens.weight
        else:
            self.output_projection = Linear(self.output_embed_dim, len(dictionary), bias=False)
            nn.init.normal_(self.output_projection.weight, mean=0, std=self.output_embed_dim ** -0.5)

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, features_only=False,
                full_context_alignment=False, alignment_layer=None, alignment_heads=None, src_lengths=None,
                return_all_hiddens=False, **unused):
        x, extra = self.extract_features(prev_output_tokens, encoder_out=encoder_out, incremental_state=incremental_state,
                                         full_context_alignment=full_context_alignment)
        if not features_only:
            x = self.output_layer(x)
        return x, extra

    def extract_features(self, prev_output_tokens, encoder_out=None, incremental_state=None, full_context_alignment=False,
                         alignment_layer=None, alignment_heads=None):
        return self.extract_features_scriptable(prev_output_tokens, encoder_out, incremental_state, full_context_alignment)

    def extract_features_scriptable(self, prev_output_tokens, encoder_out=None, incremental_state=None,
                                    full_context_alignment=False, alignment_layer=None, alignment_heads=None):
        """transformer.py:720-828."""
        if incremental_state is None and prev_output_tokens.is_cuda:
            # training / full-sequence path: embed_scale * E[tokens] + sinusoidal positions + dropout in ONE kernel
            # (cst_embed_pos_fwd: make_positions evaluated in-kernel; deterministic table gradient cst_embed_bwd)
            x = CF.embed_positions(tokens=prev_output_tokens, embed=self.embed_tokens.weight,
                                   pos_table=(self.embed_positions.table(prev_output_tokens.size(1), prev_output_tokens.device)
                                              if self.embed_positions is not None else None),
                                   scale=self.embed_scale, pad_idx=self.padding_idx,
                                   dropout_p=self.dropout_module.p if self.training else 0.0)
        else:
            positions = (self.embed_positions(prev_output_tokens, incremental_state=incremental_state)
                         if self.embed_positions is not None else None)
            if incremental_state is not None:
                prev_output_tokens = prev_output_tokens[:, -1:]
                if positions is not None:
                    positions = positions[:, -1:]
            x = self.embed_scale * self.embed_tokens(prev_output_tokens)  # B x T x C, batch-major
            if positions is not None:
                x = x + positions.to(x.dtype)
            x = self.dropout_module(x)
        x = to_time_major_view(x.contiguous())
        # transformer.py:768-770 builds this mask only `if prev_output_tokens.eq(pad).any()` — a host sync between encoder and
        # decoder; the mask of a pad-free batch is all-False and changes nothing, so the full-sequence (training) path always
        # builds it; the host-driven single-step path keeps the reference's test
        if incremental_state is None:
            self_attn_padding_mask = prev_output_tokens.eq(self.padding_idx)
        else:
            self_attn_padding_mask = prev_output_tokens.eq(self.padding_idx) if prev_output_tokens.eq(self.padding_idx).any() else None
        inner_states = [x]
        for idx, layer in enumerate(self.layers):
            if incremental_state is None and not full_context_alignment:
                self_attn_mask = self.buffered_future_mask(x)
            else:
                self_attn_mask = None
            x, _, _ = layer(x, encoder_out.encoder_out if encoder_out is not None else None,
                            encoder_out.encoder_padding_mask if encoder_out is not None else None, incremental_state,
                            self_attn_mask=self_attn_mask, self_attn_padding_mask=self_attn_padding_mask)
            inner_states.append(x)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        x = x.transpose(0, 1)  # B x T x C (contiguous: batch-major storage)
        return x, {"attn": [None], "inner_states": inner_states}

    def output_layer(self, features):
        """transformer.py:830-836 — tied vocabulary projection (cst_gemm, V x C weight k-major)."""
        return self.output_projection(features)

    def max_positions(self):
        if self.embed_positions is None:
            return self.max_target_positions
        return min(self.max_target_positions, self.embed_positions.max_positions)

    def buffered_future_mask(self, tensor):
        dim = tensor.size(0)
        if self._future_mask is None or self._future_mask.device != tensor.device or self._future_mask.size(0) < dim:
            self._future_mask = causal_mask(max(dim, 256), tensor.device)
        m = self._future_mask[:dim, :dim]
        m.cst_kind = "causal"
        return m


class TransformerDecoderScriptable(TransformerDecoder):
    """s2t_transformer.py:369-389."""

    def extract_features(self, prev_output_tokens, encoder_out=None, incremental_state=None, full_context_alignment=False,
                         alignment_layer=None, alignment_heads=None):
        x, _ = self.extract_features_scriptable(prev_output_tokens, encoder_out, incremental_state, full_context_alignment)
        return x, None


def build_embedding(dictionary, embed_dim):
    return Embedding(len(dictionary), embed_dim, dictionary.pad())


def reorder_encoder_out(encoder_out: EncoderOut, new_order):
    """w2v2_transformer.py:388-429 / s2t_transformer.py:332-366."""
    return EncoderOut(
        encoder_out=None if encoder_out.encoder_out is None else encoder_out.encoder_out.index_select(1, new_order),
        encoder_padding_mask=None if encoder_out.encoder_padding_mask is None
        else encoder_out.encoder_padding_mask.index_select(0, new_order),
        encoder_embedding=None if encoder_out.encoder_embedding is None
        else encoder_out.encoder_embedding.index_select(0, new_order),
        encoder_states=None, src_tokens=None, src_lengths=None)


@register_model("s2t_transformer")
class S2TTransformerModel(FairseqEncoderDecoderModel):
    """s2t_transformer.py:80-266 — filter-bank input variant (fbank [B,T,80] -> Conv1dSubsampler -> Transformer).
    Accepts and ignores the collater's `mask` kwarg (the reference raises TypeError there)."""
    single_use_parameters = True  # one forward pass uses every parameter once (trainer.py: deferred reductions); tied tables are detected by name

    @staticmethod
    def add_args(parser):
        add_transformer_args(parser)
        parser.add_argument("--conv-kernel-sizes", type=str)
        parser.add_argument("--conv-channels", type=int)
        parser.add_argument("--load-pretrained-encoder-from", type=str, default=None)

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        decoder_embed_tokens = build_embedding(task.target_dictionary, args.decoder_embed_dim)
        encoder = S2TTransformerEncoder(args)
        decoder = TransformerDecoderScriptable(args, task.target_dictionary, decoder_embed_tokens)
        return cls(encoder, decoder)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        lprobs = self.decoder.get_normalized_probs(net_output, log_probs, sample)
        lprobs.batch_first = True
        return lprobs

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **extra_args):
        encoder_out = self.encoder(src_tokens=src_tokens, src_lengths=src_lengths)
        return self.decoder(prev_output_tokens=prev_output_tokens, encoder_out=encoder_out)


class S2TTransformerEncoder(FairseqEncoder):
    """s2t_transformer.py:269-366."""

    def __init__(self, args):
        super().__init__(None)
        self.dropout_module = FairseqDropout(p=args.dropout, module_name=self.__class__.__name__)
        self.embed_scale = 1.0 if args.no_scale_embedding else math.sqrt(args.encoder_embed_dim)
        self.padding_idx = 1
        self.subsample = Conv1dSubsampler(args.input_feat_per_channel * args.input_channels, args.conv_channels,
                                          args.encoder_embed_dim, [int(k) for k in args.conv_kernel_sizes.split(",")])
        self.embed_positions = PositionalEmbedding(args.max_source_positions, args.encoder_embed_dim, self.padding_idx)
        self.transformer_layers = nn.ModuleList([TransformerEncoderLayer(args) for _ in range(args.encoder_layers)])
        self.layer_norm = LayerNorm(args.encoder_embed_dim) if args.encoder_normalize_before else None

    def forward(self, src_tokens, src_lengths, **extra_args):
        x, input_lengths = self.subsample(src_tokens, src_lengths)
        encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=x.size(0))
        # embed_scale * x + sinusoidal positions (from the padding mask) + dropout: one kernel (cst_embed_pos_fwd)
        x = to_time_major_view(CF.embed_positions(pad_mask=encoder_padding_mask, x=to_batch_major(x),
                                                  pos_table=self.embed_positions.table(encoder_padding_mask.size(1), x.device),
                                                  scale=self.embed_scale, pad_idx=self.padding_idx,
                                                  dropout_p=self.dropout_module.p if self.training else 0.0))
        for layer in self.transformer_layers:
            x = layer(x, encoder_padding_mask)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        return EncoderOut(encoder_out=x, encoder_padding_mask=encoder_padding_mask, encoder_embedding=None,
                          encoder_states=None, src_tokens=None, src_lengths=None)

    def reorder_encoder_out(self, encoder_out, new_order):
        return reorder_encoder_out(encoder_out, new_order)


def add_transformer_args(parser):
    """Flag set of s2t_transformer.py:96-196 / chimera w2v2_transformer.py:53-172."""
    parser.add_argument("--activation-fn", type=str, default="relu", choices=["relu", "gelu"])
    parser.add_argument("--dropout", type=float, metavar="D")
    parser.add_argument("--attention-dropout", type=float, metavar="D")
    parser.add_argument("--activation-dropout", "--relu-dropout", type=float, metavar="D")
    parser.add_argument("--encoder-embed-dim", type=int, metavar="N")
    parser.add_argument("--encoder-ffn-embed-dim", type=int, metavar="N")
    parser.add_argument("--encoder-layers", type=int, metavar="N")
    parser.add_argument("--encoder-attention-heads", type=int, metavar="N")
    parser.add_argument("--encoder-normalize-before", action="store_true")
    parser.add_argument("--decoder-embed-dim", type=int, metavar="N")
    parser.add_argument("--decoder-ffn-embed-dim", type=int, metavar="N")
    parser.add_argument("--decoder-layers", type=int, metavar="N")
    parser.add_argument("--decoder-attention-heads", type=int, metavar="N")
    parser.add_argument("--decoder-normalize-before", action="store_true")
    parser.add_argument("--share-decoder-input-output-embed", action="store_true")
    parser.add_argument("--layernorm-embedding", action="store_true")
    parser.add_argument("--no-scale-embedding", action="store_true")


@register_model_architecture(model_name="s2t_transformer", arch_name="s2t_transformer")
def base_architecture(args):
    """s2t_transformer.py:392-430."""
    args.conv_kernel_sizes = getattr(args, "conv_kernel_sizes", "5,5")
    args.conv_channels = getattr(args, "conv_channels", 1024)
    args.input_feat_per_channel = getattr(args, "input_feat_per_channel", 80)
    args.input_channels = getattr(args, "input_channels", 1)
    args.max_source_positions = getattr(args, "max_source_positions", 6000)
    args.max_target_positions = getattr(args, "max_target_positions", 1024)
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
    args.share_decoder_input_output_embed = getattr(args, "share_decoder_input_output_embed", False)
    args.no_token_positional_embeddings = getattr(args, "no_token_positional_embeddings", False)
    args.decoder_layerdrop = getattr(args, "decoder_layerdrop", 0.0)
    args.decoder_output_dim = getattr(args, "decoder_output_dim", args.decoder_embed_dim)
    args.decoder_input_dim = getattr(args, "decoder_input_dim", args.decoder_embed_dim)
    args.no_scale_embedding = getattr(args, "no_scale_embedding", False)
    args.quant_noise_pq = getattr(args, "quant_noise_pq", 0)


def _preset(name, **kw):
    @register_model_architecture("s2t_transformer", name)
    def fn(args):
        for k, v in kw.items():
            setattr(args, k, getattr(args, k, v))
        base_architecture(args)

    fn.__name__ = name
    return fn


# s2t_transformer.py:433-478
s2t_transformer_s = _preset("s2t_transformer_s", encoder_embed_dim=256, encoder_ffn_embed_dim=256 * 8,
                            encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.1)
s2t_transformer_sp = _preset("s2t_transformer_sp", encoder_layers=16, encoder_embed_dim=256, encoder_ffn_embed_dim=256 * 8,
                             encoder_attention_heads=4, decoder_attention_heads=4, dropout=0.1)
s2t_transformer_m = _preset("s2t_transformer_m", encoder_embed_dim=512, encoder_ffn_embed_dim=512 * 4,
                            encoder_attention_heads=8, decoder_attention_heads=8, dropout=0.15)
s2t_transformer_mp = _preset("s2t_transformer_mp", encoder_layers=16, encoder_embed_dim=512, encoder_ffn_embed_dim=512 * 4,
                             encoder_attention_heads=8, decoder_attention_heads=8, dropout=0.15)
s2t_transformer_l = _preset("s2t_transformer_l", encoder_embed_dim=1024, encoder_ffn_embed_dim=1024 * 4,
                            encoder_attention_heads=16, decoder_attention_heads=16, dropout=0.2)
s2t_transformer_lp = _preset("s2t_transformer_lp", encoder_layers=16, encoder_embed_dim=1024, encoder_ffn_embed_dim=1024 * 4,
                             encoder_attention_heads=16, decoder_attention_heads=16, dropout=0.2)
