"""Mirror of fairseq/models/chimera/w2v2_transformer_interlingua.py — the Chimera model:
S2TTransformerInterlinguaModelW2V2 (:28-152), S2T_W2V2_TransformerInterlinguaEncoder (:155-312) with the M-slot
shared-semantic-memory module (:264-298), arch s2t_transformer_w2v2_interlingua_base (:315-343).

Memory module on MI355X: the reference runs each memory layer over cat(h_enc, memory) (L+M rows) with a column mask
that hides the M memory columns and keeps only the last M output rows; rows are independent, so this build runs the
layer on the M query rows only, with K/V = the layer's self_attn_layer_norm + k/v projections of h_enc — the
same numbers for the rows that are kept (Q1: padded encoder frames ARE attended; Q2: memory columns are never
attended), 3.7x fewer FLOPs (credited as such in bench.py, never as the reference-shaped count)."""
import logging
import os

import torch
import torch.nn as nn

from . import functional as CF
from .fairseq_model import EncoderOut, lengths_to_padding_mask
from .modules import Embedding, TransformerEncoderLayer, to_batch_major, to_time_major_view
from .registry import register_model, register_model_architecture
from .s2t_transformer import TransformerDecoderScriptable
from .w2v2_transformer import S2T_W2V2_TransformerEncoder, S2TTransformerModelW2V2, base_architecture

logger = logging.getLogger(__name__)


@register_model("s2t_transformer_w2v2_interlingua")
class S2TTransformerInterlinguaModelW2V2(S2TTransformerModelW2V2):
    def __init__(self, encoder, decoder, debug_options):
        super().__init__(encoder, decoder)
        self.debug_options = debug_options

    @staticmethod
    def add_args(parser):
        """w2v2_transformer_interlingua.py:40-76."""
        S2TTransformerModelW2V2.add_args(parser)
        parser.add_argument("--interlingua-length", type=int, default=16)
        parser.add_argument("--interlingua-layers", type=int, default=3)
        parser.add_argument("--interlingua-debug-options", type=str, nargs="+", default=[], choices=["modal_embedding"])
        parser.add_argument("--non-shared-encoder-layers", type=int, default=0)
        for f in ("--fix-wav2vec", "--fix-interlingua", "--fix-decoder", "--fix-decoder-transformers",
                  "--fix-encoder-transformers", "--reset-encoder", "--no-interlingua"):
            parser.add_argument(f, action="store_true", default=False)

    @classmethod
    def build_model(cls, args, task):
        s2t_transformer_w2v2_interlingua_base(args)

        def build_embedding(dictionary, embed_dim):
            return Embedding(len(dictionary), embed_dim, dictionary.pad())

        encoder_embed_tokens = (build_embedding(task.source_dictionary, args.encoder_embed_dim)
                                if task.source_dictionary is not None else None)
        encoder = cls.build_encoder(args, task.source_dictionary, encoder_embed_tokens)
        decoder_embed_tokens = build_embedding(task.target_dictionary, args.decoder_embed_dim)
        decoder = cls.build_decoder(args, task.target_dictionary, decoder_embed_tokens)
        if args.fix_wav2vec:
            encoder.wav2vec_model.requires_grad_(False)
        if args.fix_encoder_transformers:
            encoder.transformer_layers.requires_grad_(False)
        if args.fix_decoder_transformers:
            decoder.layers.requires_grad_(False)
        if args.fix_decoder:
            decoder.requires_grad_(False)
        if args.fix_interlingua:
            encoder.interlingua_layers.requires_grad_(False)
            encoder.interlingua_embedding.requires_grad_(False)
        return cls(encoder, decoder, args.interlingua_debug_options)

    @classmethod
    def build_encoder(cls, args, src_dict=None, encoder_embed_tokens=None):
        return S2T_W2V2_TransformerInterlinguaEncoder(args, src_dict, encoder_embed_tokens)

    @classmethod
    def build_decoder(cls, args, tgt_dict, embed_tokens):
        return TransformerDecoderScriptable(args, tgt_dict, embed_tokens)

    def forward_with_internal(self, src_tokens, src_lengths, prev_output_tokens, **extra_args):
        """:137-146 -> ((logits, extra), memory [M,B,C])."""
        encoder_out = self.encoder(src_tokens=src_tokens, src_lengths=src_lengths)
        decoder_out = self.decoder(prev_output_tokens=prev_output_tokens, encoder_out=encoder_out)
        return decoder_out, encoder_out.encoder_out

    def upgrade_state_dict_named(self, state_dict, name):
        super().upgrade_state_dict_named(state_dict, name)
        state_dict.pop("encoder.stashed_weights", None)
        state_dict.pop("decoder.stashed_weights", None)
        return state_dict


class S2T_W2V2_TransformerInterlinguaEncoder(S2T_W2V2_TransformerEncoder):
    def __init__(self, args, src_tokens, embed_tokens):
        super().__init__(args)
        # quirk Q1: the memory layers attend EVERY padded frame (all-False key padding mask, :292-295), so the values of the wav2vec2
        # padding frames are consumed here: the padding-free wav2vec2 stack is used only while no dropout is active
        self.wav2vec_model.encoder.padding_rows_consumed = True
        self.max_source_positions = args.max_source_positions
        self.text_embed_tokens = embed_tokens
        if embed_tokens is not None:
            self.encoder_embed_dim = embed_tokens.embedding_dim
        self.debug_options = args.interlingua_debug_options
        assert "modal_embedding" not in self.debug_options, "debug-only option, not built"
        self.non_shared_encoder_layers = args.non_shared_encoder_layers
        assert self.non_shared_encoder_layers == 0, "non-shared encoder layers are not used by chimera/scripts"
        self.reset_encoder = args.reset_encoder
        self.no_interlingua = args.no_interlingua
        assert args.interlingua_layers >= 1
        self.interlingua_embedding = Embedding(args.interlingua_length, args.encoder_embed_dim, 0)
        self.interlingua_layers = nn.ModuleList([TransformerEncoderLayer(args) for _ in range(args.interlingua_layers)])
        for layer in self.interlingua_layers:
            # a memory layer normalises BOTH its query rows (the memory) and its key/value rows (h_enc) with self_attn_layer_norm:
            # that LayerNorm receives two gradients per backward pass, which autograd adds as soon as the second exists — its
            # reductions must not be deferred (kernels.DEFER, functional._may_defer)
            for p in layer.self_attn_layer_norm.parameters():
                p._cst_shared = True
        self.modal_embedding = None

    def upgrade_state_dict_named(self, state_dict, name):
        embed_weight_name = name + ".text_embed_tokens.weight"
        if self.text_embed_tokens is None:
            state_dict.pop(embed_weight_name, None)
        return state_dict

    def max_positions(self):
        return None

    def _front_end(self, src_tokens, src_lengths):
        """The modality-specific part of :207-236: batch-major rows [B, T, C] in front of the shared layers, lengths, padding mask."""
        is_text = not src_tokens.dtype.is_floating_point
        drop_p = self.dropout_module.p if self.training else 0.0
        if is_text:
            input_lengths = src_lengths
            encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=src_tokens.size(1))
            xb = CF.embed_positions(tokens=src_tokens, pad_mask=encoder_padding_mask, embed=self.text_embed_tokens.weight,
                                    pos_table=self.embed_positions.table(src_tokens.size(1), src_tokens.device),
                                    scale=self.embed_scale, pad_idx=self.padding_idx, dropout_p=drop_p)
        else:
            w2v_feature, _, input_lengths = self._get_w2v_feature(src_tokens, src_lengths)
            feature_tm, input_lengths = self.subsample(w2v_feature, input_lengths)
            encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=feature_tm.size(0))
            xb = CF.embed_positions(x=to_batch_major(feature_tm), scale=self.embed_scale, pad_idx=self.padding_idx, dropout_p=drop_p)  # no positions (Q3)
        return xb, input_lengths, encoder_padding_mask

    def _memory(self, h_enc):
        """:264-298 on one modality's encoder output h_enc [T, B, C] -> memory [M, B, C]."""
        batch_size = h_enc.shape[1]
        interlingua = self.interlingua_embedding.weight.unsqueeze(0).expand(batch_size, -1, -1)  # B x M x C
        interlingua = to_time_major_view(interlingua.contiguous())
        for layer in self.interlingua_layers:
            # == layer(cat(h_enc, mem), no key padding (Q1), column mask hiding the memory columns (Q2))[-M:]
            interlingua = layer(interlingua, None, kv=h_enc)
        return interlingua

    def forward_pair(self, audio_tokens, audio_lengths, text_tokens, text_lengths):
        """The audio pass and the text pass of one triplet sample (criterions/triplet_st_mt_contrastive.py:68-100) through the SHARED
        encoder layers as ONE row set: every row-wise operation (LayerNorm, projections, FFN) sees the audio frames of the batch — all
        of them, padding included: the memory attention reads padded frames, quirk Q1 — followed by the text tokens, and
        self-attention runs per sequence on that packed row set (functional.PackedRows: sequence b attends its first kv_len[b] rows,
        which is what the length-derived key padding mask of :232 expresses).  Row for row the values of the two separate passes;
        half the launches, and every shared-layer parameter receives one gradient.  The memory layers then run per modality (their
        key/value rows differ in length) or — default — at once, the shorter modality's rows zero-padded and masked.
        Returns (EncoderOut audio, EncoderOut text, EncoderOut of both as one batch of 2B or None)."""
        xa, len_a, _ = self._front_end(audio_tokens, audio_lengths)   # [B, Ta, C]
        xt, len_t, _ = self._front_end(text_tokens, text_lengths)     # [B, Tt, C]
        B, Ta, C = xa.shape
        Tt = xt.shape[1]
        dev = xa.device
        rows = torch.cat((xa.reshape(B * Ta, C), xt.reshape(B * Tt, C)), 0)
        off = torch.cat((torch.arange(0, B * Ta + 1, Ta, dtype=torch.int32, device=dev),
                         torch.arange(B * Ta + Tt, B * Ta + B * Tt + 1, Tt, dtype=torch.int32, device=dev))).contiguous()
        kvl = torch.cat((torch.clamp(len_a, max=Ta), torch.clamp(len_t, max=Tt))).to(torch.int32).contiguous()
        seq = CF.PackedRows(off, kvl, B * (Ta + Tt), max(Ta, Tt), 2 * B, max(Ta, Tt))
        x = to_time_major_view(rows.unsqueeze(0))  # [rows, 1, C]
        for layer in self.transformer_layers:
            x = layer(x, None, seq=seq)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        xb = to_batch_major(x)[0]
        h_ab, h_tb = xb[:B * Ta].view(B, Ta, C), xb[B * Ta:].view(B, Tt, C)

        def out(mem):
            pm = torch.zeros(mem.shape[1], mem.shape[0], device=dev, dtype=torch.bool)
            return EncoderOut(encoder_out=mem, encoder_padding_mask=pm, encoder_embedding=None, encoder_states=None, src_tokens=None,
                              src_lengths=None)

        if self.no_interlingua or os.environ.get("CST_NO_PAIR_MEMORY"):
            h_a, h_t = to_time_major_view(h_ab), to_time_major_view(h_tb)
            return (out(h_a if self.no_interlingua else self._memory(h_a)), out(h_t if self.no_interlingua else self._memory(h_t)), None)
        # the memory layers on both modalities at once: 2B memories of M slots; key/value rows = each sample's encoder output, the
        # shorter modality padded with zero rows up to the longer one and those rows masked as keys (for the audio half NO frame is
        # masked — quirk Q1; for the text half all Tt positions, padding included, stay visible: exactly the two separate passes)
        Tk = max(Ta, Tt)
        pad_a = torch.nn.functional.pad(h_ab, (0, 0, 0, Tk - Ta)) if Tk > Ta else h_ab
        pad_t = torch.nn.functional.pad(h_tb, (0, 0, 0, Tk - Tt)) if Tk > Tt else h_tb
        kvb = torch.cat((pad_a, pad_t), 0)                                   # [2B, Tk, C]
        mask = torch.zeros(2 * B, Tk, device=dev, dtype=torch.bool)
        if Tk > Ta:
            mask[:B, Ta:] = True
        if Tk > Tt:
            mask[B:, Tt:] = True
        mem = to_time_major_view(self.interlingua_embedding.weight.unsqueeze(0).expand(2 * B, -1, -1).contiguous())
        for layer in self.interlingua_layers:
            mem = layer(mem, mask, kv=to_time_major_view(kvb))
        memb = to_batch_major(mem)                                            # [2B, M, C]
        return out(to_time_major_view(memb[:B])), out(to_time_major_view(memb[B:])), out(mem)

    def forward(self, src_tokens, src_lengths, **extra_args):
        """:207-312."""
        is_text = not src_tokens.dtype.is_floating_point
        drop_p = self.dropout_module.p if self.training else 0.0
        if is_text:
            # text: embed_scale * E[tokens] + sinusoidal positions (Q3: only text gets them, :233-236; the position source is the
            # LENGTH-derived padding mask, as the reference passes it) + dropout — one kernel (cst_embed_pos_fwd)
            input_lengths = src_lengths
            encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=src_tokens.size(1))
            xb = CF.embed_positions(tokens=src_tokens, pad_mask=encoder_padding_mask, embed=self.text_embed_tokens.weight,
                                    pos_table=self.embed_positions.table(src_tokens.size(1), src_tokens.device),
                                    scale=self.embed_scale, pad_idx=self.padding_idx, dropout_p=drop_p)
        else:
            w2v_feature, _, input_lengths = self._get_w2v_feature(src_tokens, src_lengths)
            feature_tm, input_lengths = self.subsample(w2v_feature, input_lengths)
            encoder_padding_mask = lengths_to_padding_mask(input_lengths, max_len=feature_tm.size(0))
            xb = CF.embed_positions(x=to_batch_major(feature_tm), scale=self.embed_scale, pad_idx=self.padding_idx, dropout_p=drop_p)  # no positions (Q3)
        x = to_time_major_view(xb)
        for layer in self.transformer_layers:
            x = layer(x, encoder_padding_mask)
        if self.layer_norm is not None:
            x = self.layer_norm(x)
        length_h, batch_size, _ = x.shape
        if self.no_interlingua:
            interlingua = x
            length_i = length_h
        else:
            h_enc = x
            interlingua = self.interlingua_embedding.weight.unsqueeze(0).expand(batch_size, -1, -1)  # B x M x C
            length_i = interlingua.shape[1]
            interlingua = to_time_major_view(interlingua.contiguous())
            for layer in self.interlingua_layers:
                # == layer(cat(h_enc, mem), no key padding (Q1), column mask hiding the memory columns (Q2))[-M:]
                interlingua = layer(interlingua, None, kv=h_enc)
        encoder_padding_mask = torch.zeros(batch_size, length_i, device=x.device, dtype=torch.bool)
        return EncoderOut(encoder_out=interlingua, encoder_padding_mask=encoder_padding_mask, encoder_embedding=None,
                          encoder_states=None, src_tokens=None, src_lengths=None)


@register_model_architecture("s2t_transformer_w2v2_interlingua", "s2t_transformer_w2v2_interlingua_base")
def s2t_transformer_w2v2_interlingua_base(args):
    """:318-343 incl. quirk Q4: base_architecture runs first, so the 256/4-head defaults below never apply, and
    fix_encoder_transformers reads fix_decoder_transformers."""
    base_architecture(args)
    args.use_asr_finetune_w2v = getattr(args, "use_asr_finetune_w2v", False)
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", 256)
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", 256 * 8)
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", 4)
    args.max_source_positions = getattr(args, "max_source_positions", 1000000)
    args.decoder_attention_heads = getattr(args, "decoder_attention_heads", 4)
    args.dropout = getattr(args, "dropout", 0.1)
    args.fix_wav2vec = getattr(args, "fix_wav2vec", False)
    args.load_pretrained_encoder_from = getattr(args, "load_pretrained_encoder_from", None)
    args.non_shared_encoder_layers = getattr(args, "non_shared_encoder_layers", 0)
    args.fix_encoder_transformers = getattr(args, "fix_decoder_transformers", False)
    args.fix_decoder_transformers = getattr(args, "fix_decoder_transformers", False)
    args.fix_decoder = getattr(args, "fix_decoder", False)
    args.fix_interlingua = getattr(args, "fix_interlingua", False)
    args.no_interlingua = getattr(args, "no_interlingua", False)
    args.reset_encoder = getattr(args, "reset_encoder", False)
    args.interlingua_length = getattr(args, "interlingua_length", 16)
    args.interlingua_layers = getattr(args, "interlingua_layers", 3)
    args.interlingua_debug_options = getattr(args, "interlingua_debug_options", [])
