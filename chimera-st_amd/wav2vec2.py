"""wav2vec2 feature path (features_only) — mirror of fairseq/models/wav2vec/wav2vec2.py on the Chimera
path: ConvFeatureExtractionModel (:685-763), Wav2Vec2Model.forward features_only branch (:527-586) and
extract_features (:650-652), TransformerEncoder (:766-861), TransformerSentenceEncoderLayer (:864-959).

Pre-training-only parts (quantizer, negatives sampling, masking) are out of scope;
their parameters that exist in every checkpoint (mask_emb, project_q, final_proj) are kept so
state dicts round-trip.

MI355X layout: the whole CNN runs channels-last [B, L, C] so every conv after layer 0 is an implicit GEMM on
MFMA tiles with the GELU in the epilogue, and `features.transpose(1, 2)` (:539) is free."""
import math
import os
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

from . import functional as CF
from . import kernels as K
from .distributed import notify_unused_parameters
from .modules import LayerNorm, Linear, MultiheadAttention, to_batch_major, to_time_major_view
from .registry import register_model, register_model_architecture


class _Slot(nn.Module):
    """Parameter holder that keeps the reference's nn.Sequential index in state-dict keys (conv_layers.N.0.weight ...)."""


class ConvFeatureExtractionModel(nn.Module):
    """wav2vec2.py:685-763, mode "default": layer 0 = Conv1d(no bias)+GroupNorm(C,C)+GELU, layers 1.. = Conv1d+GELU."""

    def __init__(self, conv_layers, dropout=0.0, mode="default", conv_bias=False):
        super().__init__()
        assert mode == "default", "extractor_mode=layer_norm is not on the Chimera path (wav2vec_small uses default)"
        assert not conv_bias and dropout == 0.0
        self.conv_spec = [tuple(c) for c in conv_layers]
        self.conv_layers = nn.ModuleList()
        in_d = 1
        for i, (dim, k, stride) in enumerate(self.conv_spec):
            blk = nn.Module()
            conv = nn.Module()
            conv.weight = nn.Parameter(torch.empty(dim, in_d, k))
            nn.init.kaiming_normal_(conv.weight)
            blk.add_module("0", conv)
            if i == 0:
                gn = nn.Module()
                gn.weight = nn.Parameter(torch.ones(dim))
                gn.bias = nn.Parameter(torch.zeros(dim))
                blk.add_module("2", gn)
            self.conv_layers.append(blk)
            in_d = dim

    def forward(self, x, nz_last=None):
        """x [B,S] raw samples -> channels-last features [B, T1, C] (the reference returns [B,C,T1]).
        nz_last (int32 [B], optional): the caller guarantees that the gradient of the returned features is exactly zero at frames
        t >= nz_last[b] (wav2vec2 zeroes the padded frames, wav2vec2.py:820-821).  The bound is carried down the stack — a
        frame t of layer i+1 reads rows [s t, s t + k) of layer i, so layer i's gradient is zero from (nz - 1) s + k on — and the
        weight-gradient GEMMs of each layer stop their reduction over frames there (cst_gemm_desc.k_len): exact."""
        nz = [None] * len(self.conv_spec)
        res = [None] * len(self.conv_spec)
        lim = None
        if nz_last is not None:
            # nz[i][b]: frames of layer i that anything reads; res[i][r][b]: live rows of residue class r of layer i's input gradient
            lim = K.conv_row_limits(nz_last.to(torch.int32).contiguous(), self.conv_spec, x.shape[1])
            for i, (_, _, st_) in enumerate(self.conv_spec):
                nz[i] = lim[i, 0]
                res[i] = lim[i, 1:1 + st_]
        l0 = self.conv_layers[0]
        dim, k, stride = self.conv_spec[0]
        # layer 0 writes only the frames layer 1's live tiles read, its backward reads only the frames that carry gradient
        y = CF.conv0_gn_gelu(x, getattr(l0, "0").weight, getattr(l0, "2").weight, getattr(l0, "2").bias, stride,
                             write_limit=lim[0, 1] if lim is not None and len(self.conv_spec) > 1 else None,
                             grad_limit=nz[0])
        z = None
        n = len(self.conv_spec)
        for i in range(1, n):
            dim, k, stride = self.conv_spec[i]
            w = getattr(self.conv_layers[i], "0").weight
            # fold GELU' of layer i-1 into layer i's col2im pass; layer i then receives d/dz directly
            # rows t >= nz[i][b] of layer i are frames nobody reads (the stack's output is zeroed behind the utterance's end,
            # wav2vec2.py:820-821) and whose gradient is exactly zero: their GEMM tiles skip the K loop (cst_gemm_desc.m_len / k_len)
            y, z = CF.conv1d_cl(y, w, None, stride, pad=0, act="gelu", prev_z=z, grad_is_dz=(i < n - 1), nz_out=nz[i], nz_in=res[i])
        return y

    def output_length(self, s):
        for (_, k, st) in self.conv_spec:
            s = (s - k) // st + 1
        return s


class TransformerSentenceEncoderLayer(nn.Module):
    """wav2vec2.py:864-959, post-norm branch (layer_norm_first=False)."""

    def __init__(self, embedding_dim=768, ffn_embedding_dim=3072, num_attention_heads=8, dropout=0.1,
                 attention_dropout=0.1, activation_dropout=0.1, activation_fn="relu", layer_norm_first=False):
        super().__init__()
        assert not layer_norm_first, "layer_norm_first=True (wav2vec2 large) is not built yet"
        self.embedding_dim = embedding_dim
        self.dropout, self.activation_dropout = dropout, activation_dropout
        self.activation_fn = activation_fn
        self.self_attn = MultiheadAttention(embedding_dim, num_attention_heads, dropout=attention_dropout, self_attention=True)
        self.layer_norm_first = layer_norm_first
        self.self_attn_layer_norm = LayerNorm(embedding_dim)
        self.fc1 = Linear(embedding_dim, ffn_embedding_dim)
        self.fc2 = Linear(ffn_embedding_dim, embedding_dim)
        self.final_layer_norm = LayerNorm(embedding_dim)

    def forward(self, x, self_attn_mask=None, self_attn_padding_mask=None, need_weights=False, att_args=None, seq=None):
        p_drop = float(self.dropout) if self.training else 0.0             # dropout1 / dropout3 (wav2vec2.py:940-955)
        p_act = float(self.activation_dropout) if self.training else 0.0   # dropout2
        residual = x
        x, _ = self.self_attn(query=x, key=x, value=x, key_padding_mask=self_attn_padding_mask, need_weights=False,
                              resid=residual, out_dropout_p=p_drop, seq=seq)  # x = residual + dropout1(attn) (out_proj epilogue)
        x = self.self_attn_layer_norm(x)
        residual = x
        x = to_time_major_view(CF.ffn(to_batch_major(x), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                                      self.activation_fn, resid=to_batch_major(residual), activation_dropout_p=p_act,
                                      dropout_p=p_drop))
        x = self.final_layer_norm(x)
        return x, None


def init_bert_params(module):
    """modules/transformer_sentence_encoder.py:22-50 — normal(0, 0.02) for Linear / attention projections."""
    if isinstance(module, Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, MultiheadAttention):
        for p in (module.q_proj, module.k_proj, module.v_proj):
            p.weight.data.normal_(mean=0.0, std=0.02)


class TransformerEncoder(nn.Module):
    """wav2vec2.py:766-861: weight-normed grouped pos_conv + SamePad + GELU, LayerNorm, N post-norm layers, layerdrop."""

    def __init__(self, args):
        super().__init__()
        self.dropout = args.dropout
        self.embedding_dim = args.encoder_embed_dim
        self.conv_pos, self.conv_pos_groups = args.conv_pos, args.conv_pos_groups
        cg = self.embedding_dim // args.conv_pos_groups
        std = math.sqrt(4 / (args.conv_pos * self.embedding_dim))
        pc = nn.Module()  # nn.utils.weight_norm(conv, name="weight", dim=2): weight_g [1,1,k], weight_v [C, C/g, k]
        v = torch.empty(self.embedding_dim, cg, args.conv_pos).normal_(0, std)
        pc.bias = nn.Parameter(torch.zeros(self.embedding_dim))
        pc.weight_g = nn.Parameter(v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        pc.weight_v = nn.Parameter(v)
        self.pos_conv = nn.Module()
        self.pos_conv.add_module("0", pc)
        self.layers = nn.ModuleList([
            TransformerSentenceEncoderLayer(
                embedding_dim=self.embedding_dim, ffn_embedding_dim=args.encoder_ffn_embed_dim,
                num_attention_heads=args.encoder_attention_heads, dropout=self.dropout,
                attention_dropout=args.attention_dropout, activation_dropout=args.activation_dropout,
                activation_fn=args.activation_fn, layer_norm_first=args.layer_norm_first)
            for _ in range(args.encoder_layers)])
        self.layer_norm_first = args.layer_norm_first
        self.layer_norm = LayerNorm(self.embedding_dim)
        self.layerdrop = args.encoder_layerdrop
        self.apply(init_bert_params)

    def pos_conv_weight(self):
        pc = getattr(self.pos_conv, "0")
        v = pc.weight_v
        if v.is_cuda and v.shape[-1] % 8 == 0 and v.shape[-1] <= 256 and 256 % (v.shape[-1] // 8) == 0:
            return CF.weight_norm_last_dim(v, pc.weight_g)  # cst_weight_norm_fwd/bwd
        norm = v.float().pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        return (v.float() * (pc.weight_g.float() / norm)).to(v.dtype)

    def forward(self, x, padding_mask=None, plan=None):
        x, padding_mask = self.extract_features(x, padding_mask, plan)
        return x, padding_mask

    # Padding-free layer stack.  Rows of the 12 post-norm layers are independent of each other except through attention, whose
    # keys are the real frames only.  A padding frame t >= len + conv_pos/2 of an utterance sees nothing but zeros through the
    # positional convolution (the frames were zeroed at :820-821), so ALL such frames of an utterance enter the stack with one and
    # the same vector and leave every layer with one and the same vector: the stack runs on len + conv_pos/2 + 1 rows per
    # utterance and the last row is copied to the frames behind it (functional.pack_rows / unpack_rows).  With every dropout
    # inactive this reproduces the padded computation bit for bit, padding frames included.  With dropout active the copies
    # share one mask instead of drawing their own: allowed only where nobody reads those frames (`padding_rows_consumed = False`,
    # set by the s2t_transformer_w2v2 encoder: its subsampler reads a handful of frames past the end, all inside the margin, and
    # every later layer masks the padding); the Chimera memory attends every padded frame (quirk Q1), so there the padded stack
    # stays whenever a dropout is active.
    padding_rows_consumed = True
    # how many frames past an utterance's end the consumer reads when it does not consume the padding rows (s2t encoder: the reach of
    # its Conv1dSubsampler, 6 frames for two k = 5 / stride 2 layers).  None = unknown: keep the positional convolution's reach.
    padding_rows_read = None

    def packing_margin(self):
        """Padding frames kept per utterance in front of the one representative row.  Bit-identity with the padded stack on EVERY
        frame needs the positional convolution's reach (conv_pos // 2: behind it all padding frames are one and the same vector);
        a consumer that reads only `padding_rows_read` frames past the end needs only those computed exactly — the frames behind
        them are then copies of a row nobody reads (forward bits of every frame that IS read, and every gradient, are unchanged:
        rows do not interact except through attention, whose keys are the real frames).  6 % fewer rows at the bench's lengths."""
        reach = self.conv_pos // 2
        # (CST_NO_PACK_S2T=1 — the consumer's own layer stack runs padded and then computes, and returns, its padding frames from
        #  ours: that debugging mode reproduces the reference on every frame, so it keeps the full reach)
        if not self.padding_rows_consumed and self.padding_rows_read is not None and not os.environ.get("CST_NO_PACK_S2T"):
            reach = min(reach, int(self.padding_rows_read))
        return reach

    def wants_packing(self, padding_mask):
        if padding_mask is None or os.environ.get("CST_NO_PACK"):
            return False
        noisy = self.training and (self.dropout > 0 or any(l.self_attn.dropout_module.p > 0 or l.activation_dropout > 0 for l in self.layers))
        return (not noisy) or (not self.padding_rows_consumed)

    def extract_features(self, x, padding_mask=None, plan=None):
        """x [B,T,C] batch-major."""
        if padding_mask is not None:
            x = CF.mask_rows(x, padding_mask)  # x[padding_mask] = 0 (:820-821)
        pc = getattr(self.pos_conv, "0")
        if plan is None and self.wants_packing(padding_mask):
            plan = CF.plan_packed_rows(padding_mask, self.packing_margin())
        if plan is not None and plan.rows >= plan.B * plan.T:
            plan = None  # nothing to drop (no padding beyond the margin)
        # frame limits of the positional convolution (functional._PosConvFn): the input is zero from each utterance's end on (the
        # lines above), so the output tiles whose windows lie in the padding skip their products — exact on every frame — and with a
        # packing plan the gradient of the output is zero behind the rows the plan keeps, which bounds the dX GEMM the same way
        lens = grad_rows = grad_rows_host = None
        if padding_mask is not None and not os.environ.get("CST_NO_POSCONV_LIMITS") and not os.environ.get("CST_NO_MLEN"):  # (CST_NO_MLEN: every frame of every conv)
            lens = plan.kv_len if plan is not None else (~padding_mask).sum(dim=1).to(torch.int32)
            if plan is not None:
                grad_rows = plan.offsets[1:] - plan.offsets[:-1]
                if plan.host_lens is not None and plan.kept_host is not None:
                    grad_rows_host = plan.kept_host
        x = CF.pos_conv_gelu_residual(x, self.pos_conv_weight(), pc.bias, self.conv_pos_groups, lens, grad_rows, grad_rows_host)  # x += GELU(SamePad(conv(x)))
        if plan is not None:
            x = CF.pack_rows(x, plan)  # [1, rows, C]
        x = self.layer_norm(x)
        if self.training and self.dropout > 0:
            x = CF.dropout(x, self.dropout)  # F.dropout(x, p=self.dropout) (:830)
        x = to_time_major_view(x)
        for layer in self.layers:
            dropout_probability = np.random.random()  # same RNG call order as the reference (:836-840)
            if not self.training or (dropout_probability > self.layerdrop):
                x, _ = layer(x, self_attn_padding_mask=None if plan is not None else padding_mask, need_weights=False, seq=plan)
            else:
                notify_unused_parameters(layer.parameters())  # keeps the overlapped bucket order moving (distributed.py)
        x = to_batch_major(x)
        if plan is not None:
            # [B, T, C]; frames behind the kept rows repeat the last kept row — unless the consumer declared how far past an utterance's
            # end it reads and the plan keeps exactly those frames (packing_margin): then nobody reads the frames behind them, they come
            # back as zeros and the backward pass does not walk their (exactly zero) gradients to add them into the last kept row
            # (one wave per utterance summing up to a thousand rows: 4 x 52 us per update)
            unread = (not self.padding_rows_consumed and self.padding_rows_read is not None and not os.environ.get("CST_NO_PACK_S2T")
                      and not os.environ.get("CST_UNPACK_BROADCAST"))
            x = CF.unpack_rows(x, plan, broadcast=not unread)
        return x, padding_mask


@register_model("wav2vec2")
class Wav2Vec2Model(nn.Module):
    """wav2vec2.py:33-680, restricted to what extract_features (features_only) touches."""

    @staticmethod
    def add_args(parser):
        pass

    def __init__(self, args):
        super().__init__()
        self.args = args
        feature_enc_layers = eval(args.conv_feature_layers)
        self.embed = feature_enc_layers[-1][0]
        self.feature_extractor = ConvFeatureExtractionModel(conv_layers=feature_enc_layers, dropout=0.0,
                                                            mode=args.extractor_mode, conv_bias=args.conv_bias)
        self.post_extract_proj = (Linear(self.embed, args.encoder_embed_dim)
                                  if self.embed != args.encoder_embed_dim and not args.quantize_input else None)
        self.feature_grad_mult = args.feature_grad_mult
        self.dropout_input_p = getattr(args, "dropout_input", 0)
        final_dim = args.final_dim if args.final_dim > 0 else args.encoder_embed_dim
        # registration ORDER follows wav2vec2.py:306-420 (post_extract_proj, quantizer, project_q, [input_quantizer, project_inp],
        # mask_emb, encoder, layer_norm, [target_glu], final_proj): model.parameters() order is the index space of the
        # reference's optimizer state and state_dict() its key set (checkpoint interop).  The pre-training heads
        # (quantizer, project_q, final_proj, ...) are never on the ST path; they are instantiated as inert parameters with the
        # reference's shapes so that real checkpoints (wav2vec_small: quantize_targets=True) load with strict=True through
        # ANY enclosing module, are re-emitted by state_dict(), and keep every later optimizer-state index where the
        # reference has it.
        self.quantizer = self.input_quantizer = self.project_inp = self.target_glu = None
        if getattr(args, "quantize_targets", False):
            vq_dim = args.latent_dim if getattr(args, "latent_dim", 0) > 0 else final_dim
            self.quantizer = _InertGumbelVectorQuantizer(self.embed, args.latent_vars, args.latent_groups, vq_dim)
            self.project_q = Linear(vq_dim, final_dim)
        else:
            self.project_q = Linear(self.embed, final_dim)
        if getattr(args, "quantize_input", False):
            if getattr(args, "same_quantizer", False) and self.quantizer is not None:
                vq_dim = final_dim
                self.input_quantizer = self.quantizer
            else:
                vq_dim = args.latent_dim if getattr(args, "latent_dim", 0) > 0 else args.encoder_embed_dim
                self.input_quantizer = _InertGumbelVectorQuantizer(self.embed, args.latent_vars, args.latent_groups, vq_dim)
            self.project_inp = Linear(vq_dim, args.encoder_embed_dim)
            raise NotImplementedError("quantize_input wav2vec2 models feed the quantised features to the encoder: not on the ST path")
        self.mask_emb = nn.Parameter(torch.FloatTensor(args.encoder_embed_dim).uniform_())
        self.encoder = TransformerEncoder(args)
        self.layer_norm = LayerNorm(self.embed)
        if getattr(args, "target_glu", False):
            self.target_glu = nn.Sequential(Linear(final_dim, final_dim * 2), nn.GLU())
        self.final_proj = Linear(args.encoder_embed_dim, final_dim)

    @classmethod
    def build_model(cls, args, task=None):
        base_architecture(args)
        return cls(args)

    def forward(self, source, padding_mask=None, mask=True, features_only=False):
        assert features_only and not mask, "only the features_only / mask=False path is built (SURVEY §2.1 #2)"
        if self.training:
            # pre-training-only parameters never receive a gradient on this path: without this the gradient bucket that holds
            # final_proj (the FIRST wav2vec2 bucket in backward order) would wait for finish() and un-overlap every later one
            heads = [self.mask_emb] + list(self.final_proj.parameters()) + list(self.project_q.parameters())
            for m in (self.quantizer, self.target_glu):
                if m is not None:
                    heads += list(m.parameters())
            notify_unused_parameters(heads)
        # the frame-level padding mask of :543-548 is a function of the sample-level one and the CNN's output length: computed ONCE,
        # before the CNN is queued (it serves the packing plan, the conv frame limits and the encoder's key mask)
        frame_mask, plan, nz_last = None, None, None
        if padding_mask is not None:
            t1 = self.feature_extractor.output_length(source.shape[1])
            pm = padding_mask
            extra = pm.size(1) % t1
            if extra > 0:
                pm = pm[:, :-extra]
            frame_mask = pm.view(pm.size(0), t1, -1).all(-1)
            if self.encoder.wants_packing(padding_mask):
                # the packing plan needs a few integers on the host (rows in total, longest sequence, lengths): read them HERE — the
                # stream is empty at this point of an update, later the read would drain 20 ms of queued convolutions
                plan = CF.plan_packed_rows(frame_mask, self.encoder.packing_margin())
            # frames past the last real one get a zero gradient (they are overwritten with zeros at the encoder input), which bounds
            # every conv layer's weight-gradient reduction and the frames the CNN has to compute at all
            pos = torch.arange(1, t1 + 1, device=frame_mask.device, dtype=torch.int32)
            nz_last = (pos * (~frame_mask)).amax(dim=1).to(torch.int32)
        self.last_plan = plan  # (the caller's own layer stack derives its packing plan from this one's host lengths)
        feats = self.feature_extractor(source, nz_last)  # [B, T1, C] channels-last
        if self.feature_grad_mult <= 0:
            feats = feats.detach()
        elif self.feature_grad_mult != 1.0:
            feats = _GradMultiply.apply(feats, self.feature_grad_mult)
        feats = self.layer_norm(feats)  # transpose(1,2) is free in channels-last (:539-540)
        if padding_mask is not None:
            assert feats.size(1) == frame_mask.size(1)
            padding_mask = frame_mask
        if self.post_extract_proj is not None:
            feats = self.post_extract_proj(feats)
        if self.training and self.dropout_input_p > 0:
            feats = CF.dropout(feats, self.dropout_input_p)  # self.dropout_input (:553)
        x, padding_mask = self.encoder(feats, padding_mask=padding_mask, plan=plan)
        return {"x": x, "padding_mask": padding_mask}

    def extract_features(self, source, padding_mask, mask=False):
        res = self.forward(source, padding_mask, mask=mask, features_only=True)
        return res["x"], res["padding_mask"]

class _InertGumbelVectorQuantizer(nn.Module):
    """Parameter shapes of modules/gumbel_vector_quantizer.py:12-76 (combine_groups=False, weight_proj_depth=1): `vars`
    [1, groups * num_vars, vq_dim / groups] and `weight_proj` Linear(dim, groups * num_vars).  Pre-training only (the
    contrastive targets): never called on the ST path."""

    def __init__(self, dim, num_vars, groups, vq_dim):
        super().__init__()
        assert vq_dim % groups == 0
        self.vars = nn.Parameter(torch.FloatTensor(1, groups * num_vars, vq_dim // groups).uniform_())
        self.weight_proj = nn.Linear(dim, groups * num_vars)
        nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
        nn.init.zeros_(self.weight_proj.bias)

    def forward(self, *a, **k):
        raise NotImplementedError("the wav2vec2 quantizer is a pre-training module; the ST path never calls it")


class _GradMultiply(torch.autograd.Function):
    """modules/grad_multiply.py."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


@register_model_architecture("wav2vec2", "wav2vec2")
def base_architecture(args):
    """wav2vec2.py:962-1029 defaults (only the ones the feature path reads, plus checkpoint-visible ones)."""
    d = dict(
        extractor_mode="default", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
        encoder_attention_heads=12, activation_fn="gelu", dropout=0.1, attention_dropout=0.1, activation_dropout=0.0,
        final_dim=0, layer_norm_first=False, encoder_layerdrop=0.0,
        conv_feature_layers="[(512, 10, 5)] + [(512, 8, 4)] + [(512, 4, 2)] * 3 + [(512, 1, 1)]",
        logit_temp=0.1, quantize_targets=False, quantize_input=False, same_quantizer=False, feature_grad_mult=1.0,
        latent_vars=320, latent_groups=2, latent_dim=0, dropout_input=0, dropout_features=0, conv_pos=128,
        conv_pos_groups=16, conv_bias=False, target_glu=False,
    )
    for k, v in d.items():
        if not hasattr(args, k):
            setattr(args, k, v)


def wav2vec_small_args(**over):
    """The published wav2vec_small.pt hyper-parameters (SURVEY §8 caveat): 7-layer stride-320 CNN, 12x768 encoder,
    feature_grad_mult 0.1, layerdrop 0.05."""
    ns = Namespace(
        conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2", encoder_layers=12, encoder_embed_dim=768,
        encoder_ffn_embed_dim=3072, encoder_attention_heads=12, activation_fn="gelu", dropout=0.1, attention_dropout=0.1,
        activation_dropout=0.0, encoder_layerdrop=0.05, feature_grad_mult=0.1, final_dim=256, quantize_targets=True,
        conv_pos=128, conv_pos_groups=16, layer_norm_first=False, extractor_mode="default", conv_bias=False,
        dropout_input=0.1, dropout_features=0.1)
    for k, v in over.items():
        setattr(ns, k, v)
    base_architecture(ns)
    return ns
