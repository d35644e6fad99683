"""Host-side mirror of the fairseq modules on the Chimera hot path (same class names, constructor
arguments, forward signatures, parameter names / state-dict keys and init), whose forward calls the
HIP kernels through functional.py instead of ATen.

Reference files mirrored (paths relative to the reference root):
  modules/layer_norm.py:30-35                     -> LayerNorm
  models/transformer.py:906-919                   -> Embedding, Linear
  modules/sinusoidal_positional_embedding.py      -> SinusoidalPositionalEmbedding
  modules/positional_embedding.py                 -> PositionalEmbedding
  modules/fairseq_dropout.py                      -> FairseqDropout
  modules/multihead_attention.py                  -> MultiheadAttention
  modules/transformer_layer.py                    -> TransformerEncoderLayer, TransformerDecoderLayer
  incremental_decoding_utils.py                   -> incremental-state helpers

Public tensors are Time x Batch x Channel like the reference.  Internally the storage is
batch-major [B, T, C]: a (T,B,C) tensor produced by these modules is a transposed VIEW of contiguous
[B,T,C] memory, so flattening tokens for the GEMMs never copies and the attention kernel gets
(batch, head, time) strides directly.
"""
import math
import uuid
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as CF


# ---------------------------------------------------------------------------------------------
def to_batch_major(x_tbc):
    """(T,B,C) -> contiguous [B,T,C] (free when x is a view of batch-major storage)."""
    xb = x_tbc.transpose(0, 1)
    return xb if xb.is_contiguous() else xb.contiguous()


def to_time_major_view(x_btc):
    return x_btc.transpose(0, 1)


class LayerNorm(nn.Module):
    """modules/layer_norm.py:30-35 — eps 1e-5, elementwise affine; forward = cst_layernorm_fwd."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
        super().__init__()
        self.normalized_shape = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
        self.eps = eps
        assert elementwise_affine
        self.weight = nn.Parameter(torch.ones(self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(self.normalized_shape))

    def forward(self, x):
        if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            # (T,B,C) view of batch-major storage: normalise the storage in place order, hand back the same view
            return CF.layer_norm(x.transpose(0, 1), self.weight, self.bias, self.eps).transpose(0, 1)
        return CF.layer_norm(x, self.weight, self.bias, self.eps)

    def forward_residual(self, x):
        """(LN(x), x') — x' aliases x and is what the caller must feed its residual connection, so that the LN gradient
        and the residual gradient are summed inside the LN backward kernel (functional._LayerNormPassFn)."""
        if x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            y, xp = CF.layer_norm_residual(x.transpose(0, 1), self.weight, self.bias, self.eps)
            return y.transpose(0, 1), xp.transpose(0, 1)
        return CF.layer_norm_residual(x, self.weight, self.bias, self.eps)


def Embedding(num_embeddings, embedding_dim, padding_idx):
    """models/transformer.py:906-911."""
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


class Linear(nn.Module):
    """nn.Linear with fairseq's init (models/transformer.py:913-919: xavier_uniform, bias 0);
    forward = cst_gemm with fused bias (+ activation, + residual)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        else:
            self.register_parameter("bias", None)

    def forward(self, x, act=None, resid=None, dropout_p=0.0):
        return CF.linear(x, self.weight, self.bias, act=act, resid=resid, dropout_p=dropout_p)


class FairseqDropout(nn.Module):
    """modules/fairseq_dropout.py.  p = 0 is the identity; p > 0 in training runs cst_dropout with a counter-based mask
    (rng.py) — the mask is a function of (update seed, site ordinal, element index), not of torch's generator."""

    def __init__(self, p, module_name=None):
        super().__init__()
        self.p = p
        self.module_name = module_name
        self.apply_during_inference = False

    def forward(self, x, inplace: bool = False):
        if self.p > 0 and (self.training or self.apply_during_inference):
            return CF.dropout(x, self.p)
        return x


def make_positions(tensor, padding_idx: int):
    """utils.py:235-245."""
    mask = tensor.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


class SinusoidalPositionalEmbedding(nn.Module):
    """modules/sinusoidal_positional_embedding.py:15-105 (tensor2tensor sin‖cos table, pad row zero)."""

    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx
        # the reference materialises init_size rows up front (2 GB+ at max_source_positions=1e6); grown lazily here
        self.weights = SinusoidalPositionalEmbedding.get_embedding(min(init_size, 4096), embedding_dim, padding_idx)
        self.register_buffer("_float_tensor", torch.FloatTensor(1))
        self.max_positions = int(1e5)

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        half_dim = embedding_dim // 2
        emb = math.log(10000) / (half_dim - 1)
        emb = torch.exp(torch.arange(half_dim, dtype=torch.float) * -emb)
        emb = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        if padding_idx is not None:
            emb[padding_idx, :] = 0
        return emb

    def table(self, seq_len, device):
        """The fp32 sin || cos table on `device` with at least padding_idx + 1 + seq_len rows (row padding_idx is zero) — the
        operand of cst_embed_pos_fwd, which evaluates make_positions itself and gathers the rows in-kernel."""
        need = self.padding_idx + 1 + seq_len
        t = getattr(self, "_dev_table", None)
        if t is None or t.shape[0] < need or t.device != device:
            rows = max(need, 256)
            rows = 1 << (rows - 1).bit_length()  # grow geometrically: one H2D copy per doubling, not per batch shape
            t = SinusoidalPositionalEmbedding.get_embedding(rows, self.embedding_dim, self.padding_idx).to(device=device, dtype=torch.float32).contiguous()
            self._dev_table = t
        return t

    def forward(self, input, incremental_state=None, timestep=None, positions=None):
        bsz, seq_len = input.shape[:2]
        max_pos = self.padding_idx + 1 + seq_len
        if self.weights is None or max_pos > self.weights.size(0):
            self.weights = SinusoidalPositionalEmbedding.get_embedding(max_pos, self.embedding_dim, self.padding_idx)
        self.weights = self.weights.to(self._float_tensor)
        if incremental_state is not None:
            pos = timestep.view(-1)[0] + 1 if timestep is not None else seq_len
            return self.weights[self.padding_idx + pos, :].expand(bsz, 1, -1)
        positions = make_positions(input, self.padding_idx)
        return self.weights.index_select(0, positions.view(-1)).view(bsz, seq_len, -1).detach()


def PositionalEmbedding(num_embeddings, embedding_dim, padding_idx, learned=False):
    """modules/positional_embedding.py — only the sinusoidal branch is on the Chimera path."""
    if learned:
        raise NotImplementedError("learned positional embeddings are not on the Chimera path (decoder_learned_pos=False)")
    return SinusoidalPositionalEmbedding(embedding_dim, padding_idx, init_size=num_embeddings + padding_idx + 1)


# ---------------------------------------------------------------------------------------------
class FairseqIncrementalState:
    """incremental_decoding_utils.py:13-51."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._incremental_state_id = str(uuid.uuid4())

    def _get_full_incremental_state_key(self, key):
        return "{}.{}".format(self._incremental_state_id, key)

    def get_incremental_state(self, incremental_state, key):
        full_key = self._get_full_incremental_state_key(key)
        if incremental_state is None or full_key not in incremental_state:
            return None
        return incremental_state[full_key]

    def set_incremental_state(self, incremental_state, key, value):
        if incremental_state is not None:
            incremental_state[self._get_full_incremental_state_key(key)] = value
        return incremental_state


def causal_mask(dim, device, dtype=torch.float32):
    """models/transformer.py:844-856 buffered_future_mask; tagged so MultiheadAttention uses the kernel's causal mode."""
    m = torch.triu(torch.full((dim, dim), float("-inf"), device=device, dtype=dtype), 1)
    m.cst_kind = "causal"
    return m


def _classify_attn_mask(attn_mask):
    """Map a reference-style additive/boolean attn_mask onto what the fused kernel supports."""
    kind = getattr(attn_mask, "cst_kind", None)
    if kind is not None:
        return kind, None
    m = attn_mask
    tq, tk = m.shape
    blocked = (m != 0) if m.dtype != torch.bool else m
    if tq == tk and torch.equal(blocked, torch.triu(torch.ones_like(blocked), 1)):
        return "causal", None
    if bool((blocked == blocked[0:1]).all()):
        return "columns", blocked[0]  # same columns masked for every query (the memory mask, Q2)
    raise NotImplementedError("chimera-st_amd fused attention supports causal and column-only attn_mask patterns")


class MultiheadAttention(FairseqIncrementalState, nn.Module):
    """modules/multihead_attention.py:22-488.  Same parameters (q_proj/k_proj/v_proj/out_proj) and forward
    signature; scores/softmax/PV run in the fused flash kernel (cst_attn_fwd/bwd), projections in cst_gemm."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True, add_bias_kv=False,
                 add_zero_attn=False, self_attention=False, encoder_decoder_attention=False, q_noise=0.0, qn_block_size=8):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout_module = FairseqDropout(dropout, module_name=self.__class__.__name__)
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        assert not self.self_attention or self.qkv_same_dim
        assert not add_bias_kv and not add_zero_attn, "bias_kv / zero_attn are not on the Chimera path"
        self.k_proj = Linear(self.kdim, embed_dim, bias=bias)
        self.v_proj = Linear(self.vdim, embed_dim, bias=bias)
        self.q_proj = Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = Linear(embed_dim, embed_dim, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        if self.qkv_same_dim:  # multihead_attention.py:97-106
            nn.init.xavier_uniform_(self.k_proj.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.v_proj.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.q_proj.weight, gain=1 / math.sqrt(2))
        else:
            nn.init.xavier_uniform_(self.k_proj.weight)
            nn.init.xavier_uniform_(self.v_proj.weight)
            nn.init.xavier_uniform_(self.q_proj.weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, key_padding_mask=None, incremental_state=None, need_weights=True,
                static_kv=False, attn_mask=None, before_softmax=False, need_head_weights=False, resid=None, out_dropout_p=0.0, seq=None):
        """Input shape: Time x Batch x Channel.  Returns (attn [T,B,C], None).
        `resid` / `out_dropout_p` (extensions): out = resid + dropout(out_proj(attn)) inside out_proj's GEMM epilogue — the
        caller's `residual + self.dropout_module(x)` (transformer_layer.py:139-141) without a separate pass.
        Attention-probability dropout (self.dropout_module, :359) runs inside the attention kernels.
        `seq` (extension): a functional.PackedRows plan — query is [rows, 1, C], the padding-free packing of a right-padded batch;
        every sequence attends to its own real frames (what key_padding_mask expresses for the padded batch)."""
        if need_head_weights or before_softmax:
            raise NotImplementedError("attention weights never leave the fused kernel (need_weights is ignored)")
        attn_p = self.dropout_module.p if (self.training and self.dropout_module.p > 0) else 0.0
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim
        qb = to_batch_major(query)  # [B,Tq,C]
        causal = False
        if attn_mask is not None:
            kind, cols = _classify_attn_mask(attn_mask)
            if kind == "causal":
                causal = True
            else:
                cm = cols.view(1, -1).expand(bsz, -1)
                key_padding_mask = cm if key_padding_mask is None else (key_padding_mask.bool() | cm)

        saved_state = None
        if incremental_state is not None:
            saved_state = self._get_input_buffer(incremental_state)
            if saved_state is not None and "prev_key" in saved_state and static_kv:
                assert self.encoder_decoder_attention and not self.self_attention
                key = value = None

        assert seq is None or (self.self_attention and saved_state is None), "packed rows are a self-attention training path"
        if self.self_attention and saved_state is None:
            # packed projection: one [3C, C] GEMM (and one dX / dW GEMM in backward) instead of three; the fused attention
            # kernel reads q | k | v as channel slices of the [B, T, 3C] result and writes dq | dk | dv the same way.
            # (a view of the flat parameter buffer when the trainer laid q | k | v out back to back, a torch.cat otherwise)
            w = CF.stacked_rows(self.q_proj.weight, self.k_proj.weight, self.v_proj.weight)
            bqkv = (CF.stacked_rows(self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)
                    if self.q_proj.bias is not None else None)
            resid_b = to_batch_major(resid) if resid is not None else None
            if resid is query and torch.is_grad_enabled() and qb.requires_grad:
                qkv, resid_b = CF.linear_pass(qb, w, bqkv)  # post-norm block: residual gradient joins the packed dX GEMM
            else:
                qkv = CF.linear(qb, w, bqkv)
            if key_padding_mask is not None and key_padding_mask.dim() == 0:
                key_padding_mask = None
            attn = CF.attention_packed(qkv, self.num_heads, None if seq is not None else key_padding_mask, causal, self.scaling,
                                       dropout_p=attn_p, seq=seq)
            out = self.out_proj(attn, resid=resid_b, dropout_p=out_dropout_p)
            return to_time_major_view(out), None
        q = self.q_proj(qb)
        k = v = None
        if (saved_state is None and not self.self_attention and key is not None and value is key and self.kdim == self.vdim
                and (key_padding_mask is None or key_padding_mask.dim() != 0)):
            # training-time cross-attention: k_proj and v_proj read the same rows — ONE [2C, C] projection (a view of the flat
            # parameter buffer when the trainer laid k | v out back to back), the kernel reads k | v as channel halves and the
            # backward writes dk | dv into one buffer (one dX GEMM, one dW GEMM, one gradient for the encoder output per layer)
            wkv = CF.stacked_rows(self.k_proj.weight, self.v_proj.weight)
            bkv = CF.stacked_rows(self.k_proj.bias, self.v_proj.bias) if self.k_proj.bias is not None else None
            kv = CF.linear(to_batch_major(key), wkv, bkv)
            if key_padding_mask is not None:
                assert key_padding_mask.size(0) == bsz and key_padding_mask.size(1) == kv.size(1)
            attn = CF.attention_kv(q, kv, self.num_heads, key_padding_mask, self.scaling, dropout_p=attn_p)
            assert not causal
            out = self.out_proj(attn, resid=to_batch_major(resid) if resid is not None else None, dropout_p=out_dropout_p)
            return to_time_major_view(out), None
        if self.self_attention:
            k, v = self.k_proj(qb), self.v_proj(qb)
        elif key is not None:
            kb = to_batch_major(key)
            vb = kb if value is key else to_batch_major(value)
            k, v = self.k_proj(kb), self.v_proj(vb)

        if saved_state is not None:
            # caches are kept batch-major [B, T, C] (the reference keeps [B,H,T,D]; same data, kernel-friendly layout)
            if "prev_key" in saved_state:
                pk, pv = saved_state["prev_key"], saved_state["prev_value"]
                if static_kv:
                    k, v = pk, pv
                else:
                    k, v = torch.cat([pk, k], dim=1), torch.cat([pv, v], dim=1)
            prev_kpm = saved_state.get("prev_key_padding_mask", None)
            key_padding_mask = MultiheadAttention._append_prev_key_padding_mask(
                key_padding_mask, prev_kpm, bsz, k.size(1), static_kv)
            saved_state["prev_key"], saved_state["prev_value"] = k, v
            saved_state["prev_key_padding_mask"] = key_padding_mask
            incremental_state = self._set_input_buffer(incremental_state, saved_state)
            causal = False  # single-step decode: the query is the newest position
        assert k is not None and v is not None
        if key_padding_mask is not None and key_padding_mask.dim() == 0:
            key_padding_mask = None
        if key_padding_mask is not None:
            assert key_padding_mask.size(0) == bsz and key_padding_mask.size(1) == k.size(1)
        # q is NOT pre-scaled (reference: q *= scaling, :225); the kernel applies `scale` to QK^T in fp32.
        attn = CF.attention(q, k, v, self.num_heads, key_padding_mask, causal, self.scaling, "bt", "bt", dropout_p=attn_p)
        out = self.out_proj(attn, resid=to_batch_major(resid) if resid is not None else None, dropout_p=out_dropout_p)
        return to_time_major_view(out), None

    @staticmethod
    def _append_prev_key_padding_mask(key_padding_mask, prev_key_padding_mask, batch_size, src_len, static_kv):
        """multihead_attention.py:381-417."""
        if prev_key_padding_mask is not None and static_kv:
            return prev_key_padding_mask
        if prev_key_padding_mask is not None and key_padding_mask is not None:
            return torch.cat([prev_key_padding_mask.float(), key_padding_mask.float()], dim=1).bool()
        if prev_key_padding_mask is not None:
            filler = torch.zeros((batch_size, src_len - prev_key_padding_mask.size(1)), device=prev_key_padding_mask.device)
            return torch.cat([prev_key_padding_mask.float(), filler.float()], dim=1).bool()
        if key_padding_mask is not None:
            filler = torch.zeros((batch_size, src_len - key_padding_mask.size(1)), device=key_padding_mask.device)
            return torch.cat([filler.float(), key_padding_mask.float()], dim=1).bool()
        return None

    def reorder_incremental_state(self, incremental_state, new_order):
        """multihead_attention.py:419-437 (index_select on the cached K/V)."""
        input_buffer = self._get_input_buffer(incremental_state)
        if input_buffer is not None:
            for k in input_buffer.keys():
                b = input_buffer[k]
                if b is not None:
                    if self.encoder_decoder_attention and b.size(0) == new_order.size(0):
                        break
                    input_buffer[k] = b.index_select(0, new_order)
            incremental_state = self._set_input_buffer(incremental_state, input_buffer)
        return incremental_state

    def _get_input_buffer(self, incremental_state):
        result = self.get_incremental_state(incremental_state, "attn_state")
        return result if result is not None else {}

    def _set_input_buffer(self, incremental_state, buffer):
        return self.set_incremental_state(incremental_state, "attn_state", buffer)

    def upgrade_state_dict_named(self, state_dict, name):
        """multihead_attention.py:459-488: split a legacy in_proj_weight/bias into q/k/v."""
        prefix = name + "." if name != "" else ""
        items_to_add, keys_to_remove = {}, []
        for k in state_dict.keys():
            if k.endswith(prefix + "in_proj_weight"):
                dim = int(state_dict[k].shape[0] / 3)
                items_to_add[prefix + "q_proj.weight"] = state_dict[k][:dim]
                items_to_add[prefix + "k_proj.weight"] = state_dict[k][dim:2 * dim]
                items_to_add[prefix + "v_proj.weight"] = state_dict[k][2 * dim:]
                keys_to_remove.append(k)
                k_bias = prefix + "in_proj_bias"
                if k_bias in state_dict.keys():
                    dim = int(state_dict[k].shape[0] / 3)
                    items_to_add[prefix + "q_proj.bias"] = state_dict[k_bias][:dim]
                    items_to_add[prefix + "k_proj.bias"] = state_dict[k_bias][dim:2 * dim]
                    items_to_add[prefix + "v_proj.bias"] = state_dict[k_bias][2 * dim:]
                    keys_to_remove.append(prefix + "in_proj_bias")
        for k in keys_to_remove:
            del state_dict[k]
        for key, value in items_to_add.items():
            state_dict[key] = value


# ---------------------------------------------------------------------------------------------
def _act_name(fn):
    if fn in ("relu", "gelu"):
        return fn
    raise NotImplementedError("activation %r is not on the Chimera path (relu / gelu only)" % fn)


class TransformerEncoderLayer(nn.Module):
    """modules/transformer_layer.py:19-155 (pre-/post-norm).  fc1's bias+activation and fc2's bias+residual are GEMM
    epilogues; the attention residual is out_proj's epilogue."""

    def __init__(self, args):
        super().__init__()
        self.embed_dim = args.encoder_embed_dim
        self.self_attn = MultiheadAttention(self.embed_dim, args.encoder_attention_heads,
                                            dropout=args.attention_dropout, self_attention=True)
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        self.dropout_module = FairseqDropout(args.dropout, module_name=self.__class__.__name__)
        self.activation_fn = _act_name(getattr(args, "activation_fn", "relu"))
        activation_dropout_p = getattr(args, "activation_dropout", 0)
        if activation_dropout_p == 0:
            activation_dropout_p = getattr(args, "relu_dropout", 0)
        self.activation_dropout_module = FairseqDropout(float(activation_dropout_p), module_name=self.__class__.__name__)
        self.normalize_before = args.encoder_normalize_before
        self.fc1 = Linear(self.embed_dim, args.encoder_ffn_embed_dim)
        self.fc2 = Linear(args.encoder_ffn_embed_dim, self.embed_dim)
        self.final_layer_norm = LayerNorm(self.embed_dim)

    def _drop_ps(self):
        """(dropout, activation_dropout) probabilities in effect: every dropout of the layer is a GEMM epilogue."""
        if not self.training:
            return 0.0, 0.0
        return float(self.dropout_module.p), float(self.activation_dropout_module.p)

    def forward(self, x, encoder_padding_mask, attn_mask: Optional[torch.Tensor] = None, kv=None, seq=None):
        """x: (T,B,C).  `kv` (extension used by the memory module): separate key/value rows (Tk,B,C) that go through
        the same self_attn_layer_norm, i.e. exactly the rows of cat(h_enc, memory) the masked reference attends to."""
        residual = x
        if self.normalize_before:
            h, residual = self.self_attn_layer_norm.forward_residual(x)
            hk = self.self_attn_layer_norm(kv) if kv is not None else h
        else:
            h, hk = x, (kv if kv is not None else x)
        p_drop, p_act = self._drop_ps()
        if kv is None:
            # seq: x holds packed rows [rows, 1, C] (functional.PackedRows); every sequence attends to its own rows only
            x, _ = self.self_attn(h, h, h, key_padding_mask=None if seq is not None else encoder_padding_mask, attn_mask=attn_mask,
                                  resid=residual, out_dropout_p=p_drop, seq=seq)  # residual + dropout(attn)
        else:
            x, _ = self._cross(h, hk, encoder_padding_mask, residual, p_drop)
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        residual = x
        if self.normalize_before:
            h, residual = self.final_layer_norm.forward_residual(x)
        else:
            h = x
        x = to_time_major_view(CF.ffn(to_batch_major(h), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                                      self.activation_fn, resid=to_batch_major(residual), activation_dropout_p=p_act,
                                      dropout_p=p_drop))
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x

    def _cross(self, q_in, kv_in, key_padding_mask, resid, out_dropout_p=0.0):
        """self_attn's own projections with separate query / key-value rows (memory layers)."""
        sa = self.self_attn
        qb, kb = to_batch_major(q_in), to_batch_major(kv_in)
        attn_p = sa.dropout_module.p if (self.training and sa.dropout_module.p > 0) else 0.0
        q = sa.q_proj(qb)
        # k | v as ONE projection of the key/value rows (adjacent in the flat parameter buffer behind q: a view)
        wkv = CF.stacked_rows(sa.k_proj.weight, sa.v_proj.weight)
        bkv = CF.stacked_rows(sa.k_proj.bias, sa.v_proj.bias) if sa.k_proj.bias is not None else None
        attn = CF.attention_kv(q, CF.linear(kb, wkv, bkv), sa.num_heads, key_padding_mask, sa.scaling, dropout_p=attn_p)
        out = sa.out_proj(attn, resid=to_batch_major(resid) if resid is not None else None, dropout_p=out_dropout_p)
        return to_time_major_view(out), None

    def upgrade_state_dict_named(self, state_dict, name):
        layer_norm_map = {"0": "self_attn_layer_norm", "1": "final_layer_norm"}
        for old, new in layer_norm_map.items():
            for m in ("weight", "bias"):
                k = "{}.layer_norms.{}.{}".format(name, old, m)
                if k in state_dict:
                    state_dict["{}.{}.{}".format(name, new, m)] = state_dict[k]
                    del state_dict[k]


class TransformerDecoderLayer(nn.Module):
    """modules/transformer_layer.py:158-423 (pre-norm on the Chimera path)."""

    def __init__(self, args, no_encoder_attn=False, add_bias_kv=False, add_zero_attn=False):
        super().__init__()
        self.embed_dim = args.decoder_embed_dim
        self.dropout_module = FairseqDropout(args.dropout, module_name=self.__class__.__name__)
        self.cross_self_attention = getattr(args, "cross_self_attention", False)
        assert not self.cross_self_attention
        self.self_attn = MultiheadAttention(self.embed_dim, args.decoder_attention_heads, dropout=args.attention_dropout,
                                            self_attention=True)
        self.activation_fn = _act_name(getattr(args, "activation_fn", "relu"))
        activation_dropout_p = getattr(args, "activation_dropout", 0)
        if activation_dropout_p == 0:
            activation_dropout_p = getattr(args, "relu_dropout", 0)
        self.activation_dropout_module = FairseqDropout(float(activation_dropout_p), module_name=self.__class__.__name__)
        self.normalize_before = args.decoder_normalize_before
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        if no_encoder_attn:
            self.encoder_attn = None
            self.encoder_attn_layer_norm = None
        else:
            self.encoder_attn = MultiheadAttention(
                self.embed_dim, args.decoder_attention_heads, kdim=getattr(args, "encoder_embed_dim", None),
                vdim=getattr(args, "encoder_embed_dim", None), dropout=args.attention_dropout, encoder_decoder_attention=True)
            self.encoder_attn_layer_norm = LayerNorm(self.embed_dim)
        self.fc1 = Linear(self.embed_dim, args.decoder_ffn_embed_dim)
        self.fc2 = Linear(args.decoder_ffn_embed_dim, self.embed_dim)
        self.final_layer_norm = LayerNorm(self.embed_dim)
        self.need_attn = True
        self.onnx_trace = False

    def _drop_ps(self):
        if not self.training:
            return 0.0, 0.0
        return float(self.dropout_module.p), float(self.activation_dropout_module.p)

    def forward(self, x, encoder_out=None, encoder_padding_mask=None, incremental_state=None, prev_self_attn_state=None,
                prev_attn_state=None, self_attn_mask=None, self_attn_padding_mask=None, need_attn=False,
                need_head_weights=False):
        p_drop, p_act = self._drop_ps()
        residual = x
        if self.normalize_before:
            h, residual = self.self_attn_layer_norm.forward_residual(x)
        else:
            h = x
        x, _ = self.self_attn(query=h, key=h, value=h, key_padding_mask=self_attn_padding_mask,
                              incremental_state=incremental_state, need_weights=False, attn_mask=self_attn_mask,
                              resid=residual, out_dropout_p=p_drop)
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        if self.encoder_attn is not None and encoder_out is not None:
            residual = x
            if self.normalize_before:
                h, residual = self.encoder_attn_layer_norm.forward_residual(x)
            else:
                h = x
            x, _ = self.encoder_attn(query=h, key=encoder_out, value=encoder_out, key_padding_mask=encoder_padding_mask,
                                     incremental_state=incremental_state, static_kv=True, need_weights=False,
                                     resid=residual, out_dropout_p=p_drop)
            if not self.normalize_before:
                x = self.encoder_attn_layer_norm(x)
        residual = x
        if self.normalize_before:
            h, residual = self.final_layer_norm.forward_residual(x)
        else:
            h = x
        x = to_time_major_view(CF.ffn(to_batch_major(h), self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                                      self.activation_fn, resid=to_batch_major(residual), activation_dropout_p=p_act,
                                      dropout_p=p_drop))
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x, None, None

    def make_generation_fast_(self, need_attn: bool = False, **kwargs):
        self.need_attn = need_attn
