"""ORACLE — test infrastructure, NOT product code.

CPU fp32 restatement of the Chimera-ST training/decoding hot path, written as
explicit functions over a fairseq-named state dict.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product (chimera-st_amd/) never does.

Why torch-CPU and not numpy: on this path the reference's arithmetic IS torch ATen on CPU
("third-party arithmetic"; torch is un-pinned by the reference, the in-container
torch 2.10 stands in).  Every op below is spelled out from primitive ATen calls (matmul, softmax,
conv1d, mean/var) following the reference source lines cited per function; autograd of these
primitives supplies the gradients.  This is the "torch fp32 reference for a floating-point
kernel" the brief allows, arranged as the reference arranges the computation.

PINNING: tests/test_oracle_golden.py checks every function here against fixtures produced by
running the real reference in the build container (tools/ref_harness/make_goldens.py):
per-stage activations, logits, memory, all loss terms, every parameter gradient, two optimizer
updates, and greedy/beam-5 decode results.  The reference itself has no tests, so
those fixtures are the only pin ("parity unpinned by the reference's own tests").

Layout conventions follow the reference: T x B x C inside the Transformer stacks.
All file:line citations are relative to /root/reference.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]

NEG_INF = float("-inf")

# Storage-rounding emulation (tests/test_fullsize_gpu.py, bf16 check): with STORAGE = torch.bfloat16 every tensor the HIP path
# keeps in HBM in the storage dtype — projection / conv / LayerNorm / activation outputs, residual sums, attention
# probabilities and outputs, embeddings — is rounded to that dtype here too, forward AND backward (the gradient of a stored
# tensor is itself stored), while every reduction stays fp32 exactly as the kernels accumulate.  None (default) = the plain
# fp32 restatement that the golden fixtures pin; the emulation shows how much of a bf16-vs-fp32 gap is storage rounding.
STORAGE = None


class _RoundStored(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(STORAGE).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(STORAGE).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(STORAGE).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _st(x):
    return x if STORAGE is None else _RoundStored.apply(x)


# ReLU is not differentiable at 0: two correct fp32 evaluations of the same pre-activation can land on either side of it (their
# last bits depend on summation order) and then disagree on that neuron's weight-gradient row by a whole dh * x term.  With
# RELU_TAPS = {} the oracle records, per fc1 (tag = parameter prefix), the smallest |pre-activation| each neuron saw over all
# tokens, so that a parity test can tell such ties from errors (tests/parity_util.py::assert_grads_close_fp32).
RELU_TAPS = None
RELU_FULL = None  # {} -> also keep every pre-activation [tokens, neurons] per fc1 (tests/parity_util.py::detie picks bias nudges from them)


def _relu(x, tag):
    if RELU_TAPS is not None:
        z = x.detach().reshape(-1, x.shape[-1])
        m = z.abs().amin(dim=0)
        RELU_TAPS[tag] = torch.minimum(RELU_TAPS[tag], m) if tag in RELU_TAPS else m
        if RELU_FULL is not None:
            RELU_FULL[tag] = torch.cat((RELU_FULL[tag], z), 0) if tag in RELU_FULL else z
    return torch.relu(x)


def _st_fwd(x):
    """Rounded where it is consumed (MFMA operand), but its gradient never leaves fp32 registers (attention's dP)."""
    return x if STORAGE is None else _RoundFwd.apply(x)


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def lengths_to_padding_mask(lens: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    """fairseq/data/data_utils.py:491-495 — True where t >= len."""
    bsz = lens.size(0)
    m = int(lens.max().item()) if max_len is None else max_len
    ar = torch.arange(m, device=lens.device).view(1, m).expand(bsz, -1)
    return ar >= lens.view(bsz, 1).expand(-1, m)


def layer_norm(x, w, b, eps=1e-5):
    """modules/layer_norm.py:30-35 (torch.nn.LayerNorm, eps 1e-5, affine), written out."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return _st((x - mu) * torch.rsqrt(var + eps) * w + b)


def gelu(x):
    """modules/gelu.py:25 / nn.GELU — exact erf form."""
    return _st(0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0)))))


def linear(x, w, b=None):
    y = x.matmul(w.t())
    return _st(y if b is None else y + b)


def sinusoidal_table(num: int, dim: int, padding_idx: Optional[int]) -> torch.Tensor:
    """modules/sinusoidal_positional_embedding.py:36-58 (tensor2tensor layout: sin ‖ cos)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(num, 1)], dim=1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


def make_positions(tokens_or_mask: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """utils.py:235-245 — positions start at padding_idx+1, pads get padding_idx."""
    mask = tokens_or_mask.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def positional_embedding(tokens_or_mask: torch.Tensor, dim: int, padding_idx: int = 1) -> torch.Tensor:
    """modules/sinusoidal_positional_embedding.py:60-105 (non-incremental branch) -> [B,T,dim]."""
    bsz, seq = tokens_or_mask.shape
    tab = sinusoidal_table(padding_idx + 1 + seq, dim, padding_idx)
    pos = make_positions(tokens_or_mask, padding_idx)
    return tab.index_select(0, pos.view(-1)).view(bsz, seq, -1)


# --------------------------------------------------------------------------------------------
# multi-head attention — the in-tree arithmetic (modules/multihead_attention.py:189-379), which
# F.multi_head_attention_forward (the training fast path, :155-187) computes identically.
# --------------------------------------------------------------------------------------------
def mha(p: P, pre: str, query, key, value, num_heads: int,
        key_padding_mask: Optional[torch.Tensor] = None,
        attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """query [Tq,B,C], key/value [Tk,B,C] -> [Tq,B,C].
    q,k,v = xW+b (:205-224); q *= hd^-0.5 (:225); heads folded into batch (:238-254);
    scores = QK^T (:326) + attn_mask (:331-335); key padding -> -inf (:337-349);
    softmax in fp32 (:354-357); PV (:361); out_proj (:370)."""
    tq, bsz, c = query.shape
    tk = key.shape[0]
    hd = c // num_heads
    q = linear(query, p[pre + "q_proj.weight"], p[pre + "q_proj.bias"]) * (hd ** -0.5)
    k = linear(key, p[pre + "k_proj.weight"], p[pre + "k_proj.bias"])
    v = linear(value, p[pre + "v_proj.weight"], p[pre + "v_proj.bias"])
    q = q.contiguous().view(tq, bsz * num_heads, hd).transpose(0, 1)
    k = k.contiguous().view(tk, bsz * num_heads, hd).transpose(0, 1)
    v = v.contiguous().view(tk, bsz * num_heads, hd).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        w = w + attn_mask.unsqueeze(0).to(w.dtype)
    if key_padding_mask is not None:
        w = w.view(bsz, num_heads, tq, tk).masked_fill(
            key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), NEG_INF).view(bsz * num_heads, tq, tk)
    w = _st_fwd(torch.softmax(w.float(), dim=-1).type_as(w))  # the kernels feed P to the PV MFMA in the storage dtype
    a = _st(torch.bmm(w, v))
    a = a.transpose(0, 1).contiguous().view(tq, bsz, c)
    return linear(a, p[pre + "out_proj.weight"], p[pre + "out_proj.bias"])


# --------------------------------------------------------------------------------------------
# wav2vec2 feature path (features_only) — models/wav2vec/wav2vec2.py
# --------------------------------------------------------------------------------------------
def conv_feature_extractor(p: P, pre: str, wav: torch.Tensor, conv_layers: List[Tuple[int, int, int]]):
    """ConvFeatureExtractionModel.forward (wav2vec2.py:755-763), mode "default":
    Conv1d(no bias) -> [Fp32GroupNorm(C groups) on layer 0 only] -> GELU (:697-753;
    modules/fp32_group_norm.py:13-25).  wav [B,S] -> [B,C,T1]."""
    x = wav.unsqueeze(1)
    for i, (dim, k, s) in enumerate(conv_layers):
        x = F.conv1d(x, p["%sconv_layers.%d.0.weight" % (pre, i)], None, stride=s)
        if i > 0:
            x = _st(x)  # layer 0 keeps conv -> GroupNorm -> GELU in fp32 registers (cst_conv0_gn_gelu_fwd)
        if i == 0:
            # GroupNorm with one channel per group: statistics per (b, c) over time, fp32, eps 1e-5
            mu = x.mean(-1, keepdim=True)
            var = ((x - mu) ** 2).mean(-1, keepdim=True)
            x = (x - mu) * torch.rsqrt(var + 1e-5)
            x = x * p[pre + "conv_layers.0.2.weight"].view(1, -1, 1) + p[pre + "conv_layers.0.2.bias"].view(1, -1, 1)
        x = gelu(x)
    return x


class _GradMultiply(torch.autograd.Function):
    """modules/grad_multiply.py — identity forward, grad * scale backward."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.new(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def downsample_padding_mask(padding_mask: torch.Tensor, t1: int) -> torch.Tensor:
    """wav2vec2.py:543-548 — frame is pad iff all of its floor(S/T1) samples are pad."""
    extra = padding_mask.size(1) % t1
    if extra > 0:
        padding_mask = padding_mask[:, :-extra]
    return padding_mask.view(padding_mask.size(0), t1, -1).all(-1)


def pos_conv_weight(p: P, pre: str) -> torch.Tensor:
    """nn.utils.weight_norm(conv, name="weight", dim=2) (wav2vec2.py:785): w = g * v / ||v||,
    norm taken over dims (0,1) for every kernel tap."""
    v = p[pre + "pos_conv.0.weight_v"]
    g = p[pre + "pos_conv.0.weight_g"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def w2v2_sentence_layer(p: P, pre: str, x, padding_mask, heads: int):
    """TransformerSentenceEncoderLayer.forward, post-norm branch (wav2vec2.py:937-957)."""
    res = x
    x = mha(p, pre + "self_attn.", x, x, x, heads, key_padding_mask=padding_mask)
    x = _st(res + x)
    x = layer_norm(x, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
    res = x
    x = gelu(linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"]))
    x = linear(x, p[pre + "fc2.weight"], p[pre + "fc2.bias"])
    x = _st(res + x)
    return layer_norm(x, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])


def w2v2_extract_features(p: P, pre: str, wav, padding_mask, cfg) -> Tuple[torch.Tensor, torch.Tensor, dict]:
    """Wav2Vec2Model.forward(mask=False, features_only=True) (wav2vec2.py:527-586) followed by
    TransformerEncoder.extract_features (:818-845), layer_norm_first=False, layerdrop 0.
    Returns x [B,T1,C], frame padding mask [B,T1], and intermediates."""
    inter = {}
    feats = conv_feature_extractor(p, pre + "feature_extractor.", wav, cfg["conv_layers"])
    if cfg.get("feature_grad_mult", 1.0) != 1.0:
        feats = _GradMultiply.apply(feats, cfg["feature_grad_mult"])  # :529-532
    inter["w2v_cnn"] = feats
    feats = feats.transpose(1, 2)
    feats = layer_norm(feats, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])  # :540
    inter["w2v_ln"] = feats
    pm = downsample_padding_mask(padding_mask, feats.size(1)) if padding_mask is not None else None
    if (pre + "post_extract_proj.weight") in p:
        feats = linear(feats, p[pre + "post_extract_proj.weight"], p[pre + "post_extract_proj.bias"])  # :550-551
    inter["w2v_proj"] = feats
    x = feats
    e = pre + "encoder."
    if pm is not None:
        x = x.masked_fill(pm.unsqueeze(-1), 0.0)  # :820-821 (x[padding_mask] = 0)
    kpos = cfg["conv_pos"]
    xc = F.conv1d(x.transpose(1, 2), pos_conv_weight(p, e), p[e + "pos_conv.0.bias"],
                  padding=kpos // 2, groups=cfg["conv_pos_groups"])  # :773-779
    if kpos % 2 == 0:
        xc = xc[:, :, :-1]  # SamePad (modules/same_pad.py)
    x = _st(x + gelu(xc).transpose(1, 2))  # :823-825
    x = layer_norm(x, p[e + "layer_norm.weight"], p[e + "layer_norm.bias"])  # :827-828
    x = x.transpose(0, 1)
    for i in range(cfg["w2v_layers"]):
        x = w2v2_sentence_layer(p, "%slayers.%d." % (e, i), x, pm, cfg["w2v_heads"])
    x = x.transpose(0, 1)
    inter["w2v_out"] = x
    return x, pm, inter


# --------------------------------------------------------------------------------------------
# S2T / Chimera encoder
# --------------------------------------------------------------------------------------------
def conv1d_subsampler(p: P, pre: str, x, lengths, n_layers=2):
    """Conv1dSubsampler.forward (models/speech_to_text/s2t_transformer.py:69-77):
    [B,T,C] -> 2x (Conv1d k s2 pad k//2 + GLU over channels) -> [T2,B,C']; lengths :63-67."""
    x = x.transpose(1, 2).contiguous()
    for i in range(n_layers):
        w = p["%sconv_layers.%d.weight" % (pre, i)]
        x = _st(F.conv1d(x, w, p["%sconv_layers.%d.bias" % (pre, i)], stride=2, padding=w.size(2) // 2))
        a, g = x.chunk(2, dim=1)
        x = _st(a * torch.sigmoid(g))  # F.glu(dim=1)
    out = lengths.clone()
    for _ in range(n_layers):
        out = ((out.float() - 1) / 2 + 1).floor().long()
    return x.transpose(1, 2).transpose(0, 1).contiguous(), out


def encoder_layer(p: P, pre: str, x, padding_mask, heads: int, attn_mask=None, kv=None):
    """TransformerEncoderLayer.forward, pre-norm (modules/transformer_layer.py:105-155);
    attn_mask 1 -> -1e8 (:126-127).  `kv`: optional separate key/value rows (memory module)."""
    if attn_mask is not None:
        attn_mask = attn_mask.masked_fill(attn_mask.to(torch.bool), -1e8)
    res = x
    h = layer_norm(x, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
    h = mha(p, pre + "self_attn.", h, h, h, heads, key_padding_mask=padding_mask, attn_mask=attn_mask)
    x = _st(res + h)
    res = x
    h = layer_norm(x, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
    h = _relu(linear(h, p[pre + "fc1.weight"], p[pre + "fc1.bias"]), pre + "fc1")
    h = linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])
    return _st(res + h)


def audio_frontend(p: P, wav, src_lengths, cfg):
    """_get_w2v_feature (models/chimera/w2v2_transformer.py:319-336) + subsample (:350 / :228)."""
    pm = lengths_to_padding_mask(src_lengths, max_len=wav.size(1))
    feat, pm1, inter = w2v2_extract_features(p, "encoder.wav2vec_model.", wav, pm, cfg)
    out_len = (1 - pm1.int()).sum(dim=1)
    x, lens = conv1d_subsampler(p, "encoder.subsample.", feat, out_len)
    inter["subsample"] = x
    return x, lens, inter


def chimera_encoder(p: P, src_tokens, src_lengths, cfg):
    """S2T_W2V2_TransformerInterlinguaEncoder.forward (w2v2_transformer_interlingua.py:207-312).
    Returns memory [M,B,C] and intermediates."""
    d, heads = cfg["d"], cfg["heads"]
    is_text = not src_tokens.dtype.is_floating_point
    inter = {}
    if is_text:
        feat = F.embedding(src_tokens, p["encoder.text_embed_tokens.weight"], padding_idx=1).transpose(0, 1)  # :216
        lens = src_lengths
    else:
        feat, lens, inter = audio_frontend(p, src_tokens, src_lengths, cfg)
    x = _st(math.sqrt(d) * feat)  # :231
    pm = lengths_to_padding_mask(lens, max_len=x.size(0))
    if is_text:  # Q3: only text gets positions (:233-236)
        x = _st(x + positional_embedding(pm, d, 1).transpose(0, 1))
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(p, "encoder.transformer_layers.%d." % i, x, pm, heads)
    inter["enc_layer_last"] = x
    x = layer_norm(x, p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])  # :254-255
    inter["enc_ln"] = x
    h_enc = x
    L, B, _ = x.shape
    mem = p["encoder.interlingua_embedding.weight"].unsqueeze(1).repeat(1, B, 1)  # :268-269
    M = mem.size(0)
    attn_mask = torch.ones(L + M, L + M, dtype=x.dtype)  # Q2 (:284-288)
    attn_mask[:, :L] = 0
    for i in range(cfg["mem_layers"]):
        y = encoder_layer(p, "encoder.interlingua_layers.%d." % i, torch.cat((h_enc, mem), 0),
                          torch.zeros(B, L + M, dtype=torch.bool), heads, attn_mask=attn_mask)  # Q1: no key padding
        if i == 0:
            inter["mem_layer0"] = y
        mem = y[-M:]
    return mem, inter


def s2t_w2v2_encoder(p: P, wav, src_lengths, cfg):
    """S2T_W2V2_TransformerEncoder.forward (models/chimera/w2v2_transformer.py:338-386):
    audio gets sinusoidal positions here (:356-358)."""
    d, heads = cfg["d"], cfg["heads"]
    x, lens, inter = audio_frontend(p, wav, src_lengths, cfg)
    x = math.sqrt(d) * x
    pm = lengths_to_padding_mask(lens, max_len=x.size(0))
    x = _st(x + positional_embedding(pm, d, 1).transpose(0, 1))
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(p, "encoder.transformer_layers.%d." % i, x, pm, heads)
    x = layer_norm(x, p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])
    return x, (pm if pm.any() else None), inter


def s2t_encoder(p: P, feats, src_lengths, cfg):
    """S2TTransformerEncoder.forward (models/speech_to_text/s2t_transformer.py:313-345): filter-bank input [B,T,F] ->
    Conv1dSubsampler -> x sqrt(d) + sinusoidal positions -> pre-norm layers -> LayerNorm; the padding mask is dropped
    when no frame is padded (:335-336)."""
    d, heads = cfg["d"], cfg["heads"]
    x, lens = conv1d_subsampler(p, "encoder.subsample.", feats, src_lengths)
    x = math.sqrt(d) * x
    pm = lengths_to_padding_mask(lens, max_len=x.size(0))
    x = _st(x + positional_embedding(pm, d, 1).transpose(0, 1))
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(p, "encoder.transformer_layers.%d." % i, x, pm, heads)
    if "encoder.layer_norm.weight" in p:
        x = layer_norm(x, p["encoder.layer_norm.weight"], p["encoder.layer_norm.bias"])
    return x, (pm if pm.any() else None)


# --------------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------------
def decoder_layer(p: P, pre: str, x, enc, enc_pm, heads, self_mask, self_pm):
    """TransformerDecoderLayer.forward, pre-norm (modules/transformer_layer.py:278-412)."""
    res = x
    h = layer_norm(x, p[pre + "self_attn_layer_norm.weight"], p[pre + "self_attn_layer_norm.bias"])
    h = mha(p, pre + "self_attn.", h, h, h, heads, key_padding_mask=self_pm, attn_mask=self_mask)
    x = _st(res + h)
    res = x
    h = layer_norm(x, p[pre + "encoder_attn_layer_norm.weight"], p[pre + "encoder_attn_layer_norm.bias"])
    h = mha(p, pre + "encoder_attn.", h, enc, enc, heads, key_padding_mask=enc_pm)
    x = _st(res + h)
    res = x
    h = layer_norm(x, p[pre + "final_layer_norm.weight"], p[pre + "final_layer_norm.bias"])
    h = _relu(linear(h, p[pre + "fc1.weight"], p[pre + "fc1.bias"]), pre + "fc1")
    h = linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])
    return _st(res + h)


def decoder(p: P, prev_output_tokens, enc, enc_pm, cfg, return_features=False):
    """TransformerDecoder.extract_features_scriptable + output_layer (models/transformer.py:720-836),
    causal mask from buffered_future_mask (:844-856); tied output projection (:636-642)."""
    d, heads = cfg["d"], cfg["dec_heads"]
    emb = p["decoder.embed_tokens.weight"]
    x = math.sqrt(d) * F.embedding(prev_output_tokens, emb, padding_idx=1)  # Embedding(..., padding_idx) (transformer.py:906-911): pad row gets no grad
    x = _st(x + positional_embedding(prev_output_tokens, d, 1))
    x = x.transpose(0, 1)
    self_pm = prev_output_tokens.eq(1) if prev_output_tokens.eq(1).any() else None
    U = x.size(0)
    causal = torch.triu(torch.full((U, U), NEG_INF), 1)
    for i in range(cfg["dec_layers"]):
        x = decoder_layer(p, "decoder.layers.%d." % i, x, enc, enc_pm, heads, causal, self_pm)
    x = layer_norm(x, p["decoder.layer_norm.weight"], p["decoder.layer_norm.bias"])
    feats = x
    x = x.transpose(0, 1)
    out_w = p.get("decoder.output_projection.weight", emb)
    logits = _st(x.matmul(out_w.t()))
    return (logits, feats) if return_features else logits


# --------------------------------------------------------------------------------------------
# criterions
# --------------------------------------------------------------------------------------------
def label_smoothed_nll_loss(logits, target, eps: float, pad: int = 1):
    """get_normalized_probs -> fp32 log_softmax (models/fairseq_decoder.py:58-79; utils.py:469-473)
    + label_smoothed_nll_loss (criterions/label_smoothed_cross_entropy.py:13-30), reduce=True."""
    lprobs = torch.log_softmax(logits.float(), dim=-1).view(-1, logits.size(-1))
    tgt = target.reshape(-1, 1)
    nll = -lprobs.gather(dim=-1, index=tgt)
    smooth = -lprobs.sum(dim=-1, keepdim=True)
    padm = tgt.eq(pad)
    nll = nll.masked_fill(padm, 0.0).sum()
    smooth = smooth.masked_fill(padm, 0.0).sum()
    loss = (1.0 - eps) * nll + (eps / lprobs.size(-1)) * smooth
    return loss, nll


def contrastive_loss(mem_audio, mem_text, temp: float):
    """TripletSTMTContrastiveCriterion.compute_contrastive (criterions/triplet_st_mt_contrastive.py:154-169):
    cosine similarity [B, M(audio), M(text)] / temp; CE with class dim = audio slot, target = arange(M); summed."""
    a = mem_audio.transpose(0, 1).float()
    t = mem_text.transpose(0, 1).float()
    B, M, _ = a.shape
    logits = torch.cosine_similarity(a.unsqueeze(2), t.unsqueeze(1), dim=-1) / temp
    target = torch.arange(M)[None].repeat(B, 1)
    return F.cross_entropy(logits, target, reduction="sum")


def chimera_forward_with_internal(p: P, src_tokens, src_lengths, prev_output_tokens, cfg):
    """S2TTransformerInterlinguaModelW2V2.forward_with_internal (w2v2_transformer_interlingua.py:137-146)."""
    mem, inter = chimera_encoder(p, src_tokens, src_lengths, cfg)
    enc_pm = torch.zeros(mem.size(1), mem.size(0), dtype=torch.bool)  # :301-304
    logits, feats = decoder(p, prev_output_tokens, mem, enc_pm, cfg, return_features=True)
    inter["dec_features_ln"] = feats
    return logits, mem, inter


def triplet_criterion(p: P, sample: dict, cfg, eps=0.1, loss_ratio=(1.0, 1.0, 1.0), temp=0.1):
    """TripletSTMTContrastiveCriterion.forward (criterions/triplet_st_mt_contrastive.py:68-146)."""
    ni = sample["net_input"]
    st_logits, mem_a, inter_a = chimera_forward_with_internal(
        p, ni["src_tokens"], ni["src_lengths"], ni["prev_output_tokens"], cfg)
    st_loss, st_nll = label_smoothed_nll_loss(st_logits, sample["target"], eps)
    mt_logits, mem_t, inter_t = chimera_forward_with_internal(
        p, sample["src_text"], sample["src_text_lengths"], ni["prev_output_tokens"], cfg)
    mt_loss, mt_nll = label_smoothed_nll_loss(mt_logits, sample["target"], eps)
    con = contrastive_loss(mem_a, mem_t, temp)
    loss = loss_ratio[0] * st_loss + loss_ratio[1] * mt_loss + loss_ratio[2] * con
    nll = loss_ratio[0] * st_nll + loss_ratio[1] * mt_nll
    out = dict(loss=loss, nll_loss=nll, st_loss=st_loss, st_nll_loss=st_nll, mt_loss=mt_loss,
               mt_nll_loss=mt_nll, contrastive_loss=con, sample_size=sample["ntokens"],
               st_logits=st_logits, mt_logits=mt_logits, memory_audio=mem_a, memory_text=mem_t,
               inter_audio=inter_a, inter_text=inter_t)
    return out


def lsce_criterion_chimera(p: P, sample: dict, cfg, eps=0.1):
    """label_smoothed_cross_entropy on the Chimera model (criterions/label_smoothed_cross_entropy.py:56-108 over
    S2TTransformerInterlinguaModelW2V2.forward): the MT pre-training update of chimera/scripts/train-en2any-MT.sh:38-60
    (`--task translation --arch s2t_transformer_w2v2_interlingua_base`) when net_input.src_tokens are token ids — the encoder takes
    its text branch (w2v2_transformer_interlingua.py:212-217, 233-236) and wav2vec2 / the subsampler receive no gradient — and the
    plain ST update when they are samples.  The key padding mask is derived from src_lengths as a suffix mask whatever side the
    collater padded on (:232): with the translation task's default --left-pad-source True the reference masks the LAST
    src_len_max - len tokens of a short sentence and attends its leading pad embeddings (zeros + positions)."""
    ni = sample["net_input"]
    logits, mem, inter = chimera_forward_with_internal(p, ni["src_tokens"], ni["src_lengths"], ni["prev_output_tokens"], cfg)
    loss, nll = label_smoothed_nll_loss(logits, sample["target"], eps)
    return dict(loss=loss, nll_loss=nll, logits=logits, memory=mem, sample_size=sample["ntokens"])


def s2t_w2v2_forward(p: P, wav, src_lengths, prev_output_tokens, cfg):
    """S2TTransformerModelW2V2.forward (models/chimera/w2v2_transformer.py:222-236)."""
    enc, enc_pm, inter = s2t_w2v2_encoder(p, wav, src_lengths, cfg)
    return decoder(p, prev_output_tokens, enc, enc_pm, cfg), enc, enc_pm


def lsce_criterion(p: P, sample: dict, cfg, eps=0.1):
    """LabelSmoothedCrossEntropyCriterion.forward (criterions/label_smoothed_cross_entropy.py:56-86)."""
    ni = sample["net_input"]
    logits, enc, enc_pm = s2t_w2v2_forward(p, ni["src_tokens"], ni["src_lengths"], ni["prev_output_tokens"], cfg)
    loss, nll = label_smoothed_nll_loss(logits, sample["target"], eps)
    return dict(loss=loss, nll_loss=nll, logits=logits, encoder_out=enc, encoder_padding_mask=enc_pm,
                sample_size=sample["ntokens"])


def s2t_forward(p: P, feats, src_lengths, prev_output_tokens, cfg):
    """S2TTransformerModel.forward (models/speech_to_text/s2t_transformer.py:252-262)."""
    enc, enc_pm = s2t_encoder(p, feats, src_lengths, cfg)
    return decoder(p, prev_output_tokens, enc, enc_pm, cfg), enc, enc_pm


def lsce_criterion_s2t(p: P, sample: dict, cfg, eps=0.1):
    """label_smoothed_cross_entropy on the stock s2t_transformer (BASELINE config 1 / 5 model family): the criterion's
    compute_loss (label_smoothed_cross_entropy.py:87-108) over model(src_tokens, src_lengths, prev_output_tokens)."""
    ni = sample["net_input"]
    logits, enc, enc_pm = s2t_forward(p, ni["src_tokens"], ni["src_lengths"], ni["prev_output_tokens"], cfg)
    loss, nll = label_smoothed_nll_loss(logits, sample["target"], eps)
    return dict(loss=loss, nll_loss=nll, logits=logits, encoder_out=enc, encoder_padding_mask=enc_pm,
                sample_size=sample["ntokens"])


# --------------------------------------------------------------------------------------------
# optimizer path
# --------------------------------------------------------------------------------------------
def clip_grad_norm_(grads: List[torch.Tensor], max_norm: float) -> torch.Tensor:
    """utils.py:323-364 — total L2 norm; grads *= max_norm / (norm + 1e-6) clamped to <= 1."""
    total = torch.norm(torch.stack([torch.norm(g, p=2, dtype=torch.float32) for g in grads]))
    if max_norm > 0:
        coef = (max_norm / (total + 1e-6)).clamp_(max=1)
        for g in grads:
            g.mul_(coef)
    return total


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, beta1=0.9, beta2=0.98, eps=1e-8,
              weight_decay=0.0):
    """optim/adam.py:146-226 — fairseq's in-tree Adam (bias correction folded into step size,
    denom = sqrt(v)+eps un-corrected, decoupled weight decay :216-219).  In-place; step is 1-based."""
    exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    denom = exp_avg_sq.sqrt().add_(eps)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr * math.sqrt(bc2) / bc1
    if weight_decay != 0:
        param.add_(param, alpha=-weight_decay * lr)
    param.addcdiv_(exp_avg, denom, value=-step_size)


def inverse_sqrt_lr(num_updates: int, lr: float, warmup_updates: int, warmup_init_lr: float) -> float:
    """optim/lr_scheduler/inverse_square_root_schedule.py:52-94."""
    if warmup_init_lr < 0:
        warmup_init_lr = 0 if warmup_updates > 0 else lr
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * (lr - warmup_init_lr) / warmup_updates
    return lr * warmup_updates ** 0.5 * num_updates ** -0.5


# --------------------------------------------------------------------------------------------
# decode: incremental decoder step + beam search, restated
# --------------------------------------------------------------------------------------------
def decoder_step_logprobs(p: P, tokens, enc, enc_pm, cfg):
    """What EnsembleModel.forward_decoder yields for one model (sequence_generator.py:806-868):
    log_softmax (fp32) of the last position's logits given the whole prefix.  The reference's
    incremental-state path computes exactly this (incremental vs full max-abs-diff ~1e-6);
    the oracle recomputes the prefix in full."""
    logits = decoder(p, tokens, enc, enc_pm, cfg)
    return torch.log_softmax(logits[:, -1, :].float(), dim=-1)


def beam_search(p: P, enc, enc_pm, cfg, beam: int, max_len: int, min_len: int = 1,
                pad=1, eos=2, unk=3, bos=2, len_penalty=1.0, unk_penalty=0.0, normalize_scores=True):
    """SequenceGenerator._generate + BeamSearch.step (sequence_generator.py:179-541; search.py:109-144),
    restated per sentence (no batch shrinking).  enc [Tk,B,C].  Returns per sentence a list of
    finalized hypotheses sorted by score: dict(tokens, score, positional_scores)."""
    def lp(b, tokens):
        e = enc[:, b:b + 1].repeat(1, beam, 1)
        epm = enc_pm[b:b + 1].repeat(beam, 1) if enc_pm is not None else None
        return decoder_step_logprobs(p, tokens, e, epm, cfg)

    return beam_search_with(lp, enc.size(1), beam, max_len, min_len, pad, eos, unk, bos, len_penalty, unk_penalty, normalize_scores)


def beam_search_with(logprob_fn, B: int, beam: int, max_len: int, min_len: int = 1, pad=1, eos=2, unk=3, bos=2,
                     len_penalty=1.0, unk_penalty=0.0, normalize_scores=True):
    """The search loop itself, over any `logprob_fn(sentence, tokens[beam, step+1]) -> lprobs[beam, V]` (fp32
    log-probabilities of the next token).  Lets the search bookkeeping be checked apart from the decoder."""
    results = []
    for b in range(B):
        tokens = torch.full((beam, max_len + 2), pad, dtype=torch.long)
        tokens[:, 0] = bos  # eos is the bos of generation (sequence_generator.py:248-249)
        scores = torch.zeros(beam, max_len + 1)
        finalized = []
        cand_size = 2 * beam
        for step in range(max_len + 1):
            lprobs = logprob_fn(b, tokens[:, :step + 1]).clone()
            lprobs[lprobs != lprobs] = -math.inf
            lprobs[:, pad] = -math.inf
            lprobs[:, unk] -= unk_penalty
            if step >= max_len:
                lprobs[:, :eos] = -math.inf
                lprobs[:, eos + 1:] = -math.inf
            if step < min_len:
                lprobs[:, eos] = -math.inf
            V = lprobs.size(-1)
            if step == 0:
                cand = lprobs[0:1].contiguous()  # only first beam (search.py:121-124)
            else:
                cand = lprobs + scores[:, step - 1].unsqueeze(-1)
            top_s, top_i = torch.topk(cand.view(-1), k=min(cand_size, cand.numel() - 1))
            beams = top_i // V
            idx = top_i.fmod(V)
            eos_mask = idx.eq(eos) & top_s.ne(-math.inf)
            # finalize eos candidates within the top `beam` (sequence_generator.py:385-416)
            for j in range(min(beam, top_s.numel())):
                if eos_mask[j] and len(finalized) < beam:
                    bi = int(beams[j])
                    toks = torch.cat([tokens[bi, 1:step + 1], torch.tensor([eos])])
                    pos = torch.cat([scores[bi, :step], top_s[j:j + 1]])
                    pos[1:] = pos[1:] - pos[:-1].clone()
                    sc = float(top_s[j])
                    if normalize_scores:
                        sc = sc / ((step + 1) ** len_penalty)
                    finalized.append(dict(tokens=toks, score=sc, positional_scores=pos))
            if len(finalized) >= beam or step >= max_len:
                break
            # pick the first `beam` non-eos candidates as the new active set (:471-499)
            keep = [j for j in range(top_s.numel()) if not eos_mask[j]][:beam]
            kb = beams[keep]
            new_tokens = tokens[kb].clone()
            new_tokens[:, step + 1] = idx[keep]
            new_scores = scores[kb].clone()
            new_scores[:, step] = top_s[keep]
            tokens, scores = new_tokens, new_scores
        finalized.sort(key=lambda h: -h["score"])
        results.append(finalized)
    return results
