"""Device-resident beam search over the incremental decoder — the MI355X-native form of the decode loop of
fairseq/sequence_generator.py:_generate (:179-541).

The reference runs, per generated token, ~70 module calls, a K/V-cache `index_select` per layer
(multihead_attention.py:419-437) and several host synchronisations (`.any()`, `masked_select`, Python lists of
finalized hypotheses).  At batch x beam = 160 rows the arithmetic of a step is a few microseconds per kernel, so the
loop is launch- and sync-bound.  Here one decode step is a FIXED sequence of C-ABI launches whose only step-dependent
input is a device-side counter:

    cst_dec_embed -> per layer [LN, packed QKV GEMM, cst_dec_self_attn (append-only caches + ancestry table), out-proj
    GEMM (+residual), LN, q GEMM, cst_attn_fwd over the per-SENTENCE encoder K/V (beam rows are the query "time" axis, so
    the encoder K/V are neither replicated nor reordered), out-proj GEMM (+residual), LN, fc1 GEMM (+bias+act), fc2 GEMM
    (+bias+residual)] -> LN -> vocabulary GEMM -> cst_beam_step (log-softmax, masks, top-2*beam, finalisation, next rows)

The sequence is captured once per (batch, encoder length) in a HIP graph and replayed; the host reads one int32
(`num_remaining`) every `poll` steps.  Results are the reference's: same candidates, same finalisation order, same
scores (tests/test_decode_engine_gpu.py checks token ids bit-exactly against the reference's SequenceGenerator fixtures
and against the module-by-module mirror path)."""
import ctypes
import os
import math

import torch

from . import kernels as K
from . import lib as L
from .optim import PARAM_EPOCH


class BeamDecodeEngine:
    def __init__(self, decoder, tgt_dict, beam_size, max_len, min_len=1, normalize_scores=True, len_penalty=1.0,
                 unk_penalty=0.0, temperature=1.0, use_graph=True, poll=8, cross_kernel=None, lanes=None):
        self.dec = decoder
        self.pad, self.unk, self.eos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab = len(tgt_dict)
        self.beam, self.max_len, self.min_len = int(beam_size), int(max_len), int(min_len)
        self.normalize_scores, self.len_penalty = bool(normalize_scores), float(len_penalty)
        self.unk_penalty, self.temperature = float(unk_penalty), float(temperature)
        self.use_graph, self.poll = use_graph, max(1, int(poll))
        # cross attention per step: "flash" = cst_attn_fwd with batch = sentence and the beam rows as the query axis (37 us per
        # layer at 32 x beam 5 x 750 source positions, bf16); "flash_hm" = the same kernel over head-major K/V (contiguous per-head
        # streams: no faster, 0.811 vs 0.812 ms per step); "shared" = cst_dec_cross_attn (VALU kernel, one pass with online
        # softmax, K/V rows shared by the beam: ~59 us — the 5 queries' dot products per key cost more than the MFMA tile the
        # flash kernel spends on them; kept selectable, covered by the same tests)
        # "auto" (round 5): "shared" where cst_dec_cross_attn runs its matrix-core kernel (bf16, head dim 64, beam <= 32: the four waves
        # of a (sentence, head) workgroup split the keys, 21.5 us per layer against 26.4 for "flash" on s2t_transformer_l), else "flash"
        if cross_kernel is None:
            cross_kernel = os.environ.get("CST_DEC_CROSS_KERNEL", "auto")
        assert cross_kernel in ("auto", "flash", "flash_hm", "shared")
        self.cross_kernel = cross_kernel
        # lanes: the batch may be cut into groups of sentences, each with its own state, step graph and HIP stream, replayed side by
        # side (sentences never interact in beam search: same hypotheses, tests/test_decode_engine_gpu.py).  Measured on MI355X
        # (32 x beam 5, s2t_transformer_l): 1 lane 0.877 ms per step, 2 lanes 0.910, 3 lanes 1.60, 4 lanes 1.63 — the step graphs of
        # different streams do not overlap on this stack (a half-batch step costs 0.455 ms, two of them 0.91), so the default is 1.
        self.lanes = max(1, int(os.environ.get("CST_DEC_LANES", "1") if lanes is None else lanes))
        self._packed = None
        self._state = {}
        self._cfg = None
        self._streams = []

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def supported(decoder):
        """The fused loop covers the configuration every Chimera / s2t_transformer arch uses: pre-norm layers, sinusoidal
        positions, encoder attention in every layer, no layernorm_embedding / project_in / adaptive softmax."""
        try:
            ok = (decoder.embed_positions is not None and decoder.layernorm_embedding is None and decoder.project_in_dim is None
                  and decoder.project_out_dim is None and decoder.adaptive_softmax is None and len(decoder.layers) > 0)
            for l in decoder.layers:
                ok = ok and l.normalize_before and l.encoder_attn is not None and l.self_attn.head_dim in (32, 64)
                ok = ok and l.self_attn.q_proj.bias is not None and l.encoder_attn.head_dim == l.self_attn.head_dim
            return bool(ok)
        except AttributeError:
            return False

    def invalidate(self):
        """Drop the packed / folded weight copies and the captured graphs (call after changing decoder weights by any other route)."""
        self._packed = None
        self._state.clear()

    def _pack(self, dtype, device):
        """Per-layer packed [3C, C] self-attention projection (weights are constants in eval mode)."""
        # weights may have been updated since the last call (training between validations): re-pack and drop the graphs
        # (autograd versions catch load_state_dict / copy_; optim.PARAM_EPOCH catches the fused optimizer's raw-pointer updates)
        key = (dtype, device, PARAM_EPOCH[0], tuple((p.data_ptr(), p._version) for p in self.dec.parameters()))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        self._state.clear()
        layers = []
        fuse = self._fuse_ln(dtype)
        for l in self.dec.layers:
            sa = l.self_attn
            d = dict(
                wqkv=torch.cat((sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight), 0).detach().contiguous(),
                bqkv=torch.cat((sa.q_proj.bias, sa.k_proj.bias, sa.v_proj.bias), 0).detach().contiguous())
            if fuse:  # LayerNorm folded into the projection that follows it (include/cst.h: cst_dec_ln_linear)
                d["ln_qkv"] = self._fold_ln(l.self_attn_layer_norm, d["wqkv"], d["bqkv"])
                d["ln_q"] = self._fold_ln(l.encoder_attn_layer_norm, l.encoder_attn.q_proj.weight, l.encoder_attn.q_proj.bias)
                d["ln_q_frag"] = self.fragment_major(d["ln_q"][0], l.encoder_attn.num_heads)  # cst_dec_ln_q_cross_attn's weight layout
                d["ln_fc1"] = self._fold_ln(l.final_layer_norm, l.fc1.weight, l.fc1.bias)
            layers.append(d)
        pos = self.dec.embed_positions
        need = self.dec.padding_idx + 2 + self.max_len + 1
        table = pos.get_embedding(need, pos.embedding_dim, pos.padding_idx).to(device=device, dtype=torch.float32).contiguous()
        # the module path adds positions converted to the storage dtype (models/transformer.py:756 on a .half()/bf16 model)
        if dtype != torch.float32:
            table = table.to(dtype).float()
        self._packed = (key, dict(layers=layers, pos=table))
        return self._packed[1]

    def nodes_per_step(self, dtype, rows=None):
        """Kernel launches (graph nodes) of one decode step: embed + per layer (LayerNorm, qkv, self-attention, out, LayerNorm, q,
        cross-attention, out, LayerNorm, fc1, fc2 — the three LayerNorms folded into their projections on the bf16 path) + final
        LayerNorm + vocabulary projection + the two beam-search kernels; + the split-K reduce of fc2 where it is split (bf16, <= 256
        hypothesis rows, ffn >= 4096)."""
        per_layer = 8 if self._fuse_ln(dtype) else 11
        D = self.dec.layers[0].self_attn.head_dim
        if self._fuse_ln(dtype) and self._fuse_q_cross(dtype, D, self.dec.embed_dim, self._cross_mode(dtype, D)):
            per_layer -= 1  # the query projection runs inside the cross-attention launch
        F = self.dec.layers[0].fc1.out_features
        if rows is not None and rows <= 256 and F >= 4096 and dtype == torch.bfloat16 and not os.environ.get("CST_DEC_NO_SPLITK"):
            per_layer += 1
        return 1 + per_layer * len(self.dec.layers) + (1 if self.dec.layer_norm is not None else 0) + 1 + 2

    def _fuse_ln(self, dtype):
        C = self.dec.embed_dim
        return dtype == torch.bfloat16 and C % 512 == 0 and not os.environ.get("CST_DEC_NO_LINEAR") and not os.environ.get("CST_DEC_NO_LN_FUSE")

    @staticmethod
    def _fold_ln(ln, w, b):
        """(Wg, sg, sb) of cst_dec_ln_linear: gamma folded into the weight columns (one bf16 rounding), beta and the bias into sb."""
        wf = w.detach().float()
        wg = (wf * ln.weight.detach().float().unsqueeze(0)).to(w.dtype).contiguous()
        sg = wg.float().sum(dim=1).contiguous()
        sb = (wf * ln.bias.detach().float().unsqueeze(0)).sum(dim=1).contiguous()  # (a row-wise sum, not `wf @ beta`: no vendor BLAS in the package)
        if b is not None:
            sb = (sb + b.detach().float()).contiguous()
        return wg, sg, sb, float(ln.eps)

    @staticmethod
    def fragment_major(wg, H):
        """[H*64, K] -> the fragment-major packing of include/cst.h (cst_dec_ln_q_cross_attn): [H][2][K/16][2][32][8]."""
        N, K = wg.shape
        if N != H * 64 or K % 16:
            return None
        return wg.view(H, 2, 32, K // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()

    def _ln_linear(self, x, folded, out, act=L.ACT_NONE):
        wg, sg, sb, eps = folded
        M, Kd = x.shape
        L.check(L.load().cst_dec_ln_linear(L.ptr(x), L.ptr(wg), L.ptr(sg), L.ptr(sb), eps, None, L.ptr(out), M, wg.shape[0], Kd,
                                           x.stride(0), 0, out.stride(0), act, None, 0, L.dtype_code(x.dtype), L.stream_ptr()),
                "cst_dec_ln_linear")

    # ------------------------------------------------------------------------------------------------------------
    def _alloc(self, lane, bsz, S, dtype, device, has_mask):
        key = (lane, bsz, S, dtype, device, has_mask)
        st = self._state.get(key)
        if st is not None:
            return st
        beam, L1, LT = self.beam, self.max_len + 1, self.max_len + 2
        bbsz, C, nl = bsz * beam, self.dec.embed_dim, len(self.dec.layers)
        F = self.dec.layers[0].fc1.out_features
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=device)
        st = dict(
            step=z(1, dt=torch.int32), num_remaining=z(1, dt=torch.int32),
            tokens=z(2, bbsz, LT, dt=torch.int64), scores=z(2, bbsz, L1, dt=torch.float32), anc=z(2, bbsz, L1, dt=torch.int32),
            ignore=z(bsz, beam, dt=torch.uint8), finished=z(bsz, dt=torch.uint8), nfinal=z(bsz, dt=torch.int32),
            fin_tokens=z(bsz, beam, L1, dt=torch.int64), fin_pos=z(bsz, beam, L1, dt=torch.float32),
            fin_score=z(bsz, beam, dt=torch.float32), fin_len=z(bsz, beam, dt=torch.int32),
            x=z(bbsz, C), x2=z(bbsz, C), h=z(bbsz, C), qkv=z(bbsz, 3 * C), q=z(bbsz, C), attn=z(bbsz, C), f=z(bbsz, F),
            logits=z(bbsz, (self.vocab + 7) // 8 * 8),
            mean=z(bbsz, dt=torch.float32), rstd=z(bbsz, dt=torch.float32), lse=z(bsz * 64 * beam, dt=torch.float32),
            kc=[z(bbsz, L1, C) for _ in range(nl)], vc=[z(bbsz, L1, C) for _ in range(nl)],
            kx=[z(bsz, S, C) for _ in range(nl)], vx=[z(bsz, S, C) for _ in range(nl)],
            proj=z(bsz * S, C), kpm=z(bsz, S, dt=torch.uint8) if has_mask else None, graph=None,
            gemm_ws=z(8 * bbsz * C * 4 if bbsz <= 256 else 0, dt=torch.uint8))  # split-K partials of the fc2 projection (own buffer: captured)
        d = L.BeamDesc()
        d.dtype = L.dtype_code(dtype)
        d.bsz, d.beam, d.vocab, d.max_len = bsz, beam, self.vocab, self.max_len
        d.pad, d.unk, d.eos, d.min_len = self.pad, self.unk, self.eos, self.min_len
        d.unk_penalty, d.len_penalty, d.temperature = self.unk_penalty, self.len_penalty, self.temperature
        d.normalize_scores = int(self.normalize_scores)
        d.logits, d.ld_logits = st["logits"].data_ptr(), st["logits"].stride(0)
        d.step, d.tokens, d.scores, d.anc = (st[k].data_ptr() for k in ("step", "tokens", "scores", "anc"))
        d.cands_to_ignore, d.finished, d.nfinal = st["ignore"].data_ptr(), st["finished"].data_ptr(), st["nfinal"].data_ptr()
        d.num_remaining = st["num_remaining"].data_ptr()
        d.fin_tokens, d.fin_pos, d.fin_score, d.fin_len = (st[k].data_ptr() for k in ("fin_tokens", "fin_pos", "fin_score", "fin_len"))
        st["beam_ws"] = torch.zeros(L.load().cst_beam_workspace(bsz, beam), dtype=torch.uint8, device=device)
        d.workspace = st["beam_ws"].data_ptr()
        st["desc"] = d
        self._state[key] = st
        return st

    # ------------------------------------------------------------------------------------------------------------
    def _linear(self, x, w, b, out, act=L.ACT_NONE, resid=None, ws=None):
        M, Kd = x.shape
        N = w.shape[0]
        if (M <= 1024 and x.dtype == torch.bfloat16 and Kd % 512 == 0 and x.stride(0) % 8 == 0 and w.is_contiguous()
                and os.environ.get("CST_DEC_LINEAR_ALL")):
            # (A/B switch.  Without a LayerNorm to fold in, the decode-step kernel is no faster than the general GEMM's 64 x 64
            #  configuration: both sit at the ~8 us a dependent graph node costs on this stack — tools/bench_dec_linear.py)
            L.check(L.load().cst_dec_linear(L.ptr(x), L.ptr(w), L.ptr(b) if b is not None else None,
                                            L.ptr(resid) if resid is not None else None, L.ptr(out), M, N, Kd, x.stride(0),
                                            0 if resid is None else resid.stride(0), out.stride(0), act, None, 0,
                                            L.dtype_code(x.dtype), L.stream_ptr()), "cst_dec_linear")
            return
        # fc2 (K = 4096) at <= 256 rows: 48 workgroups would stream 170 KB of weights each through a 48 KB ring — 23.7 us; eight K
        # slices + the reduce launch: 15.3 us although it is a node more (tools/bench_dec_splitk.py; K = 1024 projections lose with any
        # split).  bf16 only: the fp32 parity configuration keeps the summation order of the module path it is compared with.
        split = 8 if (ws is not None and M <= 256 and Kd >= 4096 and x.dtype == torch.bfloat16 and not os.environ.get("CST_DEC_NO_SPLITK")) else 1
        K.gemm(x, w, out, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=out.stride(0), bias=b, act=act,
               resid=resid, ld_resid=0 if resid is None else resid.stride(0), split_k=split, ws=ws if split > 1 else None)

    def _cross_mode(self, dtype, D):
        if self.cross_kernel != "auto":
            return self.cross_kernel
        return "shared" if (dtype == torch.bfloat16 and D == 64 and self.beam <= 32) else "flash"

    def _fuse_q_cross(self, dtype, D, C, ck):
        """The LayerNorm-folded query projection inside the cross-attention launch (cst_dec_ln_q_cross_attn): one node less per layer."""
        return (ck == "shared" and dtype == torch.bfloat16 and D == 64 and self.beam <= 32 and C % 256 == 0
                and not os.environ.get("CST_DEC_NO_QCROSS"))

    def _cross_fits(self, dtype, D):
        """True: the encoder K/V of this engine are stored head-major [bsz, H, S, D]."""
        return self._cross_mode(dtype, D) in ("shared", "flash_hm")

    def _ln(self, x, ln, out, st):
        lib = L.load()
        L.check(lib.cst_layernorm_fwd(L.ptr(x), None, L.ptr(ln.weight), L.ptr(ln.bias), L.ptr(out), None, L.ptr(st["mean"]),
                                      L.ptr(st["rstd"]), x.shape[0], x.shape[1], ln.eps, L.dtype_code(x.dtype), L.stream_ptr()),
                "cst_layernorm_fwd")

    def _step(self, st, pk, bsz):
        """One decode step: every launch reads the step counter from device memory."""
        lib, dec = L.load(), self.dec
        bbsz, C = st["x"].shape
        dt = L.dtype_code(st["x"].dtype)
        H = dec.layers[0].self_attn.num_heads
        D = C // H
        L.check(lib.cst_dec_embed(L.ptr(st["tokens"]), L.ptr(st["step"]), L.ptr(dec.embed_tokens.weight), L.ptr(pk["pos"]),
                                  float(dec.embed_scale), dec.padding_idx, L.ptr(st["x"]), bbsz, C, self.max_len,
                                  pk["pos"].shape[0], dt, L.stream_ptr()), "cst_dec_embed")
        x, x2 = st["x"], st["x2"]
        for li, layer in enumerate(dec.layers):
            sa, ca, p = layer.self_attn, layer.encoder_attn, pk["layers"][li]
            fused = "ln_qkv" in p and bbsz <= 1024
            if fused:
                self._ln_linear(x, p["ln_qkv"], st["qkv"])
            else:
                self._ln(x, layer.self_attn_layer_norm, st["h"], st)
                self._linear(st["h"], p["wqkv"], p["bqkv"], st["qkv"])
            L.check(lib.cst_dec_self_attn(L.ptr(st["qkv"]), L.ptr(st["kc"][li]), L.ptr(st["vc"][li]), L.ptr(st["anc"]),
                                          L.ptr(st["step"]), L.ptr(st["attn"]), bbsz, H, D, self.max_len, float(sa.scaling), dt,
                                          L.stream_ptr()), "cst_dec_self_attn")
            self._linear(st["attn"], sa.out_proj.weight, sa.out_proj.bias, x2, resid=x)
            x, x2 = x2, x
            S = st["kx"][li].shape[1]
            ck = self._cross_mode(st["x"].dtype, D)
            qfused = fused and p.get("ln_q_frag") is not None and self._fuse_q_cross(st["x"].dtype, D, C, ck)
            if qfused:
                pass  # the query projection runs inside the cross-attention launch below
            elif fused:
                self._ln_linear(x, p["ln_q"], st["q"])
            else:
                self._ln(x, layer.encoder_attn_layer_norm, st["h"], st)
                self._linear(st["h"], ca.q_proj.weight, ca.q_proj.bias, st["q"])
            # cross attention: one workgroup per (sentence, head); the sentence's K/V rows serve all of its beam rows
            if qfused:
                _, sg, sb, eps = p["ln_q"]
                L.check(lib.cst_dec_ln_q_cross_attn(L.ptr(x), x.stride(0), L.ptr(p["ln_q_frag"]), L.ptr(sg), L.ptr(sb), eps, L.ptr(st["kx"][li]),
                                                    L.ptr(st["vx"][li]), L.ptr(st["kpm"]), L.ptr(st["attn"]), L.ptr(st["step"]), self.max_len,
                                                    bsz, self.beam, H, D, S, float(ca.scaling), dt, L.stream_ptr()), "cst_dec_ln_q_cross_attn")
            elif ck == "flash_hm":  # the flash kernel over head-major K/V: (b, h, t) strides = (H*S*D, S*D, D)
                q3, o3 = st["q"].view(bsz, self.beam, C), st["attn"].view(bsz, self.beam, C)
                d = K.attn_desc(q3, st["kx"][li], st["vx"][li], o3, st["lse"], H, D, st["kpm"], False, float(ca.scaling))
                d.k_sb = d.v_sb = H * S * D
                d.k_sh = d.v_sh = S * D
                d.k_st = d.v_st = D
                K.attn_fwd_desc(d)
            elif ck == "shared":
                L.check(lib.cst_dec_cross_attn(L.ptr(st["q"]), L.ptr(st["kx"][li]), L.ptr(st["vx"][li]), L.ptr(st["kpm"]), L.ptr(st["attn"]),
                                               L.ptr(st["step"]), self.max_len, bsz, self.beam, H, D, S, float(ca.scaling), dt,
                                               L.stream_ptr()), "cst_dec_cross_attn")
            else:  # very long sources: the flash kernel with batch = sentence, query "time" axis = the beam rows
                q3, o3 = st["q"].view(bsz, self.beam, C), st["attn"].view(bsz, self.beam, C)
                d = K.attn_desc(q3, st["kx"][li], st["vx"][li], o3, st["lse"], H, D, st["kpm"], False, float(ca.scaling))
                K.attn_fwd_desc(d)
            self._linear(st["attn"], ca.out_proj.weight, ca.out_proj.bias, x2, resid=x)
            x, x2 = x2, x
            act = L.ACT_GELU if layer.activation_fn == "gelu" else L.ACT_RELU
            if fused:
                self._ln_linear(x, p["ln_fc1"], st["f"], act=act)
            else:
                self._ln(x, layer.final_layer_norm, st["h"], st)
                self._linear(st["h"], layer.fc1.weight, layer.fc1.bias, st["f"], act=act)
            self._linear(st["f"], layer.fc2.weight, layer.fc2.bias, x2, resid=x, ws=st["gemm_ws"])
            x, x2 = x2, x
        if dec.layer_norm is not None:
            self._ln(x, dec.layer_norm, st["h"], st)
            feat = st["h"]
        else:
            feat = x
        w = dec.output_projection.weight
        self._linear(feat, w, None, st["logits"])
        L.check(lib.cst_beam_step(ctypes.byref(st["desc"]), L.stream_ptr()), "cst_beam_step")
        # an even number of x/x2 swaps per layer (3) x layers may leave the residual stream in x2: the NEXT step's embed always
        # writes st["x"], and every step performs the same swaps, so the captured sequence is step-invariant.

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, encoder_out, bsz):
        """encoder_out: EncoderOut with encoder_out [S, B, C] (T x B x C view) and encoder_padding_mask [B, S] or None.
        Returns the reference's `finalized` structure (list over sentences of hypothesis dicts, best first)."""
        enc = encoder_out.encoder_out
        S, B, Ce = enc.shape
        assert B == bsz
        dtype, device = enc.dtype, enc.device
        mask = encoder_out.encoder_padding_mask
        has_mask = mask is not None and mask.dim() == 2
        pk = self._pack(dtype, device)
        lanes = min(self.lanes, bsz)
        bounds = [(bsz * i // lanes, bsz * (i + 1) // lanes) for i in range(lanes)]
        cfg = (tuple(b1 - b0 for b0, b1 in bounds), S, dtype, device, has_mask)
        if cfg != self._cfg:
            self._state.clear()  # one resident configuration (the caches are the large buffers)
            self._cfg = cfg
        main = torch.cuda.current_stream()
        while lanes > 1 and len(self._streams) < lanes:
            self._streams.append(torch.cuda.Stream(device=device))
        encb = enc.transpose(0, 1)
        encb = encb if encb.is_contiguous() else encb.contiguous()
        total = self.max_len + 1
        runs = []
        for i, (b0, b1) in enumerate(bounds):
            stream = main if lanes == 1 else self._streams[i]
            if stream is not main:
                stream.wait_stream(main)
            with torch.cuda.stream(stream):
                st = self._alloc(i, b1 - b0, S, dtype, device, has_mask)
                done = self._begin(st, pk, encb[b0:b1], mask[b0:b1] if has_mask else None, b1 - b0)
            runs.append(dict(stream=stream, st=st, bsz=b1 - b0, steps=done, remaining=b1 - b0))
        while any(r["steps"] < total and r["remaining"] > 0 for r in runs):
            for r in runs:
                r["n"] = min(self.poll, total - r["steps"]) if r["remaining"] > 0 else 0
            for j in range(self.poll):  # the lanes' steps alternate in launch order; each lane's own order is its stream's
                for r in runs:
                    if j < r["n"]:
                        with torch.cuda.stream(r["stream"]):
                            if self.use_graph:
                                r["st"]["graph"].replay()
                            else:
                                self._step(r["st"], pk, r["bsz"])
            for r in runs:
                if r["n"]:
                    r["steps"] += r["n"]
                    with torch.cuda.stream(r["stream"]):
                        r["remaining"] = int(r["st"]["num_remaining"].item())  # the only host sync of the loop (one per lane)
        assert all(r["remaining"] == 0 for r in runs), "beam search did not terminate within max_len + 1 steps"
        finalized = []
        for r in runs:
            with torch.cuda.stream(r["stream"]):
                finalized += self._collect(r["st"], r["bsz"], None if r["stream"] is main else main)
            if r["stream"] is not main:
                main.wait_stream(r["stream"])
        return finalized

    def _begin(self, st, pk, encb, mask, bsz):
        """Queues the per-call work of one lane on the current stream: the static cross-attention K/V of its sentences, the beam
        state, and — first call of a configuration — the eager step 0 and the capture of the step graph.  Returns the number of
        decode steps already taken (1 after that eager step, else 0)."""
        S, Ce = encb.shape[1], encb.shape[2]
        flat = encb.reshape(bsz * S, Ce)
        for li, layer in enumerate(self.dec.layers):  # static cross-attention K/V, once per sentence (not per beam)
            ca = layer.encoder_attn
            if self._cross_fits(st["x"].dtype, ca.head_dim):  # head-major [bsz, H, S, D] for cst_dec_cross_attn (one transposing copy per call)
                for name, proj in (("kx", ca.k_proj), ("vx", ca.v_proj)):
                    self._linear(flat, proj.weight, proj.bias, st["proj"])
                    st[name][li].view(bsz, ca.num_heads, S, ca.head_dim).copy_(
                        st["proj"].view(bsz, S, ca.num_heads, ca.head_dim).transpose(1, 2))
            else:
                self._linear(flat, ca.k_proj.weight, ca.k_proj.bias, st["kx"][li].view(bsz * S, -1))
                self._linear(flat, ca.v_proj.weight, ca.v_proj.bias, st["vx"][li].view(bsz * S, -1))
        if mask is not None:
            st["kpm"].copy_(mask.to(torch.uint8))
        L.check(L.load().cst_beam_init(ctypes.byref(st["desc"]), L.stream_ptr()), "cst_beam_init")
        if self.use_graph and st["graph"] is None:
            self._step(st, pk, bsz)  # eager warm-up step 0 (loads code objects, sizes the GEMM workspace)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step(st, pk, bsz)
            st["graph"] = g
            return 1
        return 0

    def _collect(self, st, bsz, consumer=None):
        # hypotheses are views of ONE device-side copy of the result buffers (the engine state is reused by the next call)
        d_tokens, d_pos, d_score = st["fin_tokens"].clone(), st["fin_pos"].clone(), st["fin_score"].clone()
        if consumer is not None:  # made on a lane's stream, read by the caller on its own
            for t in (d_tokens, d_pos, d_score):
                t.record_stream(consumer)
        fin_score, fin_len, nfinal = d_score.cpu(), st["fin_len"].cpu(), st["nfinal"].cpu()
        finalized = []
        for b in range(bsz):
            hyps = []
            for r in range(int(nfinal[b])):
                n = int(fin_len[b, r])
                hyps.append({"tokens": d_tokens[b, r, :n], "score": d_score[b, r], "attention": None, "alignment": None,
                             "positional_scores": d_pos[b, r, :n]})
            _, order = torch.sort(fin_score[b, :len(hyps)], descending=True)  # sequence_generator.py:529-540
            finalized.append([hyps[i] for i in order.tolist()])
        return finalized
