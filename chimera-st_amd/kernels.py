"""Raw (non-autograd) wrappers: torch tensors in, C-ABI calls out.  Torch only supplies device
memory and the current HIP stream here; every arithmetic op is a kernel of libcst_hip.so."""
import ctypes
import os

import torch

from . import lib as L

_ws = {}
STATS = {}  # light counters for tests / tools (e.g. how many GEMM launches carried live-tile stamps)


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer, stream-ordered reuse (all product kernels run on the current stream)."""
    key = (device.index, L.stream_ptr().value)  # (the raw-stream getter: torch.cuda.current_stream() costs ~20 us of Python per call)
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


class _Deferred:
    """Second stages of fixed-order reductions (split-K slabs of the small weight-gradient GEMMs, bias-gradient slices, LayerNorm
    dgamma / dbeta partials) that are finished together by ONE cst_reduce_multi launch instead of a launch each.  The results
    are gradients of parameters: nothing reads them before the update gathers them, PROVIDED no second gradient of the same
    parameter arrives in the same backward pass (autograd would add the two at once) — so the mode is switched on only by a trainer
    whose criterion runs one pass over a model that uses each parameter once (trainer.py), parameters shared between modules are
    excluded at the call sites, and a gradient that is being accumulated into (p.grad is not None) takes the immediate route.
    CST_DEFER_POISON=1 (the test suite sets it) fills every deferred destination with NaN until the flush writes it: a read that
    comes too early cannot go unnoticed."""

    MAX_BYTES = 32 << 20  # slabs above this stay with their own reduce launch (they would have to be re-read cold from HBM)

    def __init__(self):
        self.on, self.items, self.keep, self.flushes = 0, [], [], 0
        self.main = None   # the stream the flush runs on (set when the mode is entered)
        self.side = None   # side stream of the small weight-gradient GEMMs (side_gemm), created on first use
        self.side_used = False
        self.poison = False

    def push(self, src_ptr, dst, stride, Lr, P, keep, order=0):
        if self.poison:
            dst.fill_(float("nan"))
        # (side-stream launches only — torch.cuda.current_stream() costs ~20 us of Python, 320 pushes per update: never in the default path)
        if self.side_used and torch.is_tensor(keep) and torch.cuda.current_stream() != self.main:
            keep.record_stream(self.main)  # allocated on the side stream's pool, read by the flush on the main stream
        self.items.append((src_ptr, dst.data_ptr(), stride, Lr, P, L.dtype_code(dst.dtype), order))
        # (an ALIAS of dst keeps its storage alive: holding dst itself would raise its reference count, and AccumulateGrad only adopts
        #  a gradient tensor nobody else holds — otherwise it clones it on the spot, i.e. reads it before the flush has written it)
        self.keep.append((keep, dst.detach()))

    def join(self):
        """The main stream waits for everything the side stream has been given (before anything reads a gradient)."""
        if self.side_used:
            torch.cuda.current_stream().wait_stream(self.side)
            self.side_used = False

    def flush(self):
        self.join()
        if self.items:
            reduce_multi(self.items)
            self.flushes += 1
            self.items, self.keep = [], []


DEFER = _Deferred()


def reduce_multi(items):
    """items: [(src_ptr, dst_ptr, stride, L, P, dst_dtype_code, order)] -> dst[i] = sum_p src[p * stride + i]: ONE launch (include/cst.h:
    order 0 = the split-K reduce's summation order, 1 = the LayerNorm second stage's)."""
    lib = L.load()
    for at in range(0, len(items), 64):
        chunk = items[at:at + 64]
        arr = (L.ReduceItem * len(chunk))()
        for a, (src, dst, stride, Lr, P, dt, order) in zip(arr, chunk):
            a.src, a.dst, a.stride, a.L, a.P, a.dst_dtype, a.order = src, dst, stride, Lr, P, dt, order
        L.check(lib.cst_reduce_multi(arr, len(chunk), L.stream_ptr()), "cst_reduce_multi")


class deferred_reductions:
    """with deferred_reductions(enabled): loss.backward()  — the flush runs on exit (and before any bucket all-reduce: distributed.py)."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and not os.environ.get("CST_NO_DEFER")

    def __enter__(self):
        if self.enabled:
            DEFER.on += 1
            DEFER.main = torch.cuda.current_stream()
            DEFER.poison = bool(os.environ.get("CST_DEFER_POISON"))
        return self

    def __exit__(self, *exc):
        if self.enabled:
            DEFER.on -= 1
            if DEFER.on == 0:
                DEFER.flush()
        return False


def side_gemm(big, reads, *args, **kw):
    """A weight-gradient GEMM of a SMALL layer (`big` False: the 4-wave configurations, a few hundred workgroups that leave most
    of every CU free) on a side stream, next to the dX chain of the backward pass it belongs to: the backward of the 512-wide
    encoder / decoder layers is a chain of 10-50 us kernels none of which fills the chip, and the weight gradients are off its
    critical path — nothing reads them before the update gathers them.  Only inside the trainer's backward context (DEFER.on: the
    same guarantee — the result is a parameter gradient nobody reads yet), joined before the deferred reductions are flushed.
    `reads`: the tensors the launch reads; they are handed to the side stream's allocator bookkeeping (record_stream) because
    autograd frees them as soon as the backward function returns."""
    # (measured on MI355X, same box, two runs each: 64.7 / 65.2 ms per update with the side stream, 64.9 / 65.0 without — the dX
    #  chain's own launches already take every CU's LDS, the side launches only slip into their tails.  Kept as a switch, off.)
    if big or not DEFER.on or DEFER.main is None or not os.environ.get("CST_SIDE_STREAM") or torch.cuda.current_stream() != DEFER.main:
        return gemm(*args, **kw)
    if DEFER.side is None:
        DEFER.side = torch.cuda.Stream()
    side = DEFER.side
    side.wait_stream(DEFER.main)  # the operands are the latest things queued on the main stream
    DEFER.side_used = True        # (before the launch: the pushes inside gemm() look at it)
    with torch.cuda.stream(side):
        out = gemm(*args, **kw)
    for t in reads:
        if t is not None:
            t.record_stream(side)
    STATS["side_gemm"] = STATS.get("side_gemm", 0) + 1
    return out


def _2d(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor, got %s / %s" % (tuple(t.shape), t.stride())
    return t


# ------------------------------------------------------------------------------------------------
def gemm(A, B, C, M, N, K, *, a_kmajor, b_kmajor, lda, ldb, ldc, bias=None, bias_mode=L.BIAS_COL, act=L.ACT_NONE,
         aux_out=None, ld_aux_out=0, dact=L.ACT_NONE, aux_in=None, ld_aux_in=0, resid=None, ld_resid=0, alpha=1.0,
         batch0=1, batch1=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), sbias=(0, 0), a_seg=0, a_seg_stride=0, b_seg=0,
         b_seg_stride=0, split_k=-1, a_off=0, b_off=0, c_off=0, drop_p=0.0, drop_key=0, k_live=None, m_live=None, k_len=None, m_len=None,
         ws=None, colsum=None, defer=False, reduce_to=None):
    """C = epilogue(alpha * Aop @ Bop); see include/cst.h.  colsum (mn-major A): [batch, M] tensor of A's dtype that receives the
    column sums of A over k — the bias gradient next to a weight-gradient GEMM (cst_gemm_desc.colsum).
    defer: the caller allows the split-K reduce of this launch (C and colsum are parameter gradients nobody reads yet) to be left to
    the deferred-reduction flush (DEFER); ignored when the mode is off, the launch does not split, or the slabs are large.
    reduce_to (batched launches into an fp32 C [batch, M, N] with sc = (M * N, 0)): the sum over the batch — and over the K slices
    — goes to `reduce_to` [M, N] through cst_reduce_multi instead of a torch sum over C (which is then never written when K is split).  Offsets *_off are in elements.  k_live = (stamps, epoch): the 64-wide
    K blocks of A that are not all-zero (LiveTiles.pair()); dead blocks may be skipped; m_live: the same stamps for the rows of A of a
    row-wise GEMM (dX): output tiles without a live row skip their K loop.  ws: the caller's own split-K scratch (uint8; launches
    captured into a graph must not depend on the shared grow-only buffer)."""
    lib = L.load()
    d = L.GemmDesc()
    d.dtype = L.dtype_code(A.dtype)
    assert B.dtype == A.dtype, "A/B dtype mismatch"
    d.c_dtype = L.dtype_code(C.dtype)
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.M, d.N, d.K = M, N, K
    es = A.element_size()
    d.A = A.data_ptr() + a_off * es
    d.lda, d.a_seg, d.a_seg_stride = lda, a_seg, a_seg_stride
    d.B = B.data_ptr() + b_off * es
    d.ldb, d.b_seg, d.b_seg_stride = ldb, b_seg, b_seg_stride
    d.C = C.data_ptr() + c_off * C.element_size()
    d.ldc = ldc
    if bias is not None:
        assert bias.dtype == A.dtype
        d.bias, d.bias_mode = bias.data_ptr(), bias_mode
    else:
        d.bias, d.bias_mode = None, L.BIAS_NONE
    d.sbias0, d.sbias1 = sbias
    d.act = act
    d.aux_out = None if aux_out is None else aux_out.data_ptr() + c_off * es
    d.ld_aux_out = ld_aux_out
    d.dact = dact
    d.aux_in = None if aux_in is None else aux_in.data_ptr() + c_off * es
    d.ld_aux_in = ld_aux_in
    d.resid = None if resid is None else resid.data_ptr() + c_off * es
    d.ld_resid = ld_resid
    d.alpha = alpha
    d.drop_p, d.drop_key = float(drop_p), int(drop_key) & 0xFFFFFFFF
    d.batch0, d.batch1 = batch0, batch1
    d.sa0, d.sa1 = sa
    d.sb0, d.sb1 = sb
    d.sc0, d.sc1 = sc
    d.split_k = split_k
    if k_live is not None:
        assert k_live[0].numel() * 64 >= K and k_live[0].dtype == torch.int32
        STATS["gemm_k_live"] = STATS.get("gemm_k_live", 0) + 1
        d.k_live, d.k_epoch = k_live[0].data_ptr(), k_live[1]
    else:
        d.k_live, d.k_epoch = None, 0
    if m_live is not None:
        assert m_live[0].numel() * 64 >= M and m_live[0].dtype == torch.int32
        d.m_live, d.m_epoch = m_live[0].data_ptr(), m_live[1]
        STATS["gemm_m_live"] = STATS.get("gemm_m_live", 0) + 1
    else:
        d.m_live, d.m_epoch = None, 0
    if k_len is not None:
        assert k_len.dtype == torch.int32 and k_len.numel() == batch0 and k_len.is_contiguous()
        d.k_len = k_len.data_ptr()
        STATS["gemm_k_len"] = STATS.get("gemm_k_len", 0) + 1
    else:
        d.k_len = None
    if m_len is not None:
        assert m_len.dtype == torch.int32 and m_len.numel() == batch0 and m_len.is_contiguous()
        d.m_len = m_len.data_ptr()
        STATS["gemm_m_len"] = STATS.get("gemm_m_len", 0) + 1
    else:
        d.m_len = None
    if colsum is not None:
        assert not a_kmajor and colsum.dtype == A.dtype and colsum.is_contiguous() and colsum.numel() == M * batch0 * batch1
        d.colsum = colsum.data_ptr()
    else:
        d.colsum = None
    d.defer_reduce = 0
    splits = 1
    need = lib.cst_gemm_workspace(ctypes.byref(d))
    nb = batch0 * batch1
    deferred = False
    if need > 0 and (reduce_to is not None or (defer and DEFER.on and nb == 1 and need <= DEFER.MAX_BYTES and ldc == N and N % 8 == 0)):
        splits = lib.cst_gemm_splits(ctypes.byref(d))
        if splits > 1:
            assert ws is None and bias is None and act == L.ACT_NONE and dact == L.ACT_NONE and resid is None and aux_out is None and alpha == 1.0
            ws = torch.empty(need, dtype=torch.uint8, device=A.device)  # private: it must outlive this call (slabs read by the flush)
            d.defer_reduce = 1
            deferred = True
    if need > 0:
        if ws is None:
            ws = workspace(need, A.device)
        assert ws.dtype == torch.uint8 and ws.numel() >= need, "split-K scratch too small: %d < %d" % (ws.numel(), need)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    else:
        d.workspace, d.workspace_bytes = None, 0
    L.check(lib.cst_gemm(ctypes.byref(d), L.stream_ptr()), "cst_gemm")
    if reduce_to is not None:
        assert C.dtype == torch.float32 and sc == (M * N, 0) and ldc == N and reduce_to.numel() == M * N and reduce_to.is_contiguous() and colsum is None
        item = ((ws.data_ptr(), reduce_to, M * N, M * N, nb * splits, ws) if deferred else (C.data_ptr(), reduce_to, M * N, M * N, nb, C))
        if defer and DEFER.on:
            DEFER.push(*item)
        else:
            reduce_multi([(item[0], item[1].data_ptr(), item[2], item[3], item[4], L.dtype_code(reduce_to.dtype), 0)])
    elif deferred:
        DEFER.push(ws.data_ptr(), C, M * N, M * N, splits, ws)
        if colsum is not None:
            DEFER.push(ws.data_ptr() + 4 * splits * M * N, colsum, M, M, splits, ws)
    return C


_DW_FUSED = {}


def dw_colsum_is_fused(n_out, k_in, tokens, dtype):
    """Whether the weight-gradient GEMM dW[n_out, k_in] = dY^T X over `tokens` rows yields the bias gradient as a by-product of its
    own kernel (cst_gemm_colsum_is_fused) — the 4-wave configurations do, the 16-wave one runs a separate column sum."""
    key = (n_out, k_in, tokens, dtype)
    r = _DW_FUSED.get(key)
    if r is None:
        d = L.GemmDesc()
        d.dtype = d.c_dtype = L.dtype_code(dtype)
        d.a_kmajor = d.b_kmajor = 0
        d.M, d.N, d.K = n_out, k_in, tokens
        d.batch0 = d.batch1 = 1
        d.split_k = -1
        d.colsum = 1  # (any non-NULL value: the query reads no memory)
        r = _DW_FUSED[key] = bool(L.load().cst_gemm_colsum_is_fused(ctypes.byref(d)))
        if len(_DW_FUSED) > 4096:
            _DW_FUSED.clear()
    return r


def layernorm_fwd(x, res, gamma, beta, eps, want_sum=False):
    x = _2d(x)
    rows, cols = x.shape
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (want_sum and res is not None) else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.load().cst_layernorm_fwd(L.ptr(x), L.ptr(res), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(s), L.ptr(mean),
                                       L.ptr(rstd), rows, cols, eps, L.dtype_code(x.dtype), L.stream_ptr()), "cst_layernorm_fwd")
    return y, s, mean, rstd


_EPOCH = [0x3C6EF372]  # (not 0: the stamp buffers are never initialised, and recycled device memory is full of small integers — a dead
#                          tile whose stale word equals the epoch would be visited for nothing; harmless, but not what the stamps are for)


def next_epoch():
    """Unique non-zero 32-bit stamp per producer call (live-tile stamps are never initialised: a tile is live iff stamp == epoch)."""
    _EPOCH[0] = (_EPOCH[0] % 0xFFFFFFFE) + 1
    return _EPOCH[0]


def layernorm_bwd(dy, s, gamma, mean, rstd, dres=None, grad_dtype=torch.float32, want_tiles=False, defer=False):
    """dgamma / dbeta come back in `grad_dtype` (fp32, or the parameter dtype: accumulated in fp32, rounded once).
    want_tiles: also returns (stamps int32 [ceil(rows/64)], epoch) marking the 64-row tiles of dx that are not exactly zero.
    defer: the caller allows the second stage (row-block partials -> dgamma / dbeta) to be left to the deferred-reduction flush."""
    dy, s = _2d(dy), _2d(s)
    rows, cols = s.shape
    lib = L.load()
    dx = torch.empty_like(s)
    dg = torch.empty(cols, dtype=grad_dtype, device=s.device)
    db = torch.empty(cols, dtype=grad_dtype, device=s.device)
    wbytes = lib.cst_layernorm_bwd_workspace(rows, cols)
    defer = bool(defer) and DEFER.on > 0
    ws = torch.empty(wbytes, dtype=torch.uint8, device=s.device) if defer else workspace(wbytes, s.device)
    pg, pb = (None, None) if defer else (L.ptr(dg), L.ptr(db))
    out = (dx, dg, db)
    if want_tiles:
        stamps, epoch = torch.empty((rows + 63) // 64, dtype=torch.int32, device=s.device), next_epoch()
        L.check(lib.cst_layernorm_bwd_tiles(L.ptr(dy), L.ptr(s), L.ptr(gamma), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx), pg,
                                            pb, L.ptr(ws), rows, cols, L.dtype_code(s.dtype), L.dtype_code(grad_dtype),
                                            L.ptr(stamps), epoch, L.stream_ptr()), "cst_layernorm_bwd_tiles")
        out = (dx, dg, db, (stamps, epoch))
    else:
        L.check(lib.cst_layernorm_bwd(L.ptr(dy), L.ptr(s), L.ptr(gamma), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx), pg,
                                      pb, L.ptr(ws), rows, cols, L.dtype_code(s.dtype), L.dtype_code(grad_dtype), L.stream_ptr()),
                "cst_layernorm_bwd")
    if defer:  # partials: fp32 [blocks][2][cols]
        blocks = wbytes // (8 * cols)
        DEFER.push(ws.data_ptr(), dg, 2 * cols, cols, blocks, ws, order=1)
        DEFER.push(ws.data_ptr() + 4 * cols, db, 2 * cols, cols, blocks, ws, order=1)
    return out


def _bhtd_strides(t, layout):
    """Element strides (sb, sh, st) of a [B,T,H*D]-shaped ("bthd") or [T,B,H*D] ("tbhd") activation; d contiguous."""
    assert t.stride(-1) == 1
    if layout == "bt":  # tensor is [B, T, C]
        return t.stride(0), None, t.stride(1)
    return t.stride(1), None, t.stride(0)  # [T, B, C]


def attn_desc(q, k, v, o, lse, H, D, kpm, causal, scale, layout_q="bt", layout_kv="bt", drop_p=0.0, drop_key=0, kv_len=None, seq=None):
    """q,o: [B,Tq,H*D] (layout "bt") or [Tq,B,H*D] ("tb"); k,v likewise with Tk.  Head h at channel offset h*D.
    seq = (seq_offsets int32 [B+1], longest sequence): packed self-attention, q/k/v/o are [1, rows, H*D] (include/cst.h)."""
    d = L.AttnDesc()
    d.dtype = L.dtype_code(q.dtype)
    if layout_q == "bt":
        B, Tq = q.shape[0], q.shape[1]
    else:
        Tq, B = q.shape[0], q.shape[1]
    Tk = k.shape[1] if layout_kv == "bt" else k.shape[0]
    if seq is not None:
        assert layout_q == "bt" and layout_kv == "bt" and B == 1 and kpm is None
        assert seq[0].dtype == torch.int32 and seq[0].is_contiguous()
        B, Tq, Tk = seq[0].numel() - 1, int(seq[1]), int(seq[1])
        d.seq_offsets = seq[0].data_ptr()
        STATS["attn_packed"] = STATS.get("attn_packed", 0) + 1
    else:
        d.seq_offsets = None
    d.B, d.H, d.Tq, d.Tk, d.D = B, H, Tq, Tk, D

    def st(t, lay):
        sb, _, stt = _bhtd_strides(t, lay)
        return sb, D, stt

    d.Q = q.data_ptr(); d.q_sb, d.q_sh, d.q_st = st(q, layout_q)
    d.K = k.data_ptr(); d.k_sb, d.k_sh, d.k_st = st(k, layout_kv)
    d.V = v.data_ptr(); d.v_sb, d.v_sh, d.v_st = st(v, layout_kv)
    d.O = o.data_ptr(); d.o_sb, d.o_sh, d.o_st = st(o, layout_q)
    d.lse = lse.data_ptr()
    if kpm is not None:
        assert kpm.dtype == torch.uint8 and kpm.stride(1) == 1 and kpm.shape == (B, Tk)
        d.key_padding_mask, d.kpm_stride = kpm.data_ptr(), kpm.stride(0)
        bits = kpm_bits(kpm)
        d.kpm_bits = bits.data_ptr()
        d._kpm_bits_owner = bits
    else:
        d.key_padding_mask, d.kpm_stride = None, 0
        d.kpm_bits = None
    d.bwd_ws = None
    d.causal, d.scale = int(causal), scale
    if q.dtype == torch.bfloat16 and D == 64 and not causal and not os.environ.get("CST_ATTN_GENERIC"):
        STATS["attn_fast"] = STATS.get("attn_fast", 0) + 1  # (mirrors attn_fast_ok of csrc/attention.hip: the DMA-staged kernels)
    d.drop_p, d.drop_key = float(drop_p), int(drop_key) & 0xFFFFFFFF
    if kv_len is not None:
        assert (kpm is not None or seq is not None) and kv_len.dtype == torch.int32 and kv_len.numel() == B and kv_len.is_contiguous()
        d.kv_len = kv_len.data_ptr()
        STATS["attn_kv_len"] = STATS.get("attn_kv_len", 0) + 1
    else:
        d.kv_len = None
    return d


_KPM_BITS = []  # [(data_ptr, shape, version, uint8 mask (kept alive), int64 words)]: one padding mask serves every layer of a pass
_BIT_WEIGHTS = {}  # device -> int64 [64]: 1 << k (built on the device: a host list would be a pageable copy, i.e. a stream sync)


def kpm_bits(kpm):
    """uint8 [B, Tk] key padding mask -> int64 [B, ceil(Tk/64)]: bit k of word j = key 64 j + k is masked (bits of keys >= Tk set).
    cst_attn_desc.kpm_bits: the DMA-staged attention kernels read one mask word per key tile from scalar registers.
    Nothing here touches the host: the backward pass asks again with the tensor autograd saved (another Python object over the same
    storage), so the cache goes by storage address + shape + version."""
    key = (kpm.data_ptr(), tuple(kpm.shape), kpm._version)
    for ptr, shape, ver, _, bits in _KPM_BITS:
        if (ptr, shape, ver) == key:
            return bits
    B, Tk = kpm.shape
    nw = (Tk + 63) // 64
    m = kpm != 0
    if nw * 64 != Tk:
        m = torch.cat([m, torch.ones(B, nw * 64 - Tk, dtype=torch.bool, device=kpm.device)], 1)
    w = _BIT_WEIGHTS.get(kpm.device)
    if w is None:
        w = _BIT_WEIGHTS[kpm.device] = torch.ones(64, dtype=torch.int64, device=kpm.device) << torch.arange(64, dtype=torch.int64, device=kpm.device)
    bits = (m.view(B, nw, 64).to(torch.int64) * w).sum(-1).contiguous()  # distinct powers of two: the wrapping sum is the OR
    _KPM_BITS.append(key + (kpm, bits))
    if len(_KPM_BITS) > 8:
        _KPM_BITS.pop(0)
    return bits


def attn_fwd(q, k, v, H, D, kpm, causal, scale, layout_q="bt", layout_kv="bt", drop_p=0.0, drop_key=0, kv_len=None):
    o = torch.empty_like(q)
    if layout_q == "bt":
        B, Tq = q.shape[0], q.shape[1]
    else:
        Tq, B = q.shape[0], q.shape[1]
    lse = torch.empty(B, H, Tq, dtype=torch.float32, device=q.device)
    d = attn_desc(q, k, v, o, lse, H, D, kpm, causal, scale, layout_q, layout_kv, drop_p, drop_key, kv_len)
    L.check(L.load().cst_attn_fwd(ctypes.byref(d), L.stream_ptr()), "cst_attn_fwd")
    return o, lse


def attn_fwd_desc(d):
    L.check(L.load().cst_attn_fwd(ctypes.byref(d), L.stream_ptr()), "cst_attn_fwd")


def attn_bwd_desc(d):
    L.check(L.load().cst_attn_bwd(ctypes.byref(d), L.stream_ptr()), "cst_attn_bwd")


def attn_bwd_fill(d, do, dq, dk, dv, delta, D, layout_q="bt", layout_kv="bt"):
    """Fill the backward half of an attention descriptor (gradient tensors may be strided views, d contiguous)."""
    def st(t, lay):
        sb, _, stt = _bhtd_strides(t, lay)
        return sb, D, stt

    d.dO = do.data_ptr(); d.do_sb, d.do_sh, d.do_st = st(do, layout_q)
    d.dQ = dq.data_ptr(); d.dq_sb, d.dq_sh, d.dq_st = st(dq, layout_q)
    d.dK = dk.data_ptr(); d.dk_sb, d.dk_sh, d.dk_st = st(dk, layout_kv)
    d.dV = dv.data_ptr(); d.dv_sb, d.dv_sh, d.dv_st = st(dv, layout_kv)
    d.delta = delta.data_ptr()
    # live-tile flags (cst_attn_desc.q_flags): the backward stops at the last 64-query tile with a non-zero upstream gradient
    flags = torch.empty(int(d.B) * int(d.H) * ((int(d.Tq) + 63) // 64), dtype=torch.uint8, device=do.device)  # d.B / d.Tq: sequences / longest one when packed
    d.q_flags = flags.data_ptr()
    STATS["attn_q_flags"] = STATS.get("attn_q_flags", 0) + 1
    d._q_flags_owner = flags  # the workspace lives as long as the descriptor (i.e. until the launch has been enqueued)
    # statistics workspace of the DMA-staged backward kernels (cst_attn_desc.bwd_ws)
    ws = torch.empty(int(L.load().cst_attn_bwd_workspace(ctypes.byref(d))) // 4, dtype=torch.float32, device=do.device)
    d.bwd_ws = ws.data_ptr()
    d._bwd_ws_owner = ws


def attn_bwd(do, q, k, v, o, lse, H, D, kpm, causal, scale, layout_q="bt", layout_kv="bt", drop_p=0.0, drop_key=0, kv_len=None):
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse)
    d = attn_desc(q, k, v, o, lse, H, D, kpm, causal, scale, layout_q, layout_kv, drop_p, drop_key, kv_len)
    attn_bwd_fill(d, do, dq, dk, dv, delta, D, layout_q, layout_kv)
    L.check(L.load().cst_attn_bwd(ctypes.byref(d), L.stream_ptr()), "cst_attn_bwd")
    return dq, dk, dv


def conv0_fwd(wav, w, gamma, beta, k, stride, eps=1e-5, frame_limit=None):
    """wav [B,S] fp32; w [C,k]; -> y [B,L,C] channels-last, (mean, rstd, gram) saved for backward.  frame_limit (int32 [B]): frames
    from there on are unread and left unwritten (include/cst.h)."""
    assert wav.dtype == torch.float32 and wav.is_contiguous()
    B, S = wav.shape
    C = w.shape[0]
    Lo = (S - k) // stride + 1
    y = torch.empty(B, Lo, C, dtype=w.dtype, device=wav.device)
    mean = torch.empty(B, C, dtype=torch.float32, device=wav.device)
    rstd = torch.empty(B, C, dtype=torch.float32, device=wav.device)
    gram = torch.empty(B, k * k + k, dtype=torch.float32, device=wav.device)
    lib = L.load()
    ws = workspace(lib.cst_conv0_fwd_workspace(B, S, k, stride), wav.device)
    L.check(lib.cst_conv0_gn_gelu_fwd(L.ptr(wav), L.ptr(w), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                      L.ptr(gram), L.ptr(ws), _lim(frame_limit, B), B, S, C, k, stride, eps, L.dtype_code(w.dtype),
                                      L.stream_ptr()), "cst_conv0_gn_gelu_fwd")
    return y, mean, rstd, gram


def _lim(frame_limit, B):
    if frame_limit is None:
        return None
    assert frame_limit.dtype == torch.int32 and frame_limit.numel() == B and frame_limit.is_contiguous()
    STATS["conv0_frame_limit"] = STATS.get("conv0_frame_limit", 0) + 1
    return L.ptr(frame_limit)


def conv0_bwd(dy, wav, w, gamma, beta, mean, rstd, gram, k, stride, frame_limit=None):
    B, S = wav.shape
    C = w.shape[0]
    lib = L.load()
    dw = torch.empty(C, k, dtype=torch.float32, device=wav.device)
    dg = torch.empty(C, dtype=torch.float32, device=wav.device)
    db = torch.empty(C, dtype=torch.float32, device=wav.device)
    ws = workspace(lib.cst_conv0_bwd_workspace(B, S, C, k, stride), wav.device)
    L.check(lib.cst_conv0_gn_gelu_bwd(L.ptr(dy), L.ptr(wav), L.ptr(w), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd),
                                      L.ptr(gram), L.ptr(dw), L.ptr(dg), L.ptr(db), L.ptr(ws), _lim(frame_limit, B), B, S, C, k, stride,
                                      L.dtype_code(w.dtype), L.stream_ptr()), "cst_conv0_gn_gelu_bwd")
    return dw, dg, db


def glu_fwd(z):
    z = _2d(z)
    rows, C2 = z.shape
    y = torch.empty(rows, C2 // 2, dtype=z.dtype, device=z.device)
    L.check(L.load().cst_glu_fwd(L.ptr(z), L.ptr(y), rows, C2 // 2, L.dtype_code(z.dtype), L.stream_ptr()), "cst_glu_fwd")
    return y


def glu_bwd(dy, z):
    dy, z = _2d(dy), _2d(z)
    rows, C2 = z.shape
    dz = torch.empty_like(z)
    L.check(L.load().cst_glu_bwd(L.ptr(dy), L.ptr(z), L.ptr(dz), rows, C2 // 2, L.dtype_code(z.dtype), L.stream_ptr()), "cst_glu_bwd")
    return dz


def act_bwd(dy, z, act):
    assert dy.is_contiguous() and z.is_contiguous()
    dx = torch.empty_like(z)
    L.check(L.load().cst_act_bwd(L.ptr(dy), L.ptr(z), L.ptr(dx), z.numel(), act, L.dtype_code(z.dtype), L.stream_ptr()), "cst_act_bwd")
    return dx


def act_fwd(x, act):
    assert x.is_contiguous()
    y = torch.empty_like(x)
    L.check(L.load().cst_act_fwd(L.ptr(x), L.ptr(y), x.numel(), act, L.dtype_code(x.dtype), L.stream_ptr()), "cst_act_fwd")
    return y


def colsum(x, out_dtype=torch.float32, live=None, defer=False):
    """Column sums (bias gradients) in `out_dtype`: row-chunk fp32 partials + a fixed-order reduce that writes the parameter
    dtype (cst_colsum_typed) — deterministic, and two launches where the atomics version needs three (zero-fill, kernel, dtype
    conversion).  (A single-launch variant with a last-arriving-block finalize was measured earlier: the agent-scope release
    every workgroup needs before its ticket made it slower, elementwise 4.4 -> 7.4 ms per update.)
    defer: the caller allows the second stage to wait for the deferred-reduction flush (the result is a parameter gradient)."""
    x = _2d(x)
    rows, cols = x.shape
    lib = L.load()
    out = torch.empty(cols, dtype=out_dtype, device=x.device)
    wbytes = lib.cst_colsum_workspace(rows, cols)
    defer = bool(defer) and DEFER.on > 0
    ws = torch.empty(wbytes, dtype=torch.uint8, device=x.device) if defer else workspace(wbytes, x.device)
    po = None if defer else L.ptr(out)
    if live is not None:  # (stamps, epoch) of x's 64-row tiles: dead tiles are all zero and skipped
        assert live[0].numel() * 64 >= rows
        L.check(lib.cst_colsum_typed_live(L.ptr(x), x.stride(0), po, L.ptr(ws), rows, cols, L.dtype_code(x.dtype),
                                          L.dtype_code(out_dtype), L.ptr(live[0]), live[1], L.stream_ptr()), "cst_colsum_typed_live")
    else:
        L.check(lib.cst_colsum_typed(L.ptr(x), x.stride(0), po, L.ptr(ws), rows, cols, L.dtype_code(x.dtype), L.dtype_code(out_dtype),
                                     L.stream_ptr()), "cst_colsum_typed")
    if defer:  # partials: fp32 [chunks][cols], the order of colsum_reduce_kernel = order 1 of cst_reduce_multi
        DEFER.push(ws.data_ptr(), out, cols, cols, wbytes // (4 * cols), ws, order=1)
    return out


def dropout_colsum(x, p, key, out_dtype=torch.float32, live=None, defer=False):
    """(x * mask / (1 - p), column sums of that) in one pass (cst_dropout_colsum): the masked gradient of a dropped Linear output and
    its bias gradient.  Same bits as dropout() followed by colsum().  defer: as in colsum()."""
    x = _2d(x)
    assert x.is_contiguous()
    rows, cols = x.shape
    lib = L.load()
    xd = torch.empty_like(x)
    out = torch.empty(cols, dtype=out_dtype, device=x.device)
    wbytes = lib.cst_colsum_workspace(rows, cols)
    defer = bool(defer) and DEFER.on > 0
    ws = torch.empty(wbytes, dtype=torch.uint8, device=x.device) if defer else workspace(wbytes, x.device)
    L.check(lib.cst_dropout_colsum(L.ptr(x), L.ptr(xd), None if defer else L.ptr(out), L.ptr(ws), rows, cols, L.dtype_code(x.dtype),
                                   L.dtype_code(out_dtype), float(p), int(key) & 0xFFFFFFFF, L.ptr(live[0]) if live is not None else None,
                                   live[1] if live is not None else 0, L.stream_ptr()), "cst_dropout_colsum")
    if defer:
        DEFER.push(ws.data_ptr(), out, cols, cols, wbytes // (4 * cols), ws, order=1)
    return xd, out


def colsum_atomic(x):
    """cst_colsum: the atomics version (fp32 output), kept in the ABI."""
    x = _2d(x)
    rows, cols = x.shape
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    L.check(L.load().cst_colsum(L.ptr(x), x.stride(0), L.ptr(out), rows, cols, L.dtype_code(x.dtype), L.stream_ptr()), "cst_colsum")
    return out


def col2im1d(dcol, z, B, Lin, Lout, C, k, stride, pad, dact):
    dx = torch.empty(B, Lin, C, dtype=dcol.dtype, device=dcol.device)
    L.check(L.load().cst_col2im1d(L.ptr(dcol), L.ptr(z), L.ptr(dx), B, Lin, Lout, C, k, stride, pad, dact,
                                  L.dtype_code(dcol.dtype), L.stream_ptr()), "cst_col2im1d")
    return dx


def mask_rows(x, mask_u8):
    x = _2d(x)
    y = torch.empty_like(x)
    L.check(L.load().cst_mask_rows(L.ptr(x), L.ptr(mask_u8), L.ptr(y), x.shape[0], x.shape[1], L.dtype_code(x.dtype),
                                   L.stream_ptr()), "cst_mask_rows")
    return y


def rows_pack(x, seq_off, n_rows, tail_sum):
    """[B, T, C] -> [n_rows, C] (include/cst.h: cst_rows_pack)."""
    assert x.dim() == 3 and x.is_contiguous() and seq_off.dtype == torch.int32 and seq_off.numel() == x.shape[0] + 1
    out = torch.empty(n_rows, x.shape[2], dtype=x.dtype, device=x.device)
    L.check(L.load().cst_rows_pack(L.ptr(x), L.ptr(seq_off), L.ptr(out), x.shape[0], x.shape[1], x.shape[2], int(bool(tail_sum)), L.dtype_code(x.dtype),
                                   L.stream_ptr()), "cst_rows_pack")
    return out


def rows_unpack(y, seq_off, B, T, tail_broadcast):
    """[n_rows, C] -> [B, T, C] (include/cst.h: cst_rows_unpack)."""
    assert y.dim() == 2 and y.is_contiguous() and seq_off.dtype == torch.int32 and seq_off.numel() == B + 1
    out = torch.empty(B, T, y.shape[1], dtype=y.dtype, device=y.device)
    L.check(L.load().cst_rows_unpack(L.ptr(y), L.ptr(seq_off), L.ptr(out), B, T, y.shape[1], int(bool(tail_broadcast)), L.dtype_code(y.dtype),
                                     L.stream_ptr()), "cst_rows_unpack")
    return out


def dropout(x, p, key):
    """y = x * keep(key, idx) / (1 - p); x contiguous."""
    assert x.is_contiguous()
    y = torch.empty_like(x)
    L.check(L.load().cst_dropout(L.ptr(x), L.ptr(y), x.numel(), float(p), int(key) & 0xFFFFFFFF, L.dtype_code(x.dtype), L.stream_ptr()),
            "cst_dropout")
    return y


def conv_row_limits(nz_last, conv_spec, samples):
    """int32 [L, 1 + smax, B] live-frame limits of every conv layer from the real frame counts behind the last one (include/cst.h:
    cst_conv_row_limits) — one launch instead of ~15 tiny torch kernels per layer."""
    import ctypes
    Lc = len(conv_spec)
    ks = (ctypes.c_int32 * Lc)(*[int(c[1]) for c in conv_spec])
    st = (ctypes.c_int32 * Lc)(*[int(c[2]) for c in conv_spec])
    smax = max([int(c[2]) for c in conv_spec[1:]] or [1])
    B = nz_last.numel()
    assert nz_last.dtype == torch.int32 and nz_last.is_contiguous()
    out = torch.empty(Lc, 1 + smax, B, dtype=torch.int32, device=nz_last.device)
    L.check(L.load().cst_conv_row_limits(L.ptr(nz_last), ks, st, Lc, int(samples), L.ptr(out), B, smax, L.stream_ptr()), "cst_conv_row_limits")
    return out


def dropout_scale(x, alpha, p, key):
    """y = alpha * x * keep(key, idx) / (1 - p); x contiguous, numel % 8 == 0."""
    assert x.is_contiguous()
    y = torch.empty_like(x)
    L.check(L.load().cst_dropout_scale(L.ptr(x), L.ptr(y), x.numel(), float(alpha), float(p), int(key) & 0xFFFFFFFF, L.dtype_code(x.dtype),
                                       L.stream_ptr()), "cst_dropout_scale")
    return y


def embed_pos_fwd(tokens, pad_mask, embed, x, pos_table, scale, pad_idx, p, key):
    """dropout(scale * (embed[tokens] | x) + pos_table[make_positions(pad_mask | tokens)]) -> [B, T, C] (include/cst.h)."""
    ref = tokens if tokens is not None else (pad_mask if pad_mask is not None else x)
    B, T = ref.shape[0], ref.shape[1]
    src = embed if embed is not None else x
    C = src.shape[-1]
    V = embed.shape[0] if embed is not None else 0
    assert src.is_contiguous() and (tokens is None or (tokens.dtype == torch.int64 and tokens.is_contiguous()))
    assert pad_mask is None or (pad_mask.dtype == torch.uint8 and pad_mask.is_contiguous() and pad_mask.shape == (B, T))
    assert pos_table is None or (pos_table.dtype == torch.float32 and pos_table.is_contiguous() and pos_table.shape[1] == C)
    out = torch.empty(B, T, C, dtype=src.dtype, device=src.device)
    L.check(L.load().cst_embed_pos_fwd(L.ptr(tokens), L.ptr(pad_mask), L.ptr(embed), L.ptr(x), L.ptr(pos_table), float(scale), int(pad_idx),
                                       L.ptr(out), B, T, C, V, 0 if pos_table is None else pos_table.shape[0], float(p), int(key) & 0xFFFFFFFF,
                                       L.dtype_code(src.dtype), L.stream_ptr()), "cst_embed_pos_fwd")
    return out


def embed_bwd(dy, tokens, V, scale, pad_idx, p, key, grad_dtype):
    """Deterministic embedding-table gradient [V, C] of embed_pos_fwd (include/cst.h)."""
    dy = dy.reshape(-1, dy.shape[-1])
    assert dy.is_contiguous() and tokens.is_contiguous() and tokens.numel() == dy.shape[0]
    dE = torch.empty(V, dy.shape[1], dtype=grad_dtype, device=dy.device)
    L.check(L.load().cst_embed_bwd(L.ptr(dy), L.ptr(tokens), L.ptr(dE), float(scale), int(pad_idx), dy.shape[0], dy.shape[1], V, float(p),
                                   int(key) & 0xFFFFFFFF, L.dtype_code(dy.dtype), L.dtype_code(grad_dtype), L.stream_ptr()), "cst_embed_bwd")
    return dE


def ls_ce_fwd(logits, target, eps, pad):
    logits = _2d(logits)
    rows, V = logits.shape
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    ws = workspace(2 * rows * 4, logits.device)
    L.check(L.load().cst_ls_ce_fwd(L.ptr(logits), L.ptr(target), L.ptr(out2), L.ptr(lse), L.ptr(ws), rows, V, eps, pad,
                                   L.dtype_code(logits.dtype), L.stream_ptr()), "cst_ls_ce_fwd")
    return out2, lse


def ls_ce_bwd(logits, target, lse, gscale, eps, pad):
    logits = _2d(logits)
    rows, V = logits.shape
    d = torch.empty_like(logits)
    L.check(L.load().cst_ls_ce_bwd(L.ptr(logits), L.ptr(target), L.ptr(lse), L.ptr(gscale), L.ptr(d), rows, V, eps, pad,
                                   L.dtype_code(logits.dtype), L.stream_ptr()), "cst_ls_ce_bwd")
    return d


def sumsq(x, out):
    lib = L.load()
    ws = workspace(lib.cst_sumsq_workspace(), x.device)
    L.check(lib.cst_sumsq(L.ptr(x), x.numel(), L.ptr(out), L.ptr(ws), L.dtype_code(x.dtype), L.stream_ptr()), "cst_sumsq")


def contrastive_fwd(a, t, temp):
    """a, t [B, M, C] contiguous -> (loss fp32[1], sim, na, nt)."""
    B, M, C = a.shape
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    sim = torch.empty(B, M, M, dtype=torch.float32, device=a.device)
    na = torch.empty(B, M, dtype=torch.float32, device=a.device)
    nt = torch.empty_like(na)
    ws = workspace(B * 4, a.device)
    L.check(L.load().cst_contrastive_fwd(L.ptr(a), L.ptr(t), L.ptr(loss), L.ptr(ws), L.ptr(sim), L.ptr(na), L.ptr(nt), B, M, C, float(temp),
                                         L.dtype_code(a.dtype), L.stream_ptr()), "cst_contrastive_fwd")
    return loss, sim, na, nt


def contrastive_bwd(a, t, sim, na, nt, gscale, temp):
    B, M, C = a.shape
    da, dt = torch.empty_like(a), torch.empty_like(t)
    L.check(L.load().cst_contrastive_bwd(L.ptr(a), L.ptr(t), L.ptr(sim), L.ptr(na), L.ptr(nt), L.ptr(gscale), L.ptr(da), L.ptr(dt), B, M, C,
                                         float(temp), L.dtype_code(a.dtype), L.stream_ptr()), "cst_contrastive_bwd")
    return da, dt


def adam_step(master, m, v, grad, param, lr, beta1, beta2, eps, wd, step, grad_scale):
    L.check(L.load().cst_adam_step(L.ptr(master), L.ptr(m), L.ptr(v), L.ptr(grad), L.ptr(param), master.numel(), lr, beta1, beta2,
                                   eps, wd, step, L.ptr(grad_scale), L.dtype_code(grad.dtype), L.dtype_code(param.dtype),
                                   L.stream_ptr()), "cst_adam_step")


def weight_norm_fwd(v, g):
    """v [..., C] contiguous, g [C] -> (w = v * g / ||v||_{all but last dim}, norm fp32 [C])."""
    C = v.shape[-1]
    R = v.numel() // C
    lib = L.load()
    w = torch.empty_like(v)
    norm = torch.empty(C, dtype=torch.float32, device=v.device)
    ws = torch.empty(int(lib.cst_weight_norm_workspace(R, C)) // 4, dtype=torch.float32, device=v.device)
    L.check(lib.cst_weight_norm_fwd(v.data_ptr(), g.data_ptr(), w.data_ptr(), norm.data_ptr(), ws.data_ptr(), R, C, L.dtype_code(v.dtype),
                                    L.stream_ptr()), "cst_weight_norm_fwd")
    return w, norm


def weight_norm_bwd(v, g, dw, norm):
    C = v.shape[-1]
    R = v.numel() // C
    lib = L.load()
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    ws = torch.empty(int(lib.cst_weight_norm_workspace(R, C)) // 4, dtype=torch.float32, device=v.device)
    L.check(lib.cst_weight_norm_bwd(v.data_ptr(), g.data_ptr(), dw.data_ptr(), norm.data_ptr(), dv.data_ptr(), dg.data_ptr(), ws.data_ptr(), R, C,
                                    L.dtype_code(v.dtype), L.stream_ptr()), "cst_weight_norm_bwd")
    return dv, dg


def transpose2d(x, out=None):
    """[R, C] contiguous -> [C, R] contiguous (cst_transpose2d)."""
    R, C = x.shape
    if out is None:
        out = torch.empty(C, R, dtype=x.dtype, device=x.device)
    assert out.shape == (C, R) and out.is_contiguous() and out.dtype == x.dtype
    L.check(L.load().cst_transpose2d(x.data_ptr(), out.data_ptr(), R, C, L.dtype_code(x.dtype), L.stream_ptr()), "cst_transpose2d")
    return out


def transpose_table(pairs):
    """[(src [R, C], dst [C, R]), ...] of one dtype / device -> (device table, n, total tiles, dtype code) for transpose2d_multi.
    The table reaches the device through pinned memory (a pageable copy would drain the stream)."""
    rows, tile0 = [], 0
    for src, dst in pairs:
        R, C = src.shape
        assert src.is_contiguous() and dst.is_contiguous() and dst.shape == (C, R) and dst.dtype == src.dtype
        rows.append([src.data_ptr(), dst.data_ptr(), R, C, tile0])
        tile0 += ((R + 63) // 64) * ((C + 63) // 64)
    host = torch.tensor(rows, dtype=torch.int64).pin_memory()
    return host.to(pairs[0][0].device, non_blocking=True), len(rows), tile0, L.dtype_code(pairs[0][0].dtype), host


def transpose2d_multi(table):
    """Every matrix of a transpose_table in one launch (cst_transpose2d_multi)."""
    dev, n, total, dt = table[:4]
    L.check(L.load().cst_transpose2d_multi(dev.data_ptr(), n, total, dt, L.stream_ptr()), "cst_transpose2d_multi")
