"""torch.autograd.Function seam over the C-ABI kernels (the reference's seam on this
path is nn.Module.forward / autograd.Function calling ATen; here they call libcst_hip.so).

Every function requires CUDA(HIP) tensors; there is no CPU path."""
import os as _os
import weakref as _weakref

import torch

from . import kernels as K
from . import lib as L

_ACT = {None: L.ACT_NONE, "none": L.ACT_NONE, "relu": L.ACT_RELU, "gelu": L.ACT_GELU}


def _flat2d(x):
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.is_contiguous() else x2.contiguous()


# ------------------------------------------------------------------------------------------------
# Linear (+bias +activation +residual) — F.linear call sites listed in include/cst.h
# ------------------------------------------------------------------------------------------------
def _vec(dtype):
    return 8 if dtype == torch.bfloat16 else 4


# Live-tile stamps: a LayerNorm backward knows, for free, which 64-row tiles of the gradient it writes are not exactly zero (the
# rows of padded frames are).  The stamps ride on the gradient tensor as an attribute (PyObject preservation keeps it on the
# way to the next node's backward; any op in between — an accumulation, a copy — yields a fresh tensor without it, which only
# means "no skipping") and the weight-gradient GEMMs that reduce over those rows skip the dead K blocks (cst_gemm_desc.k_live).
def _with_tiles(t, tiles):
    if t.is_contiguous():  # row r of the producer = row r of t flattened to [rows, cols]
        t._cst_live_tiles = (tiles[0], tiles[1], t.data_ptr(), t.numel(), t.shape[-1], t._version)
        if t._base is not None:  # views made further down the graph report the ultimate base, not `t`
            t._base._cst_live_tiles = t._cst_live_tiles
    return t


def _tiles_of(t, rows):
    """The stamps of `t` if — and only if — t is still the very tensor they were made for: same storage, same extent, same row
    width, contiguous (so that _flat2d(t) is a view with the producer's row order) and not written since (the version counter is
    shared by a base and its views; autograd may accumulate a second incoming gradient into the first one IN PLACE)."""
    lt = getattr(t, "_cst_live_tiles", None)
    if lt is None and t._base is not None:  # a view (e.g. the [T,B,C] <-> [B,T,C] transposes of the layout seam, undone again)
        lt = getattr(t._base, "_cst_live_tiles", None)
    if (lt is None or not t.is_contiguous() or lt[2] != t.data_ptr() or lt[3] != t.numel() or lt[4] != t.shape[-1]
            or lt[5] != t._version or lt[0].numel() != (rows + 63) // 64):
        return None
    return (lt[0], lt[1])


_WT_MIN_ROWS = int(_os.environ.get("CST_WT_MIN_ROWS", 512))  # token rows from which the dX GEMM takes a transposed weight copy
# (4064-row decoder shapes: 4064 x 512 x 512 20.9 us as an mn-major read, 7.6 us through the copy; the copies cost one launch per update)


def _want_wt(M, w):
    return (M >= _WT_MIN_ROWS and w.is_cuda and w.dim() == 2 and w.is_contiguous() and w.shape[0] % _vec(w.dtype) == 0
            and w.shape[1] % _vec(w.dtype) == 0 and not _os.environ.get("CST_NO_WT"))


def _wversion(w):
    """Autograd version of a weight operand; the stacked q | k | v view (stacked_rows) shares only q's counter, so it carries its
    three parameters and reports the sum of theirs."""
    parts = getattr(w, "_cst_parts", None)
    return w._version if parts is None else sum(p._version for p in parts)


class _WeightTransposes:
    """W^T copies of the Linear weights the dX GEMMs read (see _linear_backward), kept across updates.  A weight changes only in the
    optimizer step (optim.PARAM_EPOCH counts every raw-pointer write of the flat parameter buffer) or through an in-place torch op
    (its autograd version): the first request after an optimizer step refreshes EVERY registered copy in one launch
    (cst_transpose2d_multi — one launch per update instead of one small launch per Linear per backward pass: 0.75 -> 0.15 ms per
    update on the bench configuration, ~110 launches fewer); a weight changed any other way is re-transposed on its own when it is
    asked for.

    Only weights that outlive an update are cached: views of a flat parameter buffer registered by optim.FlatParamBuffers (the
    entries go when the buffers object is collected) and free-standing nn.Parameters (held weakly).  Anything else — the torch.cat
    fallback of stacked_rows, a zero-padded odd-width weight — is transposed into a fresh tensor and forgotten."""

    def __init__(self):
        self.entries = {}     # (data_ptr, shape, dtype) -> [src: detached w | weakref to a Parameter, wt, version, epoch, storage ptr]
        self.tables = {}      # (device, dtype) -> transpose table over the entries of that kind
        self.refreshes = 0    # whole-table launches (tests)
        self.persistent = {}  # storage data_ptr of a registered flat parameter buffer -> weakref to its owner

    def register_storage(self, owner, flat):
        sp = flat.untyped_storage().data_ptr()
        self.persistent[sp] = _weakref.ref(owner)
        _weakref.finalize(owner, self._drop_storage, sp)

    def _drop_storage(self, sp):
        self.persistent.pop(sp, None)
        for k in [k for k, e in self.entries.items() if e[4] == sp]:
            del self.entries[k]
        self.tables.clear()

    def invalidate(self):
        """Forget every copy (between models of one process: bench.py's extra legs, test suites)."""
        self.entries.clear()
        self.tables.clear()

    @staticmethod
    def _src(e):
        return e[0]() if isinstance(e[0], _weakref.ref) else e[0]

    def get(self, w):
        from .optim import PARAM_EPOCH
        epoch = PARAM_EPOCH[0]
        key = (w.data_ptr(), tuple(w.shape), w.dtype)
        e = self.entries.get(key)
        if e is not None and self._src(e) is None:  # a dead Parameter whose address was handed out again
            del self.entries[key]
            self.tables.pop((w.device, w.dtype), None)
            e = None
        if e is None:
            sp = w.untyped_storage().data_ptr()
            owner = self.persistent.get(sp)
            if owner is not None and owner() is not None:
                src = w.detach()
                parts = getattr(w, "_cst_parts", None)
                if parts is not None:
                    src._cst_parts = parts
            elif isinstance(w, torch.nn.Parameter):
                src, sp = _weakref.ref(w), None
            else:
                return K.transpose2d(w)  # a temporary: not cached
            e = self.entries[key] = [src, K.transpose2d(w), _wversion(w), epoch, sp]
            self.tables.pop((w.device, w.dtype), None)
            return e[1]
        if e[2] != _wversion(w):  # written by a torch op since (load_state_dict, manual edits)
            K.transpose2d(w, e[1])
            e[2], e[3] = _wversion(w), epoch
            return e[1]
        if e[3] != epoch:  # an optimizer step since: refresh every stale copy of this kind at once
            kind = (w.device, w.dtype)
            stale = []
            for k, v in list(self.entries.items()):
                t = self._src(v)
                if t is None or t.data_ptr() != k[0]:
                    # dead, or a Parameter re-homed since it was cached (FlatParamBuffers: `p.data = view`): it is asked for under
                    # its new address from now on; this entry would be re-transposed every update into a copy nobody reads
                    del self.entries[k]
                    self.tables.pop(kind, None)
                elif t.device == w.device and k[2] == w.dtype and v[3] != epoch:
                    stale.append(v)
            ids = [id(v) for v in stale]
            t = self.tables.get(kind)
            if t is None or t[5] != ids:  # (the table is rebuilt only when the set of weights changed: first updates of a run)
                t = self.tables[kind] = K.transpose_table([(self._src(v), v[1]) for v in stale]) + (ids,)
            K.transpose2d_multi(t)
            self.refreshes += 1
            for v in stale:
                v[2], v[3] = _wversion(self._src(v)), epoch
        return e[1]


WEIGHT_TRANSPOSES = _WeightTransposes()


def _may_defer(*params):
    """May the second stage of a reduction that produces the gradients of `params` wait for the deferred-reduction flush
    (kernels.DEFER)?  Only while the mode is on, none of them is shared between modules (trainer.py marks those: a second gradient
    would be added to the first at once) and none is being accumulated into (a later micro-batch of an update)."""
    if not K.DEFER.on:
        return False
    for w in params:
        if w is None:
            continue
        for p in (getattr(w, "_cst_parts_params", None) or (w,)):
            if not isinstance(p, torch.nn.Parameter) or p.grad is not None or getattr(p, "_cst_shared", False):
                return False
    return True


def _grad_out(w):
    """The flat-gradient-buffer slot of weight `w` as the output tensor of its weight-gradient GEMM (optim.grad_slot), or None."""
    if _os.environ.get("CST_NO_GRAD_SLOT"):
        return None
    from .optim import grad_slot
    return grad_slot(w)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, resid, act, drop_p, drop_key):
        """y = dropout(act(x W^T + b)) + resid  (dropout fused into the GEMM epilogue, before the residual add)."""
        x2 = _flat2d(x)
        M, Kd = x2.shape
        N = weight.shape[0]
        w = weight if weight.is_contiguous() else weight.contiguous()
        ctx.bias_obj = bias  # (the Parameter / stacked view itself: _may_defer looks at its .grad and marks in backward)
        # odd output widths (e.g. a 60-symbol test vocabulary in bf16): zero-pad the rows of W up to the 16-byte vector
        # the dX / dW loaders need; real shapes (512/768/1024/2048/3072/10000) never take this branch.
        Np = (N + _vec(x.dtype) - 1) // _vec(x.dtype) * _vec(x.dtype)
        if Np != N:
            wp = torch.zeros(Np, Kd, dtype=w.dtype, device=w.device)
            wp[:N] = w
            w = wp
            if bias is not None:
                bp = torch.zeros(Np, dtype=bias.dtype, device=bias.device)
                bp[:N] = bias
                bias = bp
            assert resid is None and drop_p == 0.0
        y = torch.empty(M, Np, dtype=x.dtype, device=x.device)
        z = torch.empty_like(y) if act != L.ACT_NONE else None
        r2 = _flat2d(resid) if resid is not None else None
        K.gemm(x2, w, y, M, Np, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=Np, bias=bias, act=act, aux_out=z, ld_aux_out=Np,
               resid=r2, ld_resid=Np, split_k=1, drop_p=drop_p, drop_key=drop_key)
        ctx.save_for_backward(x2, w, z)
        ctx.act, ctx.has_bias, ctx.has_resid = act, bias is not None, resid is not None
        ctx.drop = (drop_p, drop_key)
        ctx.xshape, ctx.N = x.shape, N
        if Np != N:
            y = y[:, :N].contiguous()
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        return _linear_backward(ctx, dy, None)


def _linear_backward(ctx, dy, dxp):
    """Shared by _LinearFn / _LinearPassFn.  dxp: gradient that reached the pass-through alias of x (the residual branch of the
    block this projection opens); it is added in the dX GEMM's epilogue instead of by an autograd accumulation kernel."""
    x2, w, z = ctx.saved_tensors
    M, Kd = x2.shape
    Np, N = w.shape[0], ctx.N
    if dy is None:  # only the pass-through was consumed
        return dxp, None, None, None, None, None, None
    dy2 = _flat2d(dy)
    live = _tiles_of(dy, M)  # rows of dy that are exactly zero stay zero through the mask / activation derivative below
    if Np != N:
        dyp = torch.zeros(M, Np, dtype=dy2.dtype, device=dy2.device)
        dyp[:, :N] = dy2
        dy2 = dyp
    db_fused = None
    want_db = ctx.has_bias and ctx.needs_input_grad[2]
    # the bias gradient rides in the weight-gradient GEMM when that GEMM's kernel builds it for free (cst_gemm_desc.colsum)
    db_in_dw = want_db and ctx.needs_input_grad[1] and Np % 8 == 0 and not _os.environ.get("CST_NO_GEMM_COLSUM") and K.dw_colsum_is_fused(Np, Kd, M, w.dtype)
    may_b = want_db and Np == N and _may_defer(w, ctx.bias_obj)  # the bias gradient's second stage may wait for the flush
    if ctx.drop[0] > 0.0:  # gradient of the dropped branch: the same mask, regenerated
        dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
        if not db_in_dw and ctx.act == L.ACT_NONE and want_db and Np == N and N % 8 == 0 and not _os.environ.get("CST_NO_DROP_COLSUM"):
            dy2, db_fused = K.dropout_colsum(dyc, ctx.drop[0], ctx.drop[1], w.dtype, live, defer=may_b)  # mask + bias gradient in one pass
        else:
            dy2 = K.dropout(dyc, *ctx.drop)
    dz = K.act_bwd(dy2, z, ctx.act) if ctx.act != L.ACT_NONE else dy2
    dx = dw = db = dres = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty(M, Kd, dtype=x2.dtype, device=x2.device)
        if _want_wt(M, w):
            # large token counts: dY W with W^T [K_in, N_out] as a k-major B operand — both operands are then read the way the
            # forward GEMM reads them (the transpose reads of an mn-major W measured 15-20 % slower on the 31 760-row shapes); the
            # transposed copy of a <= 5 MB weight costs a few microseconds
            K.gemm(dz, WEIGHT_TRANSPOSES.get(w), dx, M, Kd, Np, a_kmajor=1, b_kmajor=1, lda=Np, ldb=Np, ldc=Kd, split_k=1,
                   resid=_flat2d(dxp) if dxp is not None else None, ld_resid=Kd, m_live=live)
        else:
            K.gemm(dz, w, dx, M, Kd, Np, a_kmajor=1, b_kmajor=0, lda=Np, ldb=Kd, ldc=Kd, split_k=1,
                   resid=_flat2d(dxp) if dxp is not None else None, ld_resid=Kd, m_live=live)
        dx = dx.view(ctx.xshape)
        if live is not None and dxp is None:
            dx = _with_tiles(dx, live)  # a zero row of dz is a zero row of dz W
    if ctx.needs_input_grad[1]:
        dw = _grad_out(w) if Np == N else None
        if dw is None:
            dw = torch.empty(Np, Kd, dtype=w.dtype, device=w.device)
        if db_in_dw:
            db = torch.empty(Np, dtype=w.dtype, device=w.device)
        may = Np == N and _may_defer(w, ctx.bias_obj if db_in_dw else None)
        # (a small layer's weight gradient runs on the side stream, off the dX chain — kernels.side_gemm; not when dz IS the incoming
        #  gradient that also travels on as the residual gradient: something upstream might add into it in place)
        big = not may or not K.dw_colsum_is_fused(Np, Kd, M, w.dtype) or (dz.data_ptr() == dy.data_ptr() and ctx.has_resid)
        K.side_gemm(big, (dz, x2, live[0] if live is not None else None), dz, x2, dw, Np, Kd, M, a_kmajor=0, b_kmajor=0, lda=Np, ldb=Kd, ldc=Kd,
                    split_k=-1, k_live=live, colsum=db, defer=may)
        dw = dw[:N]
        if db is not None:
            db = db[:N]
    if want_db and db is None:
        db = db_fused if db_fused is not None else K.colsum(dz, w.dtype, live, defer=may_b)[:N]
    if ctx.has_resid and ctx.needs_input_grad[3]:
        dres = dy
    return dx, dw, db, dres, None, None, None


class _LinearPassFn(torch.autograd.Function):
    """(y, x') = (linear(x), alias of x).  A post-norm block takes x' as its residual: the residual-branch gradient and the
    projection's dX then arrive in ONE backward call and are summed in the dX GEMM's epilogue (same idea as _LayerNormPassFn)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, drop_p, drop_key):
        y = _LinearFn.forward(ctx, x, weight, bias, None, act, drop_p, drop_key)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxp):
        g = _linear_backward(ctx, dy, dxp)
        return g[0], g[1], g[2], None, None, None


def linear_pass(x, weight, bias=None, act=None, dropout_p=0.0):
    """Returns (linear(x), x') where x' aliases x and must be used for the enclosing block's residual connection."""
    return _LinearPassFn.apply(x, weight, bias, _ACT[act], *_drop_args(dropout_p))


def linear(x, weight, bias=None, act=None, resid=None, dropout_p=0.0):
    """y = dropout(act(x W^T + b)) (+ resid)."""
    return _LinearFn.apply(x, weight, bias, resid, _ACT[act], *_drop_args(dropout_p))


class _FFNFn(torch.autograd.Function):
    """y = W2 act(W1 x + b1) + b2 (+ resid): two GEMMs forward; in backward the activation derivative is the epilogue of
    the dH GEMM (dz1 = (dy W2) * act'(z1)), so no separate act-backward pass over the [tokens, ffn] tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, resid, act, p_act, key_act, p_out, key_out):
        """y = resid + dropout_out(W2 dropout_act(act(W1 x + b1)) + b2): both dropouts are GEMM epilogues."""
        x2 = _flat2d(x)
        M, d = x2.shape
        F_ = w1.shape[0]
        dout = w2.shape[0]
        z1 = torch.empty(M, F_, dtype=x.dtype, device=x.device)
        h = torch.empty_like(z1)
        K.gemm(x2, w1, h, M, F_, d, a_kmajor=1, b_kmajor=1, lda=d, ldb=d, ldc=F_, bias=b1, act=act, aux_out=z1, ld_aux_out=F_, split_k=1,
               drop_p=p_act, drop_key=key_act)
        y = torch.empty(M, dout, dtype=x.dtype, device=x.device)
        r2 = _flat2d(resid) if resid is not None else None
        K.gemm(h, w2, y, M, dout, F_, a_kmajor=1, b_kmajor=1, lda=F_, ldb=F_, ldc=dout, bias=b2, resid=r2, ld_resid=dout, split_k=1,
               drop_p=p_out, drop_key=key_out)
        ctx.save_for_backward(x2, w1, w2, z1, h)
        ctx.cfg = (act, b1 is not None, b2 is not None, resid is not None, x.shape)
        # post-norm block: the residual IS the block input -> its gradient (dy) is added in the dX GEMM's epilogue
        ctx.res_is_x = (resid is not None and resid.data_ptr() == x.data_ptr() and resid.shape == x.shape
                        and resid.stride() == x.stride() and dout == d)
        ctx.drop = (p_act, key_act, p_out, key_out)
        ctx.bias_objs = (b1, b2)
        return y.view(*x.shape[:-1], dout)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, z1, h = ctx.saved_tensors
        act, has_b1, has_b2, has_res, xshape = ctx.cfg
        M, d = x2.shape
        F_, dout = w1.shape[0], w2.shape[0]
        dy2 = _flat2d(dy)
        live = _tiles_of(dy, M)  # zero rows of dy are zero rows of dy2 (mask) and of dz1 (row-wise GEMM, act', mask)
        p_act, key_act, p_out, key_out = ctx.drop
        b1, b2 = ctx.bias_objs
        db2_fused = None
        gcs = not _os.environ.get("CST_NO_GEMM_COLSUM")
        db2_in_dw = gcs and has_b2 and ctx.needs_input_grad[4] and ctx.needs_input_grad[3] and dout % 8 == 0 and K.dw_colsum_is_fused(dout, F_, M, w2.dtype)
        db1_in_dw = gcs and has_b1 and ctx.needs_input_grad[2] and ctx.needs_input_grad[1] and F_ % 8 == 0 and K.dw_colsum_is_fused(F_, d, M, w1.dtype)
        if p_out > 0.0:  # d(fc2 output) = dy * mask_out
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            if not db2_in_dw and has_b2 and ctx.needs_input_grad[4] and dout % 8 == 0 and not _os.environ.get("CST_NO_DROP_COLSUM"):
                dy2, db2_fused = K.dropout_colsum(dyc, p_out, key_out, w2.dtype, live, defer=_may_defer(w2, b2))  # mask + fc2 bias gradient in one pass
            else:
                dy2 = K.dropout(dyc, p_out, key_out)
        dz1 = torch.empty(M, F_, dtype=dy2.dtype, device=dy2.device)
        wt = _want_wt(M, w2) and _want_wt(M, w1)  # dX GEMMs with the weights as k-major B operands (see _linear_backward)
        if wt:
            K.gemm(dy2, WEIGHT_TRANSPOSES.get(w2), dz1, M, F_, dout, a_kmajor=1, b_kmajor=1, lda=dout, ldb=dout, ldc=F_, dact=act, aux_in=z1, ld_aux_in=F_,
                   split_k=1, drop_p=p_act, drop_key=key_act, m_live=live)
        else:
            K.gemm(dy2, w2, dz1, M, F_, dout, a_kmajor=1, b_kmajor=0, lda=dout, ldb=F_, ldc=F_, dact=act, aux_in=z1, ld_aux_in=F_, split_k=1,
                   drop_p=p_act, drop_key=key_act, m_live=live)
        dx = dw1 = db1 = dw2 = db2 = None
        if ctx.needs_input_grad[3]:
            dw2 = _grad_out(w2)
            if dw2 is None:
                dw2 = torch.empty(dout, F_, dtype=w2.dtype, device=w2.device)
            if db2_in_dw:
                db2 = torch.empty(dout, dtype=w2.dtype, device=w2.device)
            may2 = _may_defer(w2, b2 if db2_in_dw else None)
            big2 = not may2 or not K.dw_colsum_is_fused(dout, F_, M, w2.dtype) or (dy2.data_ptr() == dy.data_ptr() and has_res)
            K.side_gemm(big2, (dy2, h, live[0] if live is not None else None), dy2, h, dw2, dout, F_, M, a_kmajor=0, b_kmajor=0, lda=dout, ldb=F_,
                        ldc=F_, split_k=-1, k_live=live, colsum=db2, defer=may2)
        if has_b2 and ctx.needs_input_grad[4] and db2 is None:
            db2 = db2_fused if db2_fused is not None else K.colsum(dy2, w2.dtype, live, defer=_may_defer(w2, b2))
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, d, dtype=x2.dtype, device=x2.device)
            if wt:
                K.gemm(dz1, WEIGHT_TRANSPOSES.get(w1), dx, M, d, F_, a_kmajor=1, b_kmajor=1, lda=F_, ldb=F_, ldc=d, split_k=1,
                       resid=_flat2d(dy) if ctx.res_is_x else None, ld_resid=d, m_live=live)
            else:
                K.gemm(dz1, w1, dx, M, d, F_, a_kmajor=1, b_kmajor=0, lda=F_, ldb=d, ldc=d, split_k=1,
                       resid=_flat2d(dy) if ctx.res_is_x else None, ld_resid=d, m_live=live)
            dx = dx.view(xshape)
        if ctx.needs_input_grad[1]:
            dw1 = _grad_out(w1)
            if dw1 is None:
                dw1 = torch.empty(F_, d, dtype=w1.dtype, device=w1.device)
            if db1_in_dw:
                db1 = torch.empty(F_, dtype=w1.dtype, device=w1.device)
            may1 = _may_defer(w1, b1 if db1_in_dw else None)
            big1 = not may1 or not K.dw_colsum_is_fused(F_, d, M, w1.dtype)
            K.side_gemm(big1, (dz1, x2, live[0] if live is not None else None), dz1, x2, dw1, F_, d, M, a_kmajor=0, b_kmajor=0, lda=F_, ldb=d,
                        ldc=d, split_k=-1, k_live=live, colsum=db1, defer=may1)
        if has_b1 and ctx.needs_input_grad[2] and db1 is None:
            db1 = K.colsum(dz1, w1.dtype, live, defer=_may_defer(w1, b1))
        dres = dy if has_res and ctx.needs_input_grad[5] and not (ctx.res_is_x and ctx.needs_input_grad[0]) else None
        return dx, dw1, db1, dw2, db2, dres, None, None, None, None, None


def ffn(x, w1, b1, w2, b2, act, resid=None, activation_dropout_p=0.0, dropout_p=0.0):
    """Position-wise feed-forward block with fused epilogues (fc1: bias+activation+dropout, fc2: bias+dropout+residual)."""
    assert _ACT[act] != L.ACT_NONE
    return _FFNFn.apply(x, w1.contiguous(), b1, w2.contiguous(), b2, resid, _ACT[act], *_drop_args(activation_dropout_p), *_drop_args(dropout_p))


# ------------------------------------------------------------------------------------------------
# LayerNorm with fused residual add
# ------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        x2 = _flat2d(x)
        r2 = _flat2d(res) if res is not None else None
        y, s, mean, rstd = K.layernorm_fwd(x2, r2, gamma, beta, eps, want_sum=True)
        src = s if res is not None else x2
        ctx.save_for_backward(src, gamma, mean, rstd)
        ctx.has_res = res is not None
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)  # an unused second output must not cost a zero-filled [rows, cols] gradient
        if res is not None:
            return y.view(x.shape), s.view(x.shape)
        return y.view(x.shape), None

    @staticmethod
    def backward(ctx, dy, ds):
        src, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the sum output was consumed
            return ds, (ds if ctx.has_res else None), None, None, None
        dy2 = _flat2d(dy)
        dres = _flat2d(ds) if ds is not None else None
        dx, dg, db, tiles = K.layernorm_bwd(dy2, src, gamma, mean, rstd, dres, grad_dtype=gamma.dtype, want_tiles=True, defer=_may_defer(gamma))
        dx = _with_tiles(dx.view(ctx.shape), tiles)
        return dx, (dx if ctx.has_res else None), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNormFn.apply(x, None, gamma, beta, eps)[0]


def add_layer_norm(x, res, gamma, beta, eps=1e-5):
    """s = x + res;  returns (LN(s), s) from one kernel."""
    return _LayerNormFn.apply(x, res, gamma, beta, eps)


class _LayerNormPassFn(torch.autograd.Function):
    """y = LN(x) and x handed back as a second output: a pre-norm block uses the pass-through as its residual, so the
    residual-branch gradient and the LN gradient arrive in ONE backward call and are summed inside cst_layernorm_bwd
    (no separate autograd accumulation kernel per block)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = _flat2d(x)
        y, _, mean, rstd = K.layernorm_fwd(x2, None, gamma, beta, eps)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shape = x.shape
        ctx.set_materialize_grads(False)
        return y.view(x.shape), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxp):
        x2, gamma, mean, rstd = ctx.saved_tensors
        dres = _flat2d(dxp) if dxp is not None else None
        if dy is None:
            return (dxp, None, None, None)
        dx, dg, db = K.layernorm_bwd(_flat2d(dy), x2, gamma, mean, rstd, dres, grad_dtype=gamma.dtype, defer=_may_defer(gamma))
        return dx.view(ctx.shape), dg, db, None


def layer_norm_residual(x, gamma, beta, eps=1e-5):
    """Returns (LN(x), x') where x' aliases x and must be used for the block's residual connection."""
    return _LayerNormPassFn.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------
# fused attention
# ------------------------------------------------------------------------------------------------
_MASK_CACHE = []  # [(source mask (kept alive), version, uint8 mask, kv_len)] — one padding mask serves every layer of a pass


_PAD_TILES = []  # [(uint8 mask (kept alive), bool [ceil(B*T/64)]: every row of the 64-row tile of the flattened [B*T] rows is padding)]


def _pad_tiles(u8):
    for src, pt in _PAD_TILES:
        if src is u8:
            return pt
    flat = u8.reshape(-1)
    pad = (-flat.numel()) % 64
    if pad:
        flat = torch.cat([flat, torch.ones(pad, dtype=flat.dtype, device=flat.device)])
    pt = flat.view(-1, 64).amin(dim=1) > 0
    _PAD_TILES.append((u8, pt))
    if len(_PAD_TILES) > 8:
        _PAD_TILES.pop(0)
    return pt


def _mask_and_len(key_padding_mask):
    """(uint8 [B,Tk] mask, int32 [B] kv_len) of a key padding mask; kv_len[b] = 1 + index of the last real key, so that the
    kernels skip the all-padding key tiles at the end of every utterance of a length-sorted batch (cst_attn_desc.kv_len)."""
    if key_padding_mask is None:
        return None, None
    if _os.environ.get("CST_ATTN_NO_KVLEN"):  # test hook: walk every key tile (results are bit-identical either way)
        return key_padding_mask.to(torch.uint8).contiguous(), None
    for src, ver, u8, kvl in _MASK_CACHE:
        if src is key_padding_mask and ver == key_padding_mask._version:
            return u8, kvl
    u8 = key_padding_mask.to(torch.uint8).contiguous()
    Tk = u8.shape[1]
    pos = torch.arange(1, Tk + 1, device=u8.device, dtype=torch.int32)
    kvl = (pos * (u8 == 0)).amax(dim=1).to(torch.int32).contiguous()
    _MASK_CACHE.append((key_padding_mask, key_padding_mask._version, u8, kvl))
    if len(_MASK_CACHE) > 8:
        _MASK_CACHE.pop(0)
    return u8, kvl


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kpm, kvl, H, causal, scale, layout_q, layout_kv, drop_p, drop_key):
        D = q.shape[-1] // H
        o, lse = K.attn_fwd(q, k, v, H, D, kpm, causal, scale, layout_q, layout_kv, drop_p, drop_key, kvl)
        ctx.save_for_backward(q, k, v, o, lse, kpm, kvl)
        ctx.cfg = (H, D, causal, scale, layout_q, layout_kv, drop_p, drop_key)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, kpm, kvl = ctx.saved_tensors
        H, D, causal, scale, lq, lkv, drop_p, drop_key = ctx.cfg
        if do.stride() != o.stride():
            tmp = torch.empty_like(o)
            tmp.copy_(do)
            do = tmp
        dq, dk, dv = K.attn_bwd(do, q, k, v, o, lse, H, D, kpm, causal, scale, lq, lkv, drop_p, drop_key, kvl)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


class _AttnKVFn(torch.autograd.Function):
    """Attention whose keys and values are the two channel halves of ONE projection kv [B, Tk, 2C] (cross-attention / memory
    attention: k_proj and v_proj read the same rows, so they run as one [2C, C] GEMM).  The backward writes dk | dv into one buffer:
    the projection's dX and dW are single GEMMs and the key/value rows receive a single gradient (no autograd sum of two)."""

    @staticmethod
    def forward(ctx, q, kv, kpm, kvl, H, scale, drop_p, drop_key):
        C = kv.shape[-1] // 2
        D = C // H
        k, v = kv[..., :C], kv[..., C:]
        o, lse = K.attn_fwd(q, k, v, H, D, kpm, False, scale, "bt", "bt", drop_p, drop_key, kvl)
        ctx.save_for_backward(q, kv, o, lse, kpm, kvl)
        ctx.cfg = (H, D, C, scale, drop_p, drop_key)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse, kpm, kvl = ctx.saved_tensors
        H, D, C, scale, drop_p, drop_key = ctx.cfg
        if do.stride() != o.stride():
            tmp = torch.empty_like(o)
            tmp.copy_(do)
            do = tmp
        k, v = kv[..., :C], kv[..., C:]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        delta = torch.empty_like(lse)
        d = K.attn_desc(q, k, v, o, lse, H, D, kpm, False, scale, "bt", "bt", drop_p, drop_key, kvl)
        K.attn_bwd_fill(d, do, dq, dkv[..., :C], dkv[..., C:], delta, D, "bt", "bt")
        K.attn_bwd_desc(d)
        return dq, dkv, None, None, None, None, None, None


def attention_kv(q, kv, num_heads, key_padding_mask=None, scale=None, dropout_p=0.0):
    """q [B,Tq,C], kv [B,Tk,2C] (k | v along channels, contiguous) -> [B,Tq,C]; non-causal."""
    C = q.shape[-1]
    if scale is None:
        scale = (C // num_heads) ** -0.5
    key_padding_mask, kv_len = _mask_and_len(key_padding_mask)
    assert q.stride(-1) == 1 and kv.is_contiguous() and kv.shape[-1] == 2 * C
    return _AttnKVFn.apply(q, kv, key_padding_mask, kv_len, num_heads, float(scale), *_drop_args(dropout_p))


def _drop_args(dropout_p):
    """(p, key) of one dropout site: a fresh key of the process-wide stream when p > 0."""
    if dropout_p and dropout_p > 0.0:
        from . import rng
        return float(dropout_p), rng.next_key()
    return 0.0, 0


def attention(q, k, v, num_heads, key_padding_mask=None, causal=False, scale=None, layout_q="bt", layout_kv="bt", dropout_p=0.0):
    """q [B,Tq,C] / k,v [B,Tk,C] (layout "bt") or time-major ("tb"); channels of head h at [h*D,(h+1)*D).
    key_padding_mask: bool/uint8 [B,Tk], True = pad.  dropout_p > 0: attention-probability dropout inside the kernels."""
    if scale is None:
        scale = (q.shape[-1] // num_heads) ** -0.5
    key_padding_mask, kv_len = _mask_and_len(key_padding_mask)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
    return _AttnFn.apply(q, k, v, key_padding_mask, kv_len, num_heads, bool(causal), float(scale), layout_q, layout_kv,
                         *_drop_args(dropout_p))


class _AttnPackedFn(torch.autograd.Function):
    """Self-attention on a packed projection qkv [B, T, 3C] (q | k | v along channels).  The backward writes dq/dk/dv
    into ONE [B, T, 3C] buffer, so the projection's dX / dW are single GEMMs and x receives a single gradient."""

    @staticmethod
    def forward(ctx, qkv, kpm, kvl, H, causal, scale, drop_p, drop_key, pad_tiles=None, seq=None):
        ctx.pad_tiles = pad_tiles
        C = qkv.shape[-1] // 3
        D = C // H
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        B, T = qkv.shape[0], qkv.shape[1]
        o = torch.empty(B, T, C, dtype=qkv.dtype, device=qkv.device)
        # packed sequences: lse / delta are [sequences, H, longest sequence]
        lse = torch.empty((seq[0].numel() - 1) if seq is not None else B, H, int(seq[1]) if seq is not None else T, dtype=torch.float32, device=qkv.device)
        d = K.attn_desc(q, k, v, o, lse, H, D, kpm, causal, scale, "bt", "bt", drop_p, drop_key, kvl, seq)
        K.attn_fwd_desc(d)
        ctx.save_for_backward(qkv, o, lse, kpm, kvl, seq[0] if seq is not None else None)
        ctx.cfg = (H, D, C, causal, scale, drop_p, drop_key, int(seq[1]) if seq is not None else None)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, kpm, kvl, seq_off = ctx.saved_tensors
        H, D, C, causal, scale, drop_p, drop_key, seq_max = ctx.cfg
        seq = (seq_off, seq_max) if seq_off is not None else None
        live = _tiles_of(do, qkv.shape[0] * qkv.shape[1]) if ctx.pad_tiles is not None else None
        if not do.is_contiguous():
            do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        delta = torch.empty_like(lse)
        d = K.attn_desc(q, k, v, o, lse, H, D, kpm, causal, scale, "bt", "bt", drop_p, drop_key, kvl, seq)
        K.attn_bwd_fill(d, do, dq, dk, dv, delta, D, "bt", "bt")
        K.attn_bwd_desc(d)
        if live is not None:
            # row t of dq|dk|dv is exactly zero when dO_t is (dS_t. = P_t. * 0) AND key t is padding (P_.t = 0): a 64-row tile
            # that is dead in dO and all padding is dead here; every other tile is declared live
            stamps = torch.where(ctx.pad_tiles, live[0], torch.full_like(live[0], live[1] if live[1] < 2 ** 31 else live[1] - 2 ** 32))
            dqkv = _with_tiles(dqkv, (stamps, live[1]))
        return dqkv, None, None, None, None, None, None, None, None, None


def attention_packed(qkv, num_heads, key_padding_mask=None, causal=False, scale=None, dropout_p=0.0, seq=None):
    """qkv [B, T, 3C] contiguous -> [B, T, C].
    seq = PackedRows (see pack_rows): qkv is [1, rows, 3C] holding variable-length sequences back to back; sequence b attends to its
    first seq.kv_len[b] rows (its real frames); key_padding_mask must be None."""
    C = qkv.shape[-1] // 3
    if scale is None:
        scale = (C // num_heads) ** -0.5
    assert qkv.is_contiguous()
    if seq is not None:
        assert key_padding_mask is None and qkv.shape[0] == 1 and qkv.shape[1] == seq.rows
        return _AttnPackedFn.apply(qkv, None, seq.kv_len, num_heads, bool(causal), float(scale), *_drop_args(dropout_p), None, (seq.offsets, seq.longest))
    key_padding_mask, kv_len = _mask_and_len(key_padding_mask)
    pad_tiles = _pad_tiles(key_padding_mask) if (key_padding_mask is not None and qkv.requires_grad) else None
    return _AttnPackedFn.apply(qkv, key_padding_mask, kv_len, num_heads, bool(causal), float(scale), *_drop_args(dropout_p), pad_tiles)


# ------------------------------------------------------------------------------------------------
# q | k | v parameters as one packed operand
# ------------------------------------------------------------------------------------------------
class _StackedRowsFn(torch.autograd.Function):
    """Parameters that sit back to back in memory (optim.FlatParamBuffers lays the q, k, v projections of an attention module out
    that way) seen as ONE [n C, ...] tensor: a view, no copy.  backward hands each parameter its row block of the packed gradient
    (views of the one tensor the dW GEMM wrote)."""

    @staticmethod
    def forward(ctx, *ts):
        a = ts[0]
        ctx.rows, ctx.n = a.shape[0], len(ts)
        return torch.as_strided(a, (len(ts) * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())

    @staticmethod
    def backward(ctx, g):
        r = ctx.rows
        return tuple(g[i * r:(i + 1) * r] for i in range(ctx.n))


def stacked_rows(*ts):
    """cat(ts, 0) for two or three equal-shaped parameters (q | k | v, or k | v); free when they are adjacent in memory."""
    a = ts[0]
    n = a.numel() * a.element_size()
    if (not _os.environ.get("CST_NO_QKV_VIEW") and all(t.shape == a.shape and t.is_contiguous() for t in ts)
            and all(ts[i].data_ptr() + n == ts[i + 1].data_ptr() for i in range(len(ts) - 1))
            and a.untyped_storage().data_ptr() == ts[-1].untyped_storage().data_ptr()):
        K.STATS["qkv_view"] = K.STATS.get("qkv_view", 0) + 1
        out = _StackedRowsFn.apply(*ts)
        out._cst_parts = tuple(t.detach() for t in ts)  # version counters of all of them (WEIGHT_TRANSPOSES, _wversion)
        out._cst_parts_params = tuple(ts)               # ... and the parameters themselves (optim.grad_slot)
        return out
    return torch.cat(ts, 0)


# ------------------------------------------------------------------------------------------------
# packed (padding-free) row sets — include/cst.h: cst_rows_pack / cst_rows_unpack
# ------------------------------------------------------------------------------------------------
class PackedRows:
    """Where the sequences of a packed [rows, C] tensor live: offsets int32 [B+1] (device), kv_len int32 [B] (device: real frames per
    sequence = its keys), rows (total), longest (longest sequence), B, T (the padded shape it came from)."""

    def __init__(self, offsets, kv_len, rows, longest, B, T, host_lens=None):
        self.offsets, self.kv_len, self.rows, self.longest, self.B, self.T = offsets, kv_len, rows, longest, B, T
        self.host_lens = host_lens  # real frames per sequence as host integers (read with the plan's one host transfer)
        self.kept_host = None       # rows each sequence keeps (offsets[b + 1] - offsets[b]) as host integers, where the planner knows them


def plan_packed_rows(padding_mask, margin):
    """padding_mask bool [B, T] (True = padding, a suffix per row).  Sequence b keeps n_b = min(T, len_b + margin + 1) rows: its real
    frames, the `margin` padding frames that are still influenced by real ones, and ONE more that stands for every identical
    padding frame behind it.  One host read of B integers (the plan sizes the packed allocations)."""
    B, T = padding_mask.shape
    lens = (~padding_mask).sum(dim=1).to(torch.int32)
    n = torch.clamp(lens + (margin + 1), max=T)
    off = torch.zeros(B + 1, dtype=torch.int32, device=padding_mask.device)
    off[1:] = torch.cumsum(n, 0)
    host = torch.cat((off[-1:], n.max().view(1), lens)).tolist()  # the step's packing plan: total rows, longest sequence, lengths
    plan = PackedRows(off.contiguous(), lens.contiguous(), int(host[0]), int(host[1]), B, T, [int(v) for v in host[2:]])
    plan.kept_host = [min(T, v + margin + 1) for v in plan.host_lens]
    return plan


def plan_from_lengths(lens_dev, host_lens, T):
    """The plan of a later stage whose sequence lengths follow from an earlier plan's by host arithmetic (the subsampled frames
    behind wav2vec2): device offsets by device ops, the allocation sizes from the host copies — no further host read.  Sequence b
    keeps exactly its lens[b] real rows."""
    B = lens_dev.numel()
    n = torch.clamp(lens_dev.to(torch.int32), max=T)
    off = torch.zeros(B + 1, dtype=torch.int32, device=lens_dev.device)
    off[1:] = torch.cumsum(n, 0)
    hl = [min(int(v), T) for v in host_lens]
    return PackedRows(off.contiguous(), n.contiguous(), sum(hl), max(hl), B, T, hl)


class _PackRowsFn(torch.autograd.Function):
    """[B, T, C] -> [1, rows, C]: the rows each sequence keeps; backward = unpack with zeros behind the kept rows."""

    @staticmethod
    def forward(ctx, x, seq):
        ctx.seq = seq
        return K.rows_pack(x if x.is_contiguous() else x.contiguous(), seq.offsets, seq.rows, tail_sum=False).unsqueeze(0)

    @staticmethod
    def backward(ctx, dy):
        s = ctx.seq
        dy2 = dy.reshape(s.rows, dy.shape[-1])
        return K.rows_unpack(dy2 if dy2.is_contiguous() else dy2.contiguous(), s.offsets, s.B, s.T, tail_broadcast=False), None


class _UnpackRowsFn(torch.autograd.Function):
    """[1, rows, C] -> [B, T, C]: rows behind a sequence's kept ones repeat its last kept row (they are identical padding frames);
    backward = pack with the gradients of those copies summed into that row, in index order."""

    @staticmethod
    def forward(ctx, y, seq, broadcast):
        ctx.seq, ctx.broadcast = seq, broadcast
        y2 = y.reshape(seq.rows, y.shape[-1])
        return K.rows_unpack(y2 if y2.is_contiguous() else y2.contiguous(), seq.offsets, seq.B, seq.T, tail_broadcast=broadcast)

    @staticmethod
    def backward(ctx, dx):
        s = ctx.seq
        return K.rows_pack(dx if dx.is_contiguous() else dx.contiguous(), s.offsets, s.rows, tail_sum=ctx.broadcast).unsqueeze(0), None, None


def pack_rows(x, seq):
    return _PackRowsFn.apply(x, seq)


def unpack_rows(y, seq, broadcast=True):
    """broadcast=False: the rows behind a sequence's kept ones come back as zeros (nobody reads them) and their gradient is dropped."""
    return _UnpackRowsFn.apply(y, seq, bool(broadcast))


# ------------------------------------------------------------------------------------------------
# wav2vec2 conv layer 0 + GroupNorm + GELU
# ------------------------------------------------------------------------------------------------
class _Conv0Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, w, gamma, beta, stride, eps, write_limit, grad_limit):
        C, _, k = w.shape
        w2 = w.reshape(C, k).contiguous()
        wav = wav.float().contiguous()
        if _os.environ.get("CST_NO_MLEN") or _os.environ.get("CST_GEMM_NO_KLIVE"):
            write_limit = grad_limit = None  # (with the GEMM-side skipping off, every frame of this layer is read)
        y, mean, rstd, gram = K.conv0_fwd(wav, w2, gamma, beta, k, stride, eps, frame_limit=write_limit)
        ctx.save_for_backward(wav, w2, gamma, beta, mean, rstd, gram)
        ctx.k, ctx.stride, ctx.grad_limit = k, stride, grad_limit
        return y

    @staticmethod
    def backward(ctx, dy):
        wav, w2, gamma, beta, mean, rstd, gram = ctx.saved_tensors
        dw, dg, db = K.conv0_bwd(dy.contiguous(), wav, w2, gamma, beta, mean, rstd, gram, ctx.k, ctx.stride, frame_limit=ctx.grad_limit)
        return None, dw.view(w2.shape[0], 1, ctx.k).to(w2.dtype), dg.to(gamma.dtype), db.to(gamma.dtype), None, None, None, None


def conv0_gn_gelu(wav, weight, gn_weight, gn_bias, stride, eps=1e-5, write_limit=None, grad_limit=None):
    """wav [B,S] -> channels-last [B, L, C] = GELU(GroupNorm_C(conv1d(wav, weight[C,1,k], stride))).
    write_limit / grad_limit (int32 [B], cst_conv_row_limits rows [0][1] / [0][0]): frames nobody reads are not written, frames whose
    gradient is exactly zero are not read in backward."""
    return _Conv0Fn.apply(wav, weight, gn_weight, gn_bias, stride, eps, write_limit, grad_limit)


# ------------------------------------------------------------------------------------------------
# channels-last conv1d as implicit GEMM (wav2vec2 conv layers 1.., subsampler)
# ------------------------------------------------------------------------------------------------
def _padded_base(t, L, C):
    """If `t` [B, L, C] is the interior view of a [B, L+2, C] allocation (one spare row before and after every utterance),
    return that allocation as a tensor, else None."""
    if t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == C and t.stride(0) == (L + 2) * C and t.storage_offset() >= C:
        return torch.as_strided(t, (t.shape[0], L + 2, C), t.stride(), t.storage_offset() - C)
    return None


class _Conv1dCLFn(torch.autograd.Function):
    """Channels-last conv1d as implicit GEMMs.
    forward : y[b,t,:] = act(W . x[b, t*s : t*s+k, :])            one GEMM, A rows overlap (lda = s*Cin < K = k*Cin)
    dW      : per-utterance [Cout, k*Cin] partials (both operands mn-major, B rows overlap), summed over the batch
    dX      : pad == 0, k <= 2s (the wav2vec2 CNN): for every residue r of l mod s one GEMM
                 dx[b, s*u + r, :] = sum_i dz[b, u - i, :] . W[:, :, r + s*i]
              whose A rows are again overlapping windows of the (row-padded) dz and whose C rows are written with ldc = s*Cin
              straight into dx — no [B, Lout, k*Cin] column buffer and no col2im pass; GELU'(prev_z) is the epilogue.
              otherwise (padded subsampler convs): column-gradient GEMM + cst_col2im1d.
    Layers whose incoming gradient is already d/dz (grad_is_dz) keep y / z / dx in a [B, L+2, C] allocation (interior views)
    so the zero rows the dX windows need at utterance boundaries exist without copying."""

    @staticmethod
    def forward(ctx, x, w_cl, bias, k, stride, pad, act, prev_z, grad_is_dz, nz_out=None, nz_in=None, unread_ok=False):
        if _os.environ.get("CST_NO_MLEN"):  # A/B switch of the tests: compute the frames past each utterance's end as well
            nz_in = None
            fwd_len = None
        else:
            # (with a bias the rows of a skipped tile hold act(bias), not 0: only where the caller states that nobody reads them)
            fwd_len = nz_out if (bias is None or unread_ok) else None
        ctx.nz_out = nz_out  # int32 [B]: the incoming gradient's rows t >= nz_out[b] are exactly zero (frames past the utterance's end)
        ctx.nz_in = nz_in    # int32 [stride, B]: live rows of each residue class of dx (cst_conv_row_limits); dx is zero behind them
        B, Lin, Cin = x.shape
        Cout = w_cl.shape[0]
        if pad:
            xp = torch.empty(B, Lin + 2 * pad, Cin, dtype=x.dtype, device=x.device)  # (only the pad rows are zero-filled, not the whole buffer)
            xp[:, :pad].zero_()
            xp[:, pad + Lin:].zero_()
            xp[:, pad:pad + Lin] = x
        elif x.stride(2) == 1 and x.stride(1) == Cin:
            xp = x  # contiguous or an interior view of a row-padded allocation: only the batch stride differs
        else:
            xp = x.contiguous()
        Lp = Lin + 2 * pad
        Lout = (Lp - k) // stride + 1
        padded_out = bool(grad_is_dz) and act != L.ACT_NONE
        rows = Lout + 2 if padded_out else Lout
        y_full = torch.empty(B, rows, Cout, dtype=x.dtype, device=x.device)
        z_full = torch.empty_like(y_full) if act != L.ACT_NONE else None
        coff = Cout if padded_out else 0
        K.gemm(xp, w_cl, y_full, Lout, Cout, k * Cin, a_kmajor=1, b_kmajor=1, lda=stride * Cin, ldb=k * Cin, ldc=Cout, bias=bias,
               act=act, aux_out=z_full, ld_aux_out=Cout, batch0=B, sa=(xp.stride(0), 0), sc=(rows * Cout, 0), split_k=1,
               a_off=0, c_off=coff, m_len=fwd_len)  # frames past an utterance's end: unread, left at act(0) = 0
        y = y_full[:, 1:1 + Lout] if padded_out else y_full
        z = (z_full[:, 1:1 + Lout] if padded_out else z_full) if z_full is not None else None
        ctx.save_for_backward(xp, w_cl, z, prev_z)
        ctx.cfg = (B, Lin, Cin, Cout, k, stride, pad, Lout, act, bias is not None, grad_is_dz)
        ctx.set_materialize_grads(False)  # the pre-activation output never carries a gradient
        if act != L.ACT_NONE:
            ctx.mark_non_differentiable(z)
            return y, z
        return y, None

    @staticmethod
    def backward(ctx, dy, _dz_unused):
        xp, w_cl, z, prev_z = ctx.saved_tensors
        B, Lin, Cin, Cout, k, stride, pad, Lout, act, has_bias, grad_is_dz = ctx.cfg
        fast = pad == 0 and k <= stride + 1  # every dX window then stays inside the row-padded dz
        # ---- dz, in a row-padded allocation when the windowed dX GEMMs will read it ----
        dzp = _padded_base(dy, Lout, Cout) if (act == L.ACT_NONE or grad_is_dz) else None
        if dzp is None:
            dyc = dy if dy.is_contiguous() else dy.contiguous()
            dzc = K.act_bwd(dyc, z if z.is_contiguous() else z.contiguous(), act) if (act != L.ACT_NONE and not grad_is_dz) else dyc
            if fast:
                dzp = torch.empty(B, Lout + 2, Cout, dtype=dy.dtype, device=dy.device)
                dzp[:, 0].zero_()
                dzp[:, Lout + 1].zero_()
                dzp[:, 1:1 + Lout] = dzc
            dz_rows, dz_bs, dz_off = dzc, Lout * Cout, 0
        else:
            dz_rows, dz_bs, dz_off = dzp, (Lout + 2) * Cout, Cout
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if fast:
                dx_padded = prev_z is not None and _padded_base(prev_z, Lin, Cin) is not None
                rows = Lin + 2 if dx_padded else Lin
                dx_full = torch.empty(B, rows, Cin, dtype=dy.dtype, device=dy.device)
                if dx_padded:
                    dx_full[:, 0].zero_()
                    dx_full[:, Lin + 1].zero_()
                pz = prev_z if prev_z is None or dx_padded else (prev_z if prev_z.is_contiguous() else prev_z.contiguous())
                pz_base = _padded_base(prev_z, Lin, Cin) if dx_padded else pz
                w3 = w_cl.view(Cout, k, Cin)
                for r in range(stride):
                    taps = list(range(r, k, stride))
                    n = len(taps)
                    U = (Lin - r + stride - 1) // stride
                    if n == 0 or U <= 0:
                        continue
                    # B_r[ci][q*Cout + co] = W[co][ci][r + s*(n-1-q)]   (window position q <-> dz[u - (n-1) + q])
                    # (taps r, r + s, ... in reverse order by a strided slice and a flip: indexing with a Python list would build the
                    #  index tensor on the host — a pageable copy, i.e. a stream synchronisation in the middle of the backward pass)
                    wr = w3[:, r::stride, :].flip(1).permute(2, 1, 0).reshape(Cin, n * Cout).contiguous()
                    # rows u with stride * u + r >= nz_in[b] read only dz rows that are exactly zero: their K loop is skipped
                    ml = ctx.nz_in[r] if ctx.nz_in is not None else None
                    K.gemm(dzp, wr, dx_full, U, Cin, n * Cout, a_kmajor=1, b_kmajor=1, lda=Cout, ldb=n * Cout, ldc=stride * Cin,
                           batch0=B, sa=((Lout + 2) * Cout, 0), sc=(rows * Cin, 0), a_off=(2 - n) * Cout,
                           c_off=(Cin if dx_padded else 0) + r * Cin, dact=L.ACT_GELU if prev_z is not None else L.ACT_NONE,
                           aux_in=pz_base, ld_aux_in=stride * Cin, split_k=1, m_len=ml)
                dx = dx_full[:, 1:1 + Lin] if dx_padded else dx_full
            else:
                dz2 = dz_rows.reshape(B * Lout, Cout) if dzp is None else dzp[:, 1:1 + Lout].reshape(B * Lout, Cout)
                dcol = torch.empty(B * Lout, k * Cin, dtype=dy.dtype, device=dy.device)
                K.gemm(dz2, w_cl, dcol, B * Lout, k * Cin, Cout, a_kmajor=1, b_kmajor=0, lda=Cout, ldb=k * Cin, ldc=k * Cin, split_k=1)
                pzc = prev_z if prev_z is None or prev_z.is_contiguous() else prev_z.contiguous()
                dx = K.col2im1d(dcol, pzc, B, Lin, Lout, Cin, k, stride, pad, L.ACT_GELU if prev_z is not None else 0)
        if ctx.needs_input_grad[1]:
            part = torch.empty(B, Cout, k * Cin, dtype=torch.float32, device=dy.device)
            # one 256 x 256 workgroup per CU: B x tiles workgroups should fill whole rounds of 256 (B = 32, k = 3: 384 = 1.5
            # rounds -> a 2-way split of the frame axis gives 768 = 3 rounds of half the length)
            wgs = B * ((Cout + 255) // 256) * ((k * Cin + 255) // 256)
            rem = wgs % 256
            sk = 2 if (wgs >= 256 and 0 < rem < 192 and Lout >= 2048) else 1
            # the sum over the utterances (and over the K slices) is ONE fixed-order pass of cst_reduce_multi over the fp32 partials,
            # written in the parameter dtype (before: split-K reduce launch + torch sum over the batch + dtype conversion)
            dw = torch.empty(Cout, k * Cin, dtype=w_cl.dtype, device=dy.device)
            K.gemm(dz_rows, xp, part, Cout, k * Cin, Lout, a_kmajor=0, b_kmajor=0, lda=Cout, ldb=stride * Cin, ldc=k * Cin, batch0=B,
                   sa=(dz_bs, 0), sb=(xp.stride(0), 0), sc=(Cout * k * Cin, 0), a_off=dz_off, split_k=sk, k_len=ctx.nz_out, reduce_to=dw)
            # (not deferred: autograd re-lays this gradient out — permute / reshape back to [Cout, Cin, k] — as soon as it is returned)
        if has_bias and ctx.needs_input_grad[2]:
            db = K.colsum(dz_rows.reshape(B * Lout, Cout) if dzp is None else dzp[:, 1:1 + Lout].reshape(B * Lout, Cout), w_cl.dtype)
        return dx, dw, db, None, None, None, None, None, None, None, None, None


def conv1d_cl(x, weight, bias, stride, pad=0, act=None, prev_z=None, grad_is_dz=False, nz_out=None, nz_in=None, unread_ok=False):
    """Channels-last conv1d.  x [B,Lin,Cin]; weight [Cout,Cin,k] (torch layout).  Returns (y, z) where z is
    the pre-activation (None without activation).  If `prev_z` (pre-GELU tensor that produced x = GELU(prev_z))
    is given, the returned input-gradient is already multiplied by GELU'(prev_z), i.e. it is d/d prev_z; the
    producing layer must then be built with grad_is_dz=True so it does not apply GELU' a second time
    (see the wav2vec2 feature extractor: one fused col2im+GELU' pass per layer instead of two)."""
    Cout, Cin, k = weight.shape
    w_cl = weight.permute(0, 2, 1).reshape(Cout, k * Cin)
    if not w_cl.is_contiguous():
        w_cl = w_cl.contiguous()
    return _Conv1dCLFn.apply(x, w_cl, bias, k, stride, pad, _ACT[act], prev_z, bool(grad_is_dz), nz_out, nz_in, bool(unread_ok))


# ------------------------------------------------------------------------------------------------
# grouped positional conv of wav2vec2 (k taps, `groups` groups, pad k//2, SamePad, GELU, + residual)
# ------------------------------------------------------------------------------------------------
# K slices of the positional convolution's weight-gradient GEMM (16 groups x 48 column tiles = 768 workgroups of the 64 x 128 configuration,
# 1.5 rounds of the 512 that fit the chip): 2 or 3 slices measured the same as 1 inside the update (64.5-65.8 ms per update in all
# three settings on one box) — a switch for the tools, default 1
_POSCONV_DW_SPLIT = int(_os.environ.get("CST_POSCONV_DW_SPLIT", 1))


class _PosConvFn(torch.autograd.Function):
    """Grouped conv as a batched implicit GEMM.  The input is re-staged GROUP-MAJOR ([B, G, T+k, C/G], one 74 MB copy at
    B=32) so that for one (utterance, group) the im2col row of frame t is the contiguous window starting at frame t:
    a plain k-major operand with lda = C/G < K (overlapping rows) — no segmented addressing in the inner loop."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, lens=None, grad_rows=None, grad_rows_host=None):
        """x [B,T,C]; weight [C, C/g, k] -> x + GELU(conv(x) + bias)   (one fused GEMM launch).
        lens (int32 [B], device): frames of x from lens[b] on are ZERO (wav2vec2.py:820-821 zeroes the padding in front of this
        convolution).  The window of output frame t >= lens[b] + k // 2 then holds nothing but zeros, so its row of the implicit
        GEMM's A operand is zero: cst_gemm_desc.m_len = lens + k // 2 lets the tiles behind it skip their K loop, and their epilogue
        on zero accumulators — GELU(bias) + x — IS the value of those frames: every output frame keeps its bits.
        grad_rows (int32 [B], device): the gradient of the output is exactly zero from frame grad_rows[b] on (the rows a packing plan
        keeps: pack_rows' backward writes zeros behind them); the windows of dx frames >= grad_rows[b] + k - 1 - k // 2 are zero."""
        B, T, C = x.shape
        k = weight.shape[2]
        cg = C // groups
        padl = k // 2
        Tp = T + k - 1 + (1 if k % 2 == 0 else 0)  # even k: torch pads k//2 both sides, SamePad drops the last output
        # [G, B, Tp, C/G]: group-major OUTSIDE the batch, so that one group's frames of all utterances are one [B * Tp, C/G] matrix
        # (the weight-gradient GEMM below runs over it with K = every frame of the batch)
        xg = torch.empty(groups, B, Tp, cg, dtype=x.dtype, device=x.device)  # (zero rows around the frames: only those are filled)
        xg[:, :, :padl].zero_()
        xg[:, :, padl + T:].zero_()
        xg[:, :, padl:padl + T] = x.view(B, T, groups, cg).permute(2, 0, 1, 3)
        wg = weight.view(groups, cg, cg, k).permute(0, 1, 3, 2).contiguous()  # [g][co][j][ci]
        y = torch.empty(B, T, C, dtype=x.dtype, device=x.device)
        z = torch.empty_like(y)
        xc = x if x.is_contiguous() else x.contiguous()
        ml = None if lens is None else torch.clamp(lens.to(torch.int32) + padl, max=T).contiguous()
        K.gemm(xg, wg, y, T, cg, k * cg, a_kmajor=1, b_kmajor=1, lda=cg, ldb=k * cg, ldc=C, batch0=B, batch1=groups,
               sa=(Tp * cg, B * Tp * cg), sb=(0, cg * k * cg), sc=(T * C, cg), bias=bias, sbias=(0, cg), act=L.ACT_GELU,
               aux_out=z, ld_aux_out=C, resid=xc, ld_resid=C, split_k=1, m_len=ml)
        ctx.save_for_backward(xg, weight, z)
        ctx.cfg = (B, T, C, k, groups, cg, padl, Tp)
        ctx.grad_rows, ctx.grad_rows_host = grad_rows, grad_rows_host
        return y

    @staticmethod
    def backward(ctx, dy):
        xg, weight, z = ctx.saved_tensors
        B, T, C, k, groups, cg, padl, Tp = ctx.cfg
        dy = dy.contiguous()
        dz = K.act_bwd(dy, z, L.ACT_GELU)
        dx = dw = db = None
        # dzg[g, b, lp + t] = dz[b, t, group g], zero rows around it, in the frame stride Tp of xg: the operand of BOTH gradient GEMMs
        lp = k - 1 - padl
        dzg = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dzg = torch.empty(groups, B, Tp, cg, dtype=dy.dtype, device=dy.device)
            dzg[:, :, :lp].zero_()
            dzg[:, :, lp + T:].zero_()
            dzg[:, :, lp:lp + T] = dz.view(B, T, groups, cg).permute(2, 0, 1, 3)
        if ctx.needs_input_grad[0]:
            # dx[m] = dy[m] + sum_j dz[m + padl - j] w_j  = dy[m] + sum_j' dzp[m + j'] wflip[j'],  dzp[(k-1-padl) + t] = dz[t]
            wflip = weight.view(groups, cg, cg, k).flip(3).permute(0, 2, 3, 1).contiguous()  # [g][ci][j'][co]
            dx = torch.empty(B, T, C, dtype=dy.dtype, device=dy.device)
            gr = ctx.grad_rows
            ml = None if gr is None else torch.clamp(gr.to(torch.int32) + lp, max=T).contiguous()  # (dead tiles: dx = 0 + dy = dy, exactly)
            K.gemm(dzg, wflip, dx, T, cg, k * cg, a_kmajor=1, b_kmajor=1, lda=cg, ldb=k * cg, ldc=C, batch0=B, batch1=groups,
                   sa=(Tp * cg, B * Tp * cg), sb=(0, cg * k * cg), sc=(T * C, cg), resid=dy, ld_resid=C, split_k=1, m_len=ml)
        if ctx.needs_input_grad[1]:
            # dw[g][co][(j,ci)] = sum_{b,t} dz[b, t, g, co] xg[g, b, t + j, ci]: ONE reduction over the frames of the whole batch per
            # group.  Row kappa = b Tp + t of A is dzg's row kappa + lp (zero for t >= T: the windows that would run from one
            # utterance into the next contribute nothing); row kappa of B is the k-frame window of xg starting at frame kappa
            # (overlapping mn-major rows, ldb = cg).  (Before: one [cg, k cg] product per (utterance, group) — 600 MB of fp32
            # partial sums written and read back by a sum over the batch.)
            Kr = B * Tp - (k - 1)
            dwg = torch.empty(groups, cg, k * cg, dtype=torch.float32, device=dy.device)
            if ctx.needs_input_grad[2] and not _os.environ.get("CST_NO_GEMM_COLSUM"):
                # the bias gradient = column sums of this GEMM's A operand (every dz row is inside its K range, the rest are zeros):
                # [groups][cg] = channel order (cst_gemm_desc.colsum)
                db = torch.empty(C, dtype=dy.dtype, device=dy.device)
            live = None
            if ctx.grad_rows is not None and not _os.environ.get("CST_GEMM_NO_KLIVE"):
                # dz is zero from frame grad_rows[b] of utterance b on: the 64-row K blocks of A that lie in those stretches (a third
                # of them at the bench's lengths) are skipped (cst_gemm_desc.k_live: a block is live iff its stamp equals the epoch, 1).
                # Row kappa = b Tp + t; a block may run from one utterance's tail into the head of the next.
                if ctx.grad_rows_host is not None:
                    # the plan's row counts are host integers already (its one host read): the stamps are a few hundred numbers of
                    # numpy arithmetic and ONE asynchronous copy from a pinned buffer instead of a dozen small launches
                    import numpy as np
                    n = np.asarray(ctx.grad_rows_host, dtype=np.int64)
                    ks = np.arange((Kr + 63) // 64, dtype=np.int64) * 64
                    b0 = ks // Tp
                    t0 = ks - b0 * Tp
                    nxt = np.minimum(b0 + 1, B - 1)
                    st = ((t0 < n[b0]) | ((t0 + 64 > Tp) & (b0 + 1 < B) & (n[nxt] > 0))).astype(np.int32)
                    # (a fresh pinned block per call: torch's host allocator does not hand it out again before the copy has run, which
                    #  a buffer kept here could not promise across the micro-batches of an accumulated update)
                    stamps = torch.from_numpy(st).pin_memory().to(dy.device, non_blocking=True)
                else:
                    n = ctx.grad_rows.to(torch.int64)
                    ks = torch.arange((Kr + 63) // 64, device=dy.device, dtype=torch.int64) * 64
                    b0 = torch.div(ks, Tp, rounding_mode="floor")
                    t0 = ks - b0 * Tp
                    nxt = torch.clamp(b0 + 1, max=B - 1)
                    stamps = ((t0 < n[b0]) | ((t0 + 64 > Tp) & (b0 + 1 < B) & (n[nxt] > 0))).to(torch.int32)
                live = (stamps, 1)
            K.gemm(dzg, xg, dwg, cg, k * cg, Kr, a_kmajor=0, b_kmajor=0, lda=cg, ldb=cg, ldc=k * cg, batch0=1, batch1=groups,
                   sa=(0, B * Tp * cg), sb=(0, B * Tp * cg), sc=(0, cg * k * cg), a_off=lp * cg, split_k=_POSCONV_DW_SPLIT, colsum=db, k_live=live)
            dw = dwg.view(groups, cg, k, cg).permute(0, 1, 3, 2).reshape(C, cg, k).to(weight.dtype)
            if db is not None:
                db = db.to(weight.dtype)
        if ctx.needs_input_grad[2] and db is None:
            db = K.colsum(dz.view(B * T, C), weight.dtype)
        return dx, dw, db, None, None, None, None


class _WeightNormFn(torch.autograd.Function):
    """nn.utils.weight_norm(conv, dim=2) (wav2vec2.py:773-779): w = v * g / ||v||, norms over all dims but the last."""

    @staticmethod
    def forward(ctx, v, g):
        vc, gc = v.contiguous(), g.reshape(-1).contiguous()
        w, norm = K.weight_norm_fwd(vc, gc)
        ctx.save_for_backward(vc, gc, norm)
        ctx.gshape = g.shape
        return w

    @staticmethod
    def backward(ctx, dw):
        vc, gc, norm = ctx.saved_tensors
        dv, dg = K.weight_norm_bwd(vc, gc, dw.contiguous(), norm)
        return dv, dg.view(ctx.gshape)


def weight_norm_last_dim(v, g):
    return _WeightNormFn.apply(v, g)


def pos_conv_gelu_residual(x, weight, bias, groups, lens=None, grad_rows=None, grad_rows_host=None):
    return _PosConvFn.apply(x, weight, bias, groups, lens, grad_rows, grad_rows_host)


# ------------------------------------------------------------------------------------------------
class _GluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z2 = _flat2d(z)
        ctx.save_for_backward(z2)
        ctx.shape = z.shape
        return K.glu_fwd(z2).view(*z.shape[:-1], z.shape[-1] // 2)

    @staticmethod
    def backward(ctx, dy):
        (z2,) = ctx.saved_tensors
        return K.glu_bwd(_flat2d(dy), z2).view(ctx.shape)


def glu(z):
    """F.glu over the last (channel) dimension of a channels-last tensor."""
    return _GluFn.apply(z)


class _MaskRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        m = mask.reshape(-1).to(torch.uint8).contiguous()
        ctx.save_for_backward(m)
        ctx.shape = x.shape
        return K.mask_rows(_flat2d(x), m).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        return K.mask_rows(_flat2d(dy), m).view(ctx.shape), None


def mask_rows(x, mask):
    """x[mask] = 0 over the leading dims (wav2vec2.py:820-821)."""
    return _MaskRowsFn.apply(x, mask)


class _DropoutFn(torch.autograd.Function):
    """FairseqDropout (modules/fairseq_dropout.py): y = x * keep / (1 - p); backward re-evaluates the same counter-based mask."""

    @staticmethod
    def forward(ctx, x, p, key):
        ctx.p, ctx.key = p, key
        xc = x if x.is_contiguous() else x.contiguous()
        return K.dropout(xc, p, key).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dc = dy if dy.is_contiguous() else dy.contiguous()
        return K.dropout(dc, ctx.p, ctx.key).view(dy.shape), None, None


def dropout(x, p, key=None):
    """Training-mode dropout on the HIP path; `key` defaults to the next site key of the process-wide stream (rng.py)."""
    if p <= 0.0:
        return x
    from . import rng
    return _DropoutFn.apply(x, float(p), rng.next_key() if key is None else key)


class _EmbedPosFn(torch.autograd.Function):
    """dropout(scale * (E[tokens] | x) + sinusoidal_positions): one kernel forward (cst_embed_pos_fwd); backward = the
    deterministic table gradient cst_embed_bwd (embedding) or cst_dropout_scale (dense features).  The positional table
    is data (no gradient), make_positions is evaluated inside the kernel."""

    @staticmethod
    def forward(ctx, tokens, pad_mask, embed, x, pos_table, scale, pad_idx, p, key):
        out = K.embed_pos_fwd(tokens, pad_mask, embed, x, pos_table, scale, pad_idx, p, key)
        ctx.save_for_backward(tokens)
        ctx.cfg = (scale, pad_idx, p, key, embed.shape[0] if embed is not None else 0, embed.dtype if embed is not None else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        (tokens,) = ctx.saved_tensors
        scale, pad_idx, p, key, V, gdt = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if V:
            dE = K.embed_bwd(dy, tokens, V, scale, pad_idx, p, key, gdt) if ctx.needs_input_grad[2] else None
            return None, None, dE, None, None, None, None, None, None
        dx = K.dropout_scale(dy, scale, p, key) if ctx.needs_input_grad[3] else None
        return None, None, None, dx, None, None, None, None, None


def embed_positions(tokens=None, pad_mask=None, embed=None, x=None, pos_table=None, scale=1.0, pad_idx=1, dropout_p=0.0):
    """[B, T, C] = dropout(scale * (embed[tokens] or x) + pos_table[make_positions(pad_mask or tokens, pad_idx)]).
    tokens int64 [B, T]; pad_mask bool/uint8 [B, T] (True = pad; takes precedence as the position source); pos_table fp32
    [>= pad_idx + 1 + T, C] or None."""
    if pad_mask is not None:
        pad_mask = pad_mask.to(torch.uint8).contiguous()
    if tokens is not None:
        tokens = tokens.contiguous()
    if x is not None and not x.is_contiguous():
        x = x.contiguous()
    return _EmbedPosFn.apply(tokens, pad_mask, embed, x, pos_table, float(scale), int(pad_idx), *_drop_args(dropout_p))


class _LsCeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, eps, pad):
        l2 = _flat2d(logits)
        t = target.reshape(-1).contiguous()
        out2, lse = K.ls_ce_fwd(l2, t, eps, pad)
        ctx.save_for_backward(l2, t, lse)
        ctx.cfg = (eps, pad, logits.shape)
        ctx.mark_non_differentiable(out2)
        return out2[0].clone(), out2

    @staticmethod
    def backward(ctx, dloss, _):
        l2, t, lse = ctx.saved_tensors
        eps, pad, shape = ctx.cfg
        g = dloss.reshape(1).float().contiguous()
        return K.ls_ce_bwd(l2, t, lse, g, eps, pad).view(shape), None, None, None


def label_smoothed_nll_loss(logits, target, eps, pad):
    """Returns (loss_sum [differentiable], nll_sum [detached]); fp32 scalars on device."""
    loss, out2 = _LsCeFn.apply(logits, target, float(eps), int(pad))
    return loss, out2[1]


class _ContrastiveFn(torch.autograd.Function):
    """compute_contrastive (criterions/triplet_st_mt_contrastive.py:154-169), summed over utterances and text slots."""

    @staticmethod
    def forward(ctx, a, t, temp):
        a, t = a.contiguous(), t.contiguous()
        loss, sim, na, nt = K.contrastive_fwd(a, t, temp)
        ctx.save_for_backward(a, t, sim, na, nt)
        ctx.temp = temp
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        a, t, sim, na, nt = ctx.saved_tensors
        da, dt = K.contrastive_bwd(a, t, sim, na, nt, dloss.reshape(1).float().contiguous(), ctx.temp)
        return da, dt, None


def contrastive_loss(mem_audio_bm, mem_text_bm, temp):
    """mem_* [B, M, C] batch-major (same dtype) -> fp32 scalar, sum over utterances and text slots."""
    assert mem_audio_bm.shape == mem_text_bm.shape and mem_audio_bm.dtype == mem_text_bm.dtype
    return _ContrastiveFn.apply(mem_audio_bm, mem_text_bm, float(temp))
