"""Kernel-level parity on a real MI355X: every C-ABI entry point against a plain torch fp32
restatement of the same op on the same (dtype-rounded) inputs.

Tolerances (stated per dtype):
  f32 : rtol 2e-4, atol 2e-4 * scale   (fp32 products, different summation order)
  bf16: rtol 2e-2, atol 2e-2 * scale   (inputs identical bf16 values; fp32 accumulate; one bf16
        rounding of the output (2^-8) plus, in attention, bf16 rounding of P / dS like the reference's
        half-precision path)
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import load_pkg

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dt):
    return (2e-4, 2e-4) if dt == torch.float32 else (2e-2, 2e-2)


def check(got, ref, dt, what, scale=None):
    rt, at = tol(dt)
    got, ref = got.float(), ref.float()
    s = float(ref.abs().max()) if scale is None else scale
    err = (got - ref).abs()
    bound = at * max(s, 1e-6) + rt * ref.abs()
    bad = (err > bound)
    assert not torch.isnan(got).any(), what + ": NaN in output"
    assert not bad.any(), "%s: max err %.3e (ref max %.3e), %d/%d out of tolerance" % (
        what, float(err.max()), s, int(bad.sum()), bad.numel())


@pytest.fixture(scope="module")
def K():
    pkg = load_pkg()
    from importlib import import_module
    return import_module("chimera-st_amd.kernels"), import_module("chimera-st_amd.lib")


def rnd(*shape, dt, dev="cuda", scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) % 100000))
    return (torch.randn(*shape, generator=g) * scale).to(dt).to(dev)


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (304, 200, 136), (1000, 520, 264), (64, 48, 2056)])
def test_gemm_layouts(K, dt, ak, bk, M, N, K_):
    k, L = K
    # asymmetric operands (transposes are detectable)
    A = rnd(M, K_, dt=dt, seed=1) if ak else rnd(K_, M, dt=dt, seed=1)
    B = rnd(N, K_, dt=dt, seed=2) if bk else rnd(K_, N, dt=dt, seed=2)
    Aop = A.float() if ak else A.float().t()
    Bop = B.float().t() if bk else B.float()
    ref = Aop @ Bop
    C = torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(A, B, C, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.stride(0), ldb=B.stride(0), ldc=N, split_k=1)
    check(C, ref, dt, "gemm %d%d %dx%dx%d" % (ak, bk, M, N, K_))


@pytest.mark.parametrize("dt", DT)
def test_gemm_epilogue_bias_gelu_aux_resid(K, dt):
    k, L = K
    M, N, K_ = 260, 192, 128
    A, B = rnd(M, K_, dt=dt, seed=3), rnd(N, K_, dt=dt, seed=4, scale=0.1)
    bias, resid = rnd(N, dt=dt, seed=5), rnd(M, N, dt=dt, seed=6)
    C = torch.empty(M, N, dtype=dt, device="cuda")
    Z = torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(A, B, C, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, bias=bias, act=L.ACT_GELU, aux_out=Z,
           ld_aux_out=N, resid=resid, ld_resid=N, alpha=0.5, split_k=1)
    z = 0.5 * (A.float() @ B.float().t()) + bias.float()
    check(Z, z, dt, "aux_out")
    check(C, F.gelu(z) + resid.float(), dt, "bias+gelu+resid")
    # relu + row bias + dact epilogue
    rb = rnd(M, dt=dt, seed=7)
    zin = rnd(M, N, dt=dt, seed=8)
    k.gemm(A, B, C, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, bias=rb, bias_mode=L.BIAS_ROW, act=L.ACT_RELU,
           dact=L.ACT_GELU, aux_in=zin, ld_aux_in=N, split_k=1)
    zz = zin.float()
    dgelu = 0.5 * (1 + torch.erf(zz / math.sqrt(2))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2 * math.pi)
    check(C, torch.relu(A.float() @ B.float().t() + rb.float()[:, None]) * dgelu, dt, "rowbias+relu+dgelu")


@pytest.mark.parametrize("dt", DT)
def test_gemm_splitk_and_f32_out(K, dt):
    k, L = K
    M, N, K_ = 96, 160, 4096
    A, B = rnd(K_, M, dt=dt, seed=9, scale=0.3), rnd(K_, N, dt=dt, seed=10, scale=0.3)
    ref = A.float().t() @ B.float()
    for sk in (-1, 4, 7):
        C = torch.empty(M, N, dtype=torch.float32, device="cuda")
        k.gemm(A, B, C, M, N, K_, a_kmajor=0, b_kmajor=0, lda=M, ldb=N, ldc=N, split_k=sk)
        check(C, ref, dt, "split_k=%d f32 out" % sk)


@pytest.mark.parametrize("dt", DT)
def test_gemm_batched(K, dt):
    k, L = K
    b0, b1, M, N, K_ = 3, 2, 70, 40, 72
    A = rnd(b0, b1, M, K_, dt=dt, seed=11)
    B = rnd(b1, N, K_, dt=dt, seed=12)
    C = torch.empty(b0, b1, M, N, dtype=dt, device="cuda")
    bias = rnd(b1, N, dt=dt, seed=13)
    k.gemm(A, B, C, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, batch0=b0, batch1=b1, sa=(b1 * M * K_, M * K_),
           sb=(0, N * K_), sc=(b1 * M * N, M * N), bias=bias, sbias=(0, N), split_k=1)
    ref = torch.einsum("xymk,ynk->xymn", A.float(), B.float()) + bias.float()[None, :, None, :]
    check(C, ref, dt, "batched gemm")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("k_,s_", [(3, 2), (2, 2), (5, 2)])
def test_gemm_implicit_conv1d(K, dt, k_, s_):
    """channels-last conv1d as a GEMM over overlapping rows (lda = stride*Cin < K = k*Cin)."""
    k, L = K
    B_, Lin, Cin, Cout = 3, 61, 32, 48
    x = rnd(B_, Lin, Cin, dt=dt, seed=14)
    w = rnd(Cout, Cin, k_, dt=dt, seed=15, scale=0.2)
    Lout = (Lin - k_) // s_ + 1
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), stride=s_).transpose(1, 2)
    wcl = w.permute(0, 2, 1).contiguous().view(Cout, k_ * Cin)
    y = torch.empty(B_, Lout, Cout, dtype=dt, device="cuda")
    k.gemm(x, wcl, y, Lout, Cout, k_ * Cin, a_kmajor=1, b_kmajor=1, lda=s_ * Cin, ldb=k_ * Cin, ldc=Cout, batch0=B_,
           sa=(Lin * Cin, 0), sc=(Lout * Cout, 0), split_k=1)
    check(y, ref, dt, "implicit conv k%d s%d" % (k_, s_))
    # weight gradient: dW[co, (j,ci)] = sum_t dy[t,co] x[t*s + j, ci]   (both operands mn-major, overlapping B rows)
    dy = rnd(B_, Lout, Cout, dt=dt, seed=16)
    dw = torch.empty(B_, Cout, k_ * Cin, dtype=torch.float32, device="cuda")
    k.gemm(dy, x, dw, Cout, k_ * Cin, Lout, a_kmajor=0, b_kmajor=0, lda=Cout, ldb=s_ * Cin, ldc=k_ * Cin, batch0=B_,
           sa=(Lout * Cout, 0), sb=(Lin * Cin, 0), sc=(Cout * k_ * Cin, 0), split_k=1)
    xr = x.float().transpose(1, 2).detach().requires_grad_(False)
    wref = w.float().detach().requires_grad_(True)
    out = F.conv1d(xr, wref, stride=s_)
    out.backward(dy.float().transpose(1, 2))
    check(dw.sum(0).view(Cout, k_, Cin).permute(0, 2, 1), wref.grad, dt, "implicit conv dW k%d s%d" % (k_, s_))


@pytest.mark.parametrize("dt", DT)
def test_gemm_grouped_posconv(K, dt):
    """wav2vec2 pos_conv (grouped, k=16 here) as segmented-K implicit GEMM with fused bias+GELU+residual."""
    k, L = K
    B_, T, C, G, Kp = 2, 37, 64, 4, 16
    cg = C // G
    x = rnd(B_, T, C, dt=dt, seed=17)
    w = rnd(C, cg, Kp, dt=dt, seed=18, scale=0.1)  # [Cout, Cin/g, k]
    bias = rnd(C, dt=dt, seed=19)
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias.float(), padding=Kp // 2, groups=G)[:, :, :-1]
    ref = x.float() + F.gelu(ref).transpose(1, 2)
    xp = torch.zeros(B_, T + Kp, C, dtype=dt, device="cuda")
    xp[:, Kp // 2:Kp // 2 + T] = x
    wg = w.view(G, cg, cg, Kp).permute(0, 1, 3, 2).contiguous()  # [g][co][j][ci]
    y = torch.empty(B_, T, C, dtype=dt, device="cuda")
    z = torch.empty(B_, T, C, dtype=dt, device="cuda")
    k.gemm(xp, wg, y, T, cg, Kp * cg, a_kmajor=1, b_kmajor=1, lda=C, ldb=Kp * cg, ldc=C, a_seg=cg, a_seg_stride=C,
           batch0=B_, batch1=G, sa=((T + Kp) * C, cg), sb=(0, cg * Kp * cg), sc=(T * C, cg), bias=bias, sbias=(0, cg),
           act=L.ACT_GELU, aux_out=z, ld_aux_out=C, resid=x, ld_resid=C, split_k=1)
    check(y, ref, dt, "grouped pos-conv")


@pytest.mark.parametrize("dt", DT)
def test_pos_conv_function_wav2vec_small_shape(K, dt):
    """functional.pos_conv_gelu_residual at the wav2vec2-small shape (C 768, 16 groups of 48 channels, k 128; T > 256 so the
    narrow-tile GEMM configurations run): forward and all three gradients against torch's grouped conv1d."""
    CF = __import__("importlib").import_module("chimera-st_amd.functional")
    B_, T, C, G, Kp = 2, 300, 768, 16, 128
    x = rnd(B_, T, C, dt=dt, seed=60, scale=0.5).requires_grad_(True)
    w = rnd(C, C // G, Kp, dt=dt, seed=61, scale=0.02).requires_grad_(True)
    bias = rnd(C, dt=dt, seed=62, scale=0.1).requires_grad_(True)
    y = CF.pos_conv_gelu_residual(x, w, bias, G)
    dy = rnd(B_, T, C, dt=dt, seed=63)
    y.backward(dy)
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, bias))
    conv = F.conv1d(xr.transpose(1, 2), wr, br, padding=Kp // 2, groups=G)[:, :, :-1]
    ref = xr + F.gelu(conv).transpose(1, 2)
    ref.backward(dy.float())
    check(y, ref, dt, "pos-conv fwd")
    check(x.grad, xr.grad, dt, "pos-conv dx")
    check(w.grad, wr.grad, dt, "pos-conv dw", scale=float(wr.grad.abs().max()))
    check(bias.grad, br.grad, dt, "pos-conv db", scale=float(br.grad.abs().max()))


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,cols", [(7, 64), (1000, 512), (333, 768), (50, 1024)])
def test_layernorm(K, dt, rows, cols):
    k, L = K
    x, res = rnd(rows, cols, dt=dt, seed=20), rnd(rows, cols, dt=dt, seed=21)
    g, b = (1 + 0.1 * torch.randn(cols)).to(dt).cuda(), (0.1 * torch.randn(cols)).to(dt).cuda()
    y, s, mean, rstd = k.layernorm_fwd(x, res, g, b, 1e-5, want_sum=True)
    sr = (x.float() + res.float())
    check(s, sr, dt, "ln sum")
    sref = s.float().detach().requires_grad_(True)  # the kernel normalises the rounded sum it wrote
    gr, br = g.float().requires_grad_(True), b.float().requires_grad_(True)
    yr = F.layer_norm(sref, (cols,), gr, br, 1e-5)
    check(y, yr, dt, "ln fwd")
    dy, dres = rnd(rows, cols, dt=dt, seed=22), rnd(rows, cols, dt=dt, seed=23)
    yr.backward(dy.float())
    dx, dg, db = k.layernorm_bwd(dy, s, g, mean, rstd, dres)
    _, dg_t, db_t = k.layernorm_bwd(dy, s, g, mean, rstd, dres, grad_dtype=dt)  # gradients written in the parameter dtype
    assert dg_t.dtype == dt and torch.equal(dg_t, dg.to(dt)) and torch.equal(db_t, db.to(dt))
    check(dx, sref.grad + dres.float(), dt, "ln dx")
    check(dg, gr.grad, dt, "ln dgamma")
    check(db, br.grad, dt, "ln dbeta")


# --------------------------------------------------------------------------------------------
def attn_ref(q, k, v, H, kpm, causal, scale):
    """[B,T,C] fp32 reference written like modules/multihead_attention.py:326-361."""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    D = C // H
    qh = q.view(B, Tq, H, D).transpose(1, 2)
    kh = k.view(B, Tk, H, D).transpose(1, 2)
    vh = v.view(B, Tk, H, D).transpose(1, 2)
    w = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        w = w + torch.triu(torch.full((Tq, Tk), float("-inf"), device=q.device), 1 + Tk - Tq)
    if kpm is not None:
        w = w.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(w, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, C)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,D,Tq,Tk,causal,pad,layout", [
    (2, 2, 32, 40, 40, False, True, "bt"),
    (2, 3, 64, 200, 200, False, True, "bt"),
    (1, 2, 64, 150, 150, True, False, "tb"),
    (2, 2, 64, 8, 77, False, False, "bt"),      # memory slots -> encoder frames
    (3, 4, 64, 33, 64, False, True, "tb"),      # decoder cross-attention
    (2, 2, 64, 1, 19, False, False, "bt"),      # incremental decode step
    (2, 8, 64, 128, 128, True, True, "bt"),
])
def test_attention_fwd_bwd(K, dt, B, H, D, Tq, Tk, causal, pad, layout):
    k, L = K
    C = H * D
    q, kk, v = rnd(B, Tq, C, dt=dt, seed=30), rnd(B, Tk, C, dt=dt, seed=31), rnd(B, Tk, C, dt=dt, seed=32)
    kpm = None
    if pad:
        lens = torch.tensor([Tk - (3 * i + 1) % max(Tk // 2, 1) for i in range(B)])
        kpm = (torch.arange(Tk)[None] >= lens[:, None]).to(torch.uint8).cuda()
    scale = D ** -0.5
    qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, kk, v))
    ref = attn_ref(qr, kr, vr, H, kpm, causal, scale)
    do = rnd(B, Tq, C, dt=dt, seed=33)
    ref.backward(do.float())
    if layout == "tb":
        qx, kx, vx, dox = (t.transpose(0, 1).contiguous() for t in (q, kk, v, do))
    else:
        qx, kx, vx, dox = q, kk, v, do
    o, lse = k.attn_fwd(qx, kx, vx, H, D, kpm, causal, scale, layout, layout)
    dq, dk, dv = k.attn_bwd(dox, qx, kx, vx, o, lse, H, D, kpm, causal, scale, layout, layout)
    if layout == "tb":
        o, dq, dk, dv = (t.transpose(0, 1) for t in (o, dq, dk, dv))
    check(o, ref, dt, "attn fwd")
    check(dq, qr.grad, dt, "attn dq")
    check(dk, kr.grad, dt, "attn dk")
    check(dv, vr.grad, dt, "attn dv")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Tq,Tk,drop", [(300, 300, 0.0), (130, 517, 0.0), (257, 257, 0.1)])
def test_attention_kv_len_skip_is_bit_identical(K, dt, Tq, Tk, drop):
    """cst_attn_desc.kv_len (skip the all-padding key tiles at the end of each utterance): outputs and all three gradients are
    bit-identical to the run that walks every tile (the skipped tiles contribute exact zeros), incl. an all-padding row."""
    k, L = K
    B, H, D = 4, 2, 64
    C = H * D
    q, kk, v, do = (rnd(B, T, C, dt=dt, seed=50 + i) for i, T in enumerate((Tq, Tk, Tk, Tq)))
    lens = torch.tensor([Tk, Tk // 2 + 3, 65, 1])
    kpm = (torch.arange(Tk)[None] >= lens[:, None]).to(torch.uint8).cuda()
    kpm[1, 5] = 1  # a masked key INSIDE the valid range stays governed by the mask
    kvl = lens.to(torch.int32).cuda()
    scale = D ** -0.5
    outs = []
    for kv in (None, kvl):
        o, lse = k.attn_fwd(q, kk, v, H, D, kpm, False, scale, "bt", "bt", drop, 1234, kv_len=kv)
        dq, dk, dv = k.attn_bwd(do, q, kk, v, o, lse, H, D, kpm, False, scale, "bt", "bt", drop, 1234, kv_len=kv)
        outs.append((o, lse, dq, dk, dv))
    for a, b, name in zip(outs[0], outs[1], ("o", "lse", "dq", "dk", "dv")):
        assert torch.equal(a, b), name
    assert float(outs[1][3][2, 65:].abs().max()) == 0.0 and float(outs[1][4][3, 1:].abs().max()) == 0.0
    # zero upstream gradient on trailing query tiles (padded frames nothing downstream reads): the backward stops at the last live
    # tile; gradients must equal the full walk bit for bit (here: q_flags on vs a descriptor without the workspace)
    do2 = do.clone()
    cut = Tq // 2
    do2[:, cut:] = 0
    do2[3] = 0
    do2[0, :128] = 0  # a dead 128-query block in FRONT of live ones: the dQ kernel (which finds this out itself) writes zeros
    o, lse = k.attn_fwd(q, kk, v, H, D, kpm, False, scale, "bt", "bt", drop, 1234, kv_len=kvl)
    got = k.attn_bwd(do2, q, kk, v, o, lse, H, D, kpm, False, scale, "bt", "bt", drop, 1234, kv_len=kvl)
    dq0, dk0, dv0 = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
    d = k.attn_desc(q, kk, v, o, lse, H, D, kpm, False, scale, "bt", "bt", drop, 1234, kvl)
    k.attn_bwd_fill(d, do2, dq0, dk0, dv0, torch.empty_like(lse), D)
    d.q_flags = None
    k.attn_bwd_desc(d)
    for a, b, name in zip(got, (dq0, dk0, dv0), ("dq", "dk", "dv")):
        assert torch.equal(a, b), "live-tile skip changed " + name
    assert float(got[0][:, cut:].abs().max()) == 0.0 and float(got[0][3].abs().max()) == 0.0 and float(got[0][0, :128].abs().max()) == 0.0
    # the host helper derives kv_len from the mask
    CF = __import__("importlib").import_module("chimera-st_amd.functional")
    u8, got = CF._mask_and_len(kpm.bool())
    assert got.tolist() == lens.tolist() and torch.equal(u8, kpm)


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,S,C", [(2, 4000, 32), (2, 16000, 512)])
def test_conv0_gn_gelu(K, dt, B, S, C):
    k, L = K
    kk, st = 10, 5
    wav = (0.1 * torch.randn(B, S, generator=torch.Generator().manual_seed(40))).cuda()
    w = rnd(C, kk, dt=dt, seed=41, scale=0.5)
    g, b = (1 + 0.1 * torch.randn(C)).to(dt).cuda(), (0.1 * torch.randn(C)).to(dt).cuda()
    wr, gr, br = w.float().requires_grad_(True), g.float().requires_grad_(True), b.float().requires_grad_(True)
    u = F.conv1d(wav[:, None], wr[:, None], stride=st)
    ref = F.gelu(F.group_norm(u, C, gr, br, 1e-5)).transpose(1, 2)
    y, mean, rstd, gram = k.conv0_fwd(wav, w, g, b, kk, st)
    check(y, ref, dt, "conv0 fwd")
    dy = rnd(B, ref.shape[1], C, dt=dt, seed=42)
    ref.backward(dy.float())
    dw, dg, db = k.conv0_bwd(dy, wav, w, g, b, mean, rstd, gram, kk, st)
    # bf16: backward recomputes z from bf16 weights exactly like forward; tolerance is on reductions over B*L terms
    check(dw, wr.grad, dt, "conv0 dW", scale=float(wr.grad.abs().max()))
    check(dg, gr.grad, dt, "conv0 dgamma")
    check(db, br.grad, dt, "conv0 dbeta")


@pytest.mark.parametrize("dt", DT)
def test_conv0_frame_limits_are_exact(K, dt):
    """cst_conv0_gn_gelu_fwd/bwd with frame_limit: forward writes exactly the frames below the limit (same bits as the unlimited call)
    and leaves the rest of the buffer untouched; backward with a dy that is zero from the limit on gives the same bits as reading
    every frame.  cst_conv_row_limits supplies the two limits from the real-frame counts behind the last conv layer."""
    k, L = K
    kk, st, B, S, C = 10, 5, 3, 16000, 512
    wav = (0.1 * torch.randn(B, S, generator=torch.Generator().manual_seed(40))).cuda()
    w = rnd(C, kk, dt=dt, seed=41, scale=0.5)
    g, b = (1 + 0.1 * torch.randn(C)).to(dt).cuda(), (0.1 * torch.randn(C)).to(dt).cuda()
    spec = [(512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 2, 2)]
    lens = [S]
    for (_, kq, sq) in spec:
        lens.append((lens[-1] - kq) // sq + 1)
    nz_last = torch.tensor([lens[-1], lens[-1] // 3, 0], dtype=torch.int32, device="cuda")
    lim = k.conv_row_limits(nz_last, spec, S)
    # the recursion by hand
    for bi in range(B):
        cur = int(nz_last[bi])
        for i in range(len(spec) - 1, 0, -1):
            cur = min(cur, lens[i + 1])
            assert int(lim[i, 0, bi]) == cur
            for r in range(spec[i][2]):
                below = (cur - 1) * spec[i][2] + spec[i][1] if cur > 0 else 0
                assert int(lim[i, 1 + r, bi]) == max(0, -(-(min(below, lens[i]) - r) // spec[i][2]))
            cur = (cur - 1) * spec[i][2] + spec[i][1] if cur > 0 else 0
        assert int(lim[0, 0, bi]) == min(cur, lens[1])
        assert int(lim[0, 1, bi]) == min(lens[1], -(-int(lim[1, 0, bi]) // 256) * 256 * spec[1][2] + spec[1][1])
    y_full, mean, rstd, gram = k.conv0_fwd(wav, w, g, b, kk, st)
    wl, gl = lim[0, 1].contiguous(), lim[0, 0].contiguous()
    k.workspace  # noqa: B018 (the limited call below re-uses the same scratch)
    y_lim, mean2, rstd2, gram2 = k.conv0_fwd(wav, w, g, b, kk, st, frame_limit=wl)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(gram, gram2)
    for bi in range(B):
        n = int(wl[bi])
        assert torch.equal(y_lim[bi, :n], y_full[bi, :n])
    dy = rnd(B, y_full.shape[1], C, dt=dt, seed=42)
    for bi in range(B):
        dy[bi, int(gl[bi]):] = 0
    ref = k.conv0_bwd(dy, wav, w, g, b, mean, rstd, gram, kk, st)
    got = k.conv0_bwd(dy, wav, w, g, b, mean, rstd, gram, kk, st, frame_limit=gl)
    for a_, b_ in zip(got, ref):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,cols", [(47968, 768), (4128, 512), (333, 64)])
def test_dropout_colsum_equals_dropout_then_colsum(K, dt, rows, cols):
    """cst_dropout_colsum: the masked gradient and its column sums in one pass, bit-equal to cst_dropout followed by
    cst_colsum_typed (the sums are taken over the values as stored); with live-tile stamps, dead tiles come back as zeros."""
    k, L = K
    x = rnd(rows, cols, dt=dt, seed=60)
    p, key = 0.1, 0x1234567
    xd, db = k.dropout_colsum(x, p, key, dt)
    ref = k.dropout(x, p, key)
    assert torch.equal(xd, ref)
    assert torch.equal(db, k.colsum(ref, dt))
    # stamps: rows of every second 64-row tile are zero and marked dead
    x2 = x.clone()
    ntile = (rows + 63) // 64
    stamps = torch.zeros(ntile, dtype=torch.int32, device="cuda")
    epoch = 7
    for t in range(ntile):
        if t % 2 == 0:
            stamps[t] = epoch
        else:
            x2[t * 64:(t + 1) * 64] = 0
    xd2, db2 = k.dropout_colsum(x2, p, key, dt, (stamps, epoch))
    ref2 = k.dropout(x2, p, key)
    assert torch.equal(xd2, ref2)
    assert torch.equal(db2, k.colsum(ref2, dt, (stamps, epoch)))
    # second stage left to the deferred-reduction flush (out = NULL in the C ABI, cst_reduce_multi order 1): the same bits, and the
    # destination is untouched (NaN under CST_DEFER_POISON) until the flush
    with k.deferred_reductions(True):
        xd3, db3 = k.dropout_colsum(x2, p, key, dt, (stamps, epoch), defer=True)
        cs3 = k.colsum(ref, dt, defer=True)
        cs4 = k.colsum(ref2, dt, (stamps, epoch), defer=True)
        assert len(k.DEFER.items) == 3
        if k.DEFER.poison:
            assert torch.isnan(db3.float()).all() and torch.isnan(cs3.float()).all()
    assert torch.equal(xd3, ref2) and torch.equal(db3, db2) and torch.equal(cs3, db) and torch.equal(cs4, db2)
    assert len(k.DEFER.items) == 0
    xd5, db5 = k.dropout_colsum(x, p, key, dt, defer=True)  # outside the mode: the immediate route
    assert torch.equal(db5, db) and len(k.DEFER.items) == 0


@pytest.mark.parametrize("dt", DT)
def test_elementwise(K, dt):
    k, L = K
    z = rnd(77, 128, dt=dt, seed=50)
    dy = rnd(77, 64, dt=dt, seed=51)
    zr = z.float().requires_grad_(True)
    yr = F.glu(zr, dim=1)
    yr.backward(dy.float())
    check(k.glu_fwd(z), yr, dt, "glu fwd")
    check(k.glu_bwd(dy, z), zr.grad, dt, "glu bwd")
    for act, fn in ((L.ACT_GELU, F.gelu), (L.ACT_RELU, torch.relu)):
        x = rnd(40, 64, dt=dt, seed=52)
        xr = x.float().requires_grad_(True)
        o = fn(xr)
        g = rnd(40, 64, dt=dt, seed=53)
        o.backward(g.float())
        check(k.act_fwd(x, act), o, dt, "act fwd")
        check(k.act_bwd(g, x, act), xr.grad, dt, "act bwd")
    x = rnd(1234, 72, dt=dt, seed=54)
    check(k.colsum(x), x.float().sum(0), dt, "colsum")
    for rows, cols in ((47968, 768), (33, 8), (4096, 10000)):
        big = rnd(rows, cols + 8, dt=dt, seed=55)[:, :cols]
        ref = big.float().sum(0)
        a = k.colsum(big, dt)
        assert a.dtype == dt
        check(a, ref, dt, "colsum %dx%d" % (rows, cols), scale=float(ref.abs().max()))
        assert torch.equal(a, k.colsum(big, dt))  # fixed-order partials + reduce: bit-reproducible
        check(k.colsum_atomic(big), ref, torch.float32, "colsum (atomics) %dx%d" % (rows, cols), scale=float(ref.abs().max()) * (1 if dt == torch.float32 else 4))
    m = (torch.arange(1234) % 3 == 0).to(torch.uint8).cuda()
    check(k.mask_rows(x, m), x.float() * (1 - m.float())[:, None], dt, "mask_rows")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("k_,s_,pad", [(3, 2, 0), (2, 2, 0), (5, 2, 2)])
def test_col2im(K, dt, k_, s_, pad):
    k, L = K
    B_, Lin, C = 2, 41, 16
    Lout = (Lin + 2 * pad - k_) // s_ + 1
    dcol = rnd(B_, Lout, k_ * C, dt=dt, seed=55)
    z = rnd(B_, Lin, C, dt=dt, seed=56)
    ref = torch.zeros(B_, Lin + 2 * pad, C, device="cuda")
    d4 = dcol.float().view(B_, Lout, k_, C)
    for t in range(Lout):
        for j in range(k_):
            ref[:, t * s_ + j] += d4[:, t, j]
    ref = ref[:, pad:pad + Lin]
    check(k.col2im1d(dcol, None, B_, Lin, Lout, C, k_, s_, pad, 0), ref, dt, "col2im")
    zz = z.float()
    dg = 0.5 * (1 + torch.erf(zz / math.sqrt(2))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2 * math.pi)
    check(k.col2im1d(dcol, z, B_, Lin, Lout, C, k_, s_, pad, L.ACT_GELU), ref * dg, dt, "col2im+dgelu")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,V", [(37, 60), (300, 10000)])
def test_label_smoothed_ce(K, dt, rows, V):
    k, L = K
    logits = rnd(rows, V, dt=dt, seed=60, scale=2.0)
    tgt = torch.randint(4, V, (rows,), generator=torch.Generator().manual_seed(61))
    tgt[::5] = 1
    tgt = tgt.cuda()
    lr = logits.float().requires_grad_(True)
    lp = torch.log_softmax(lr, -1)
    nll = -lp.gather(1, tgt[:, None]).squeeze(1)
    sm = -lp.sum(-1)
    padm = tgt.eq(1)
    nll_s = nll.masked_fill(padm, 0).sum()
    loss = 0.9 * nll_s + 0.1 / V * sm.masked_fill(padm, 0).sum()
    (loss * 0.37).backward()
    out2, lse = k.ls_ce_fwd(logits, tgt, 0.1, 1)
    check(out2, torch.stack([loss, nll_s]).detach(), torch.float32, "ls-ce loss")
    gs = torch.tensor([0.37], device="cuda")
    d = k.ls_ce_bwd(logits, tgt, lse, gs, 0.1, 1)
    check(d, lr.grad, dt, "ls-ce dlogits", scale=0.37)


def test_adam_and_sumsq(K):
    k, L = K
    n = 100003
    for gdt, pdt in ((torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)):
        master = rnd(n, dt=torch.float32, seed=70)
        m, v = rnd(n, dt=torch.float32, seed=71, scale=0.1), rnd(n, dt=torch.float32, seed=72).abs() * 0.01
        g = rnd(n, dt=gdt, seed=73)
        p = torch.empty(n, dtype=pdt, device="cuda")
        rm, rmm, rv = master.clone(), m.clone(), v.clone()
        gs = torch.tensor([0.25], device="cuda")
        k.adam_step(master, m, v, g, p, 1e-3, 0.9, 0.98, 1e-8, 0.01, 3, gs)
        gg = g.float() * 0.25
        rmm.mul_(0.9).add_(gg, alpha=0.1)
        rv.mul_(0.98).addcmul_(gg, gg, value=0.02)
        step_size = 1e-3 * math.sqrt(1 - 0.98 ** 3) / (1 - 0.9 ** 3)
        rm.add_(rm, alpha=-0.01 * 1e-3)
        rm.addcdiv_(rmm, rv.sqrt().add_(1e-8), value=-step_size)
        check(master, rm, torch.float32, "adam master")
        check(m, rmm, torch.float32, "adam m")
        check(v, rv, torch.float32, "adam v")
        check(p, rm, pdt, "adam model param")
        out = torch.zeros(1, device="cuda")
        k.sumsq(g, out)
        check(out, (g.float() ** 2).sum()[None], torch.float32, "sumsq")


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Lin", [97, 200, 333])
def test_conv_stack_windowed_dx(dt, Lin):
    """wav2vec2 conv layers 1.. as implicit GEMMs: forward, dX through the residue-class windowed GEMMs (row-padded dz,
    GELU' folded into the epilogue, no col2im) and dW, against torch autograd of conv1d+GELU on the same values."""
    load_pkg()
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    B, C = 3, 64
    spec = [(3, 2), (3, 2), (2, 2), (2, 2)]
    x = rnd(B, Lin, C, dt=dt, seed=5).requires_grad_(True)
    ws = [(rnd(C, C, k, dt=dt, seed=10 + i, scale=1.0 / math.sqrt(C * k))).requires_grad_(True) for i, (k, s) in enumerate(spec)]
    y, z = x, None
    for i, (k, s) in enumerate(spec):
        y, z = CF.conv1d_cl(y, ws[i], None, s, pad=0, act="gelu", prev_z=z, grad_is_dz=(i < len(spec) - 1))
    g = rnd(*y.shape, dt=dt, seed=77)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = [w.detach().float().requires_grad_(True) for w in ws]
    h = xr.transpose(1, 2)
    for i, (k, s) in enumerate(spec):
        h = F.gelu(F.conv1d(h, wr[i], stride=s))
        if dt != torch.float32:
            h = h.to(dt).float()  # the HIP path stores every layer's activation in bf16
    ref = h.transpose(1, 2)
    check(y, ref, dt, "conv stack fwd")
    ref.backward(g.float())
    # layer 0 of this stack has no GELU'(prev_z) folded in (prev_z=None), like layer 1 behind conv0
    check(x.grad, xr.grad, dt, "conv stack dX", scale=float(xr.grad.abs().max()) * (1 if dt == torch.float32 else 2))
    for i in range(len(spec)):
        check(ws[i].grad, wr[i].grad, dt, "conv stack dW%d" % i, scale=float(wr[i].grad.abs().max()) * (1 if dt == torch.float32 else 2))


@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K_", [(3608, 3592, 328), (512, 1024, 16648), (4096, 4096, 64)])
def test_gemm_8phase_large(K, ak, bk, M, N, K_):
    """bf16 problems that fill the chip take the 256x256x64 8-phase DMA kernel (gemm8p.hip): ragged M/N/K edges, a K
    shorter than the prologue depth, and split-K with empty trailing splits."""
    k, L = K
    dt = torch.bfloat16
    A = rnd(M, K_, dt=dt, seed=1) if ak else rnd(K_, M, dt=dt, seed=1)
    B = rnd(N, K_, dt=dt, seed=2) if bk else rnd(K_, N, dt=dt, seed=2)
    bias = rnd(N, dt=dt, seed=3)
    C = torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(A, B, C, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.shape[1], ldb=B.shape[1], ldc=N, bias=bias, act=L.ACT_RELU)
    Af = A.float() if ak else A.float().t()
    Bf = B.float() if bk else B.float().t()
    ref = torch.relu(Af @ Bf.t() + bias.float())
    check(C, ref, dt, "8-phase gemm %d%d %dx%dx%d" % (ak, bk, M, N, K_))


@pytest.mark.parametrize("dt", DT)
def test_gemm_live_k_tiles_skip(K, dt):
    """Weight-gradient GEMM (k = token, both operands token-major) with 64-token blocks of dY that are exactly zero: the launch
    that is told so (cst_gemm_desc.k_live, stamps from the LayerNorm backward) must return the same dW (bit-identical with one
    split), zeros when every block is dead, and the stamps themselves must mark exactly the tiles holding a non-zero dx row."""
    k, L = K
    from importlib import import_module
    Kn = import_module("chimera-st_amd.kernels")
    rows, C, F_ = 64 * 37 + 24, 256, 512
    x = rnd(rows, C, dt=dt, seed=1)
    gamma, beta = rnd(C, dt=dt, seed=2) + 1.0, rnd(C, dt=dt, seed=3)
    y, _, mean, rstd = k.layernorm_fwd(x, None, gamma, beta, 1e-5)
    dy = rnd(rows, C, dt=dt, seed=4)
    dead = torch.zeros(rows, dtype=torch.bool, device="cuda")
    for lo, hi in ((0, 130), (700, 1500), (2300, rows)):
        dead[lo:hi] = True
    dy[dead] = 0
    dx, dg, db, (stamps, epoch) = k.layernorm_bwd(dy, x, gamma, mean, rstd, None, grad_dtype=dt, want_tiles=True)
    dx0, dg0, db0 = k.layernorm_bwd(dy, x, gamma, mean, rstd, None, grad_dtype=dt)
    assert torch.equal(dx, dx0) and torch.equal(dg, dg0) and torch.equal(db, db0)
    nz_row = (dx.float() != 0).any(dim=1)
    assert not nz_row[dead].any()
    ntile = (rows + 63) // 64
    live_ref = torch.stack([nz_row[i * 64:(i + 1) * 64].any() for i in range(ntile)])
    assert torch.equal(stamps == epoch, live_ref) and 0 < int(live_ref.sum()) < ntile
    assert torch.equal(k.colsum(dx, dt, (stamps, epoch)), k.colsum(dx, dt))  # bias gradient: dead tiles skipped, same bits
    h = rnd(rows, F_, dt=dt, seed=5)
    n0 = Kn.STATS.get("gemm_k_live", 0)
    for split in (-1, 1, 3):
        dw_ref = torch.empty(C, F_, dtype=dt, device="cuda")
        k.gemm(dx, h, dw_ref, C, F_, rows, a_kmajor=0, b_kmajor=0, lda=C, ldb=F_, ldc=F_, split_k=split)
        dw = torch.full((C, F_), float("nan"), dtype=dt, device="cuda")
        k.gemm(dx, h, dw, C, F_, rows, a_kmajor=0, b_kmajor=0, lda=C, ldb=F_, ldc=F_, split_k=split, k_live=(stamps, epoch))
        if split == 1:
            assert torch.equal(dw, dw_ref)  # one split: the same products in the same order, minus exact zeros
        else:  # split-K deals the LIVE tiles out evenly: another (equally valid, deterministic) fp32 summation order
            ref32 = dx.float().t() @ h.float()
            check(dw, ref32, dt, "dW, live tiles, split %d" % split, scale=float(ref32.abs().max()))
            dw2 = torch.full((C, F_), float("nan"), dtype=dt, device="cuda")
            k.gemm(dx, h, dw2, C, F_, rows, a_kmajor=0, b_kmajor=0, lda=C, ldb=F_, ldc=F_, split_k=split, k_live=(stamps, epoch))
            assert torch.equal(dw, dw2)
    assert Kn.STATS.get("gemm_k_live", 0) == n0 + 5  # 3 + the two determinism re-runs
    check(dw_ref, dx.float().t() @ h.float(), dt, "dW", scale=float((dx.float().t() @ h.float()).abs().max()))
    # every block dead: zeros
    z = torch.zeros_like(dx)
    none_live = torch.zeros(ntile, dtype=torch.int32, device="cuda")
    dwz = torch.full((C, F_), float("nan"), dtype=dt, device="cuda")
    k.gemm(z, h, dwz, C, F_, rows, a_kmajor=0, b_kmajor=0, lda=C, ldb=F_, ldc=F_, split_k=-1, k_live=(none_live, 7))
    assert torch.equal(dwz, torch.zeros_like(dwz))


@pytest.mark.parametrize("rows,N,Kd", [(64 * 37 + 24, 512, 256), (64 * 700 + 8, 768, 3072), (64 * 700 + 8, 3072, 768)])
def test_gemm_live_m_tiles_skip_is_bit_identical(K, rows, N, Kd):
    """dX-type GEMMs (row-wise in A) with 64-row blocks of A that are exactly zero: told so through cst_gemm_desc.m_live, the
    generic and the persistent kernel skip the K loop of all-dead output tiles; plain, act'(aux_in) and residual epilogues must give
    bit-identical results."""
    k, L = K
    dt = torch.bfloat16
    A = rnd(rows, Kd, dt=dt, seed=1)
    nt = (rows + 63) // 64
    g = torch.Generator().manual_seed(3)
    live = torch.rand(nt, generator=g) > 0.45
    live[5:12] = False  # a run long enough to cover whole 256-row tiles
    live[nt // 2: nt // 2 + 40] = False
    rowlive = live.repeat_interleave(64)[:rows].cuda()
    A[~rowlive] = 0
    epoch = 0x9ABCDEF1
    stamps = torch.where(live, torch.tensor(epoch - 2 ** 32, dtype=torch.int64), torch.tensor(12345, dtype=torch.int64)).to(torch.int32).cuda()
    W = rnd(Kd, N, dt=dt, seed=2)  # mn-major B ([K, N]), as the dX GEMMs have it
    z = rnd(rows, N, dt=dt, seed=4)
    res = rnd(rows, N, dt=dt, seed=5)
    for kw in (dict(), dict(dact=L.ACT_GELU, aux_in=z, ld_aux_in=N), dict(resid=res, ld_resid=N)):
        ref = torch.full((rows, N), float("nan"), dtype=dt, device="cuda")
        k.gemm(A, W, ref, rows, N, Kd, a_kmajor=1, b_kmajor=0, lda=Kd, ldb=N, ldc=N, split_k=1, **kw)
        out = torch.full((rows, N), float("nan"), dtype=dt, device="cuda")
        k.gemm(A, W, out, rows, N, Kd, a_kmajor=1, b_kmajor=0, lda=Kd, ldb=N, ldc=N, split_k=1, m_live=(stamps, epoch), **kw)
        assert torch.equal(out, ref), sorted(kw)
    check(ref, A.float() @ W.float() + res.float(), dt, "dX + resid", scale=float((A.float() @ W.float()).abs().max()))


@pytest.mark.parametrize("split", [1, 2])
def test_gemm_batched_k_len_is_exact(K, split):
    """Per-utterance weight-gradient GEMM of a conv layer (batch0 = utterances, k = frame, both operands frame-major) whose A rows
    are zero from k_len[b] on: the launch told so (cst_gemm_desc.k_len) stops each batch's reduction there — bit-identical with one
    split, the same sums in another split order otherwise; k_len = 0 gives zeros."""
    k, L = K
    dt = torch.bfloat16
    B, Lf, Cout, Cin = 5, 1000, 256, 384
    dz = rnd(B, Lf, Cout, dt=dt, seed=1)
    x = rnd(B, Lf, Cin, dt=dt, seed=2)
    klen = torch.tensor([1000, 777, 64, 1, 0], dtype=torch.int32, device="cuda")
    for b in range(B):
        dz[b, int(klen[b]):] = 0
    def run(**kw):
        out = torch.full((B, Cout, Cin), float("nan"), dtype=torch.float32, device="cuda")
        k.gemm(dz, x, out, Cout, Cin, Lf, a_kmajor=0, b_kmajor=0, lda=Cout, ldb=Cin, ldc=Cin, batch0=B, sa=(Lf * Cout, 0), sb=(Lf * Cin, 0),
               sc=(Cout * Cin, 0), split_k=split, **kw)
        return out
    ref, got = run(), run(k_len=klen)
    if split == 1:
        assert torch.equal(got, ref)
    else:
        assert torch.equal(got, run(k_len=klen))
    ref32 = torch.einsum("blo,bli->boi", dz.float(), x.float())
    check(got, ref32, torch.float32, "batched dW with k_len", scale=float(ref32.abs().max()) * 4)
    assert torch.equal(got[4], torch.zeros_like(got[4]))


@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0), (0, 0)])
def test_gemm_8phase_claimed_items(K, ak, bk):
    """More work items than workgroups: every item after a workgroup's first is claimed from the per-XCD counters (gemm8p.hip).
    Each item must be computed exactly once (ragged 20 x 17 tile grid, also with split-K slabs), repeated launches re-arm their
    own state (bit-identical results), and launches in flight on two streams do not share counters."""
    k, L = K
    dt = torch.bfloat16
    M, N, K_ = 5000, 4104, 200
    A = rnd(M, K_, dt=dt, seed=1) if ak else rnd(K_, M, dt=dt, seed=1)
    B = rnd(N, K_, dt=dt, seed=2) if bk else rnd(K_, N, dt=dt, seed=2)
    Af = A.float() if ak else A.float().t()
    Bf = B.float() if bk else B.float().t()
    ref = Af @ Bf.t()
    outs = []
    for _ in range(3):
        C = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
        k.gemm(A, B, C, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.shape[1], ldb=B.shape[1], ldc=N)
        check(C, ref, dt, "claimed items %d%d" % (ak, bk))
        outs.append(C)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    Cs = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    k.gemm(A, B, Cs, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.shape[1], ldb=B.shape[1], ldc=N, split_k=2)
    check(Cs, ref, dt, "claimed items, split-K")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    C1 = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    C2 = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            k.gemm(A, B, C1, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.shape[1], ldb=B.shape[1], ldc=N)
        with torch.cuda.stream(s2):
            k.gemm(A, B, C2, M, N, K_, a_kmajor=ak, b_kmajor=bk, lda=A.shape[1], ldb=B.shape[1], ldc=N)
    torch.cuda.synchronize()
    assert torch.equal(C1, outs[0]) and torch.equal(C2, outs[0])


# --------------------------------------------------------------------------------------------
# dropout: the kernels' counter-based masks against the numpy statement of the same function (chimera-st_amd/rng.py)
# --------------------------------------------------------------------------------------------
def _keep(key, n, p):
    from importlib import import_module
    rng = import_module("chimera-st_amd.rng")
    return torch.from_numpy(rng.keep_mask_numpy(key, n, p))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n", [8, 1000, 4099, 1 << 20])
def test_dropout_elementwise(K, dt, n):
    k, L = K
    p, key = 0.1, 0x1234ABCD
    x = rnd(n, dt=dt, seed=3)
    y = k.dropout(x, p, key)
    keep = _keep(key, n, p).cuda()
    ref = torch.where(keep, x.float() / (1 - p), torch.zeros_like(x.float()))
    check(y, ref, dt, "dropout n=%d" % n)
    assert abs(float(keep.float().mean()) - 0.9) < (0.35 if n < 100 else 0.03)
    # backward of the same site = the same call on the gradient
    g = rnd(n, dt=dt, seed=4)
    check(k.dropout(g, p, key), torch.where(keep, g.float() / (1 - p), torch.zeros_like(g.float())), dt, "dropout bwd")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("big", [False, True])
def test_gemm_epilogue_dropout(K, dt, big):
    """C = resid + dropout(relu(A B^T + bias)): mask index = row * N + col (old kernels and the 8-phase kernel)."""
    k, L = K
    if big and dt != torch.bfloat16:
        pytest.skip("8-phase kernel is bf16")
    M, N, K_ = (3608, 3592, 136) if big else (304, 200, 136)
    p, key = 0.25, 77
    A, B = rnd(M, K_, dt=dt, seed=1), rnd(N, K_, dt=dt, seed=2)
    bias, resid = rnd(N, dt=dt, seed=3), rnd(M, N, dt=dt, seed=4)
    C = torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(A, B, C, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, bias=bias, act=L.ACT_RELU, resid=resid, ld_resid=N,
           drop_p=p, drop_key=key)
    keep = _keep(key, M * N, p).view(M, N).cuda()
    z = torch.relu(A.float() @ B.float().t() + bias.float())
    ref = resid.float() + torch.where(keep, z / (1 - p), torch.zeros_like(z))
    check(C, ref, dt, "gemm dropout epilogue", scale=float(ref.abs().max()))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Tq,Tk,causal", [(100, 100, False), (77, 131, False), (64, 64, True)])
def test_attention_dropout(K, dt, Tq, Tk, causal):
    """Attention-probability dropout inside the fused kernels, forward and all three gradients, against explicit
    softmax -> mask -> PV in fp32 with the mask from rng.keep_mask_attn_numpy (row (b*H+h)*Tq + q, key k)."""
    k, L = K
    B, H, D = 2, 3, 64
    p, key = 0.2, 4242
    q = rnd(B, Tq, H * D, dt=dt, seed=1).requires_grad_(True)
    kk = rnd(B, Tk, H * D, dt=dt, seed=2).requires_grad_(True)
    v = rnd(B, Tk, H * D, dt=dt, seed=3).requires_grad_(True)
    kpm = torch.zeros(B, Tk, dtype=torch.uint8, device="cuda")
    kpm[1, Tk - 9:] = 1
    scale = D ** -0.5
    o, lse = k.attn_fwd(q.detach(), kk.detach(), v.detach(), H, D, kpm, causal, scale, "bt", "bt", p, key)
    do = rnd(B, Tq, H * D, dt=dt, seed=5)
    dq, dk, dv = k.attn_bwd(do, q.detach(), kk.detach(), v.detach(), o, lse, H, D, kpm, causal, scale, "bt", "bt", p, key)
    from importlib import import_module
    keep = torch.from_numpy(import_module("chimera-st_amd.rng").keep_mask_attn_numpy(key, B * H * Tq, Tk, p)).view(B, H, Tq, Tk).cuda()
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, kk, v))
    qh = qf.view(B, Tq, H, D).transpose(1, 2); kh = kf.view(B, Tk, H, D).transpose(1, 2); vh = vf.view(B, Tk, H, D).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    s = s.masked_fill(kpm.bool()[:, None, None, :], float("-inf"))
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(Tq, Tk, dtype=torch.bool, device="cuda"), 1 + Tk - Tq), float("-inf"))
    pr = torch.softmax(s, -1)
    prd = torch.where(keep, pr / (1 - p), torch.zeros_like(pr))
    ref = (prd @ vh).transpose(1, 2).reshape(B, Tq, H * D)
    check(o, ref, dt, "attn dropout fwd")
    ref.backward(do.float())
    sc = 2.0 if dt == torch.bfloat16 else 1.0
    check(dq, qf.grad, dt, "attn dropout dq", scale=float(qf.grad.abs().max()) * sc)
    check(dk, kf.grad, dt, "attn dropout dk", scale=float(kf.grad.abs().max()) * sc)
    check(dv, vf.grad, dt, "attn dropout dv", scale=float(vf.grad.abs().max()) * sc)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,M,C", [(3, 64, 512), (2, 8, 64), (1, 33, 72)])
def test_contrastive(K, dt, B, M, C):
    """cst_contrastive_fwd/bwd against torch: cosine-similarity logits / temp, CE over the AUDIO slot dim, target(j) = j
    (criterions/triplet_st_mt_contrastive.py:154-169)."""
    k, L = K
    temp = 0.1
    a = rnd(B, M, C, dt=dt, seed=1)
    t = (0.5 * a.float() + rnd(B, M, C, dt=torch.float32, seed=2)).to(dt)
    loss, sim, na, nt = k.contrastive_fwd(a, t, temp)
    ar, tr = a.float().requires_grad_(True), t.float().requires_grad_(True)
    logits = F.cosine_similarity(ar.unsqueeze(2), tr.unsqueeze(1), dim=-1) / temp
    ref = F.cross_entropy(logits, torch.arange(M, device="cuda")[None].repeat(B, 1), reduction="sum")
    assert abs(float(loss) - float(ref)) <= 2e-4 * abs(float(ref)) + 1e-3, (float(loss), float(ref))
    ref.backward()
    g = torch.full((1,), 0.7, device="cuda")
    da, dtt = k.contrastive_bwd(a, t, sim, na, nt, g, temp)
    check(da, 0.7 * ar.grad, dt, "contrastive da", scale=float(ar.grad.abs().max()))
    check(dtt, 0.7 * tr.grad, dt, "contrastive dt", scale=float(tr.grad.abs().max()))


def test_live_tile_stamps_are_dropped_when_the_gradient_is_rewritten():
    """functional._tiles_of must not hand out stamps for a tensor that was written after stamping (autograd may accumulate a
    second gradient into the first one in place), for a transposed view, or for a different tensor."""
    from importlib import import_module
    load_pkg()
    F_ = import_module("chimera-st_amd.functional")
    rows, C = 640, 64
    base = torch.zeros(rows, C, device="cuda")
    stamps = torch.zeros(rows // 64, dtype=torch.int32, device="cuda")
    t = F_._with_tiles(base.view(10, 64, C), (stamps, 5))
    assert F_._tiles_of(t, rows) is not None
    assert F_._tiles_of(t.transpose(0, 1).transpose(0, 1), rows) is not None      # the layout seam: transposed and back
    assert F_._tiles_of(t.transpose(0, 1), rows) is None                          # another row order
    assert F_._tiles_of(t.clone(), rows) is None                                  # another tensor
    assert F_._tiles_of(t, rows + 64) is None
    t.add_(1.0)                                                                   # in-place accumulation
    assert F_._tiles_of(t, rows) is None and F_._tiles_of(t.transpose(0, 1).transpose(0, 1), rows) is None


# --------------------------------------------------------------------------------------------
# cst_embed_pos_fwd / cst_embed_bwd / cst_dropout_scale against the oracle's make_positions / sinusoidal table
# (restated from modules/sinusoidal_positional_embedding.py:36-105, utils.py:235-245) and F.embedding's gradient
# --------------------------------------------------------------------------------------------
def _tokens(B, T, V, pad, seed, lens):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(4, V, (B, T), generator=g)
    for b, l in enumerate(lens):
        t[b, l:] = pad
    t[0, 3] = pad  # a pad INSIDE a sequence: make_positions does not count it and gives it the pad position
    t[:, 1] = t[:, 0]  # repeated symbols: several occurrences feed one row of the table gradient
    return t


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,C,V", [(3, 7, 64, 60), (32, 128, 512, 10000), (5, 300, 1024, 997)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_embed_positions_token_path(K, dt, B, T, C, V, p):
    from oracle import chimera_oracle as O
    k, L = K
    pad, scale, key = 1, math.sqrt(C), 0xBEEF01
    lens = [T - (3 * b) % (T - 5) for b in range(B)]
    tok = _tokens(B, T, V, pad, 5, lens)
    E = rnd(V, C, dt=dt, seed=6, scale=C ** -0.5)
    E[pad].zero_()
    table = O.sinusoidal_table(pad + 1 + T + 9, C, pad).cuda()
    out = k.embed_pos_fwd(tok.cuda(), None, E, None, table, scale, pad, p, key)
    ref = scale * E.float()[tok.cuda()] + O.positional_embedding(tok, C, pad).cuda()
    keep = _keep(key, B * T * C, p).view(B, T, C).cuda() if p > 0 else torch.ones(B, T, C, dtype=torch.bool, device="cuda")
    ref = torch.where(keep, ref / (1 - p), torch.zeros_like(ref))
    check(out, ref, dt, "embed_pos fwd")
    assert float(out[0, 3].float().abs().max()) == 0.0  # pad symbol: zero embedding row + the zero pad row of the table
    # table gradient: deterministic, pad row untouched, equals scale * index_add of the (masked) upstream gradient
    dy = rnd(B, T, C, dt=dt, seed=7)
    dE = k.embed_bwd(dy, tok.cuda().view(-1), V, scale, pad, p, key, dt)
    dE2 = k.embed_bwd(dy, tok.cuda().view(-1), V, scale, pad, p, key, dt)
    assert torch.equal(dE, dE2)
    g = torch.where(keep, dy.float() / (1 - p), torch.zeros_like(dy.float())) * scale
    want = torch.zeros(V, C, device="cuda").index_add_(0, tok.cuda().view(-1), g.view(-1, C))
    want[pad].zero_()
    check(dE, want, dt, "embed bwd")
    assert float(dE[pad].float().abs().max()) == 0.0
    unused = torch.ones(V, dtype=torch.bool)
    unused[tok.view(-1)] = False
    assert float(dE[unused.cuda()].float().abs().max() if unused.any() else 0.0) == 0.0
    dE32 = k.embed_bwd(dy, tok.cuda().view(-1), V, scale, pad, p, key, torch.float32)  # fp32 accumulation buffer variant
    check(dE32, want, torch.float32 if dt == torch.float32 else dt, "embed bwd fp32 out")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("with_pos", [True, False])
def test_embed_positions_dense_path_and_mask_source(K, dt, with_pos):
    """The encoders' form: dense features scaled by sqrt(d), positions from the 0/1 padding mask (`mask.ne(padding_idx)`)."""
    from oracle import chimera_oracle as O
    k, L = K
    B, T, C, pad, p, key = 4, 93, 512, 1, 0.2, 4242
    lens = torch.tensor([93, 80, 41, 7])
    mask = torch.arange(T)[None, :] >= lens[:, None]
    x = rnd(B, T, C, dt=dt, seed=2)
    table = O.sinusoidal_table(pad + 1 + T, C, pad).cuda() if with_pos else None
    scale = math.sqrt(C)
    out = k.embed_pos_fwd(None, mask.to(torch.uint8).cuda() if with_pos else None, None, x, table, scale, pad, p, key)
    ref = scale * x.float() + (O.positional_embedding(mask, C, pad).cuda() if with_pos else 0.0)
    keep = _keep(key, B * T * C, p).view(B, T, C).cuda()
    check(out, torch.where(keep, ref / (1 - p), torch.zeros_like(ref)), dt, "embed_pos dense fwd")
    dy = rnd(B, T, C, dt=dt, seed=3)
    dx = k.dropout_scale(dy, scale, p, key)
    check(dx, torch.where(keep, dy.float() * scale / (1 - p), torch.zeros_like(dy.float())), dt, "dropout_scale")
    # token ids with a mask as the position source (the Chimera text encoder): embedding from the ids, positions from the mask
    if with_pos:
        tok = torch.randint(4, 50, (B, T), generator=torch.Generator().manual_seed(1))
        E = rnd(50, C, dt=dt, seed=9)
        out = k.embed_pos_fwd(tok.cuda(), mask.to(torch.uint8).cuda(), E, None, table, scale, pad, 0.0, 0)
        check(out, scale * E.float()[tok.cuda()] + O.positional_embedding(mask, C, pad).cuda(), dt, "embed_pos ids + mask")


# --------------------------------------------------------------------------------------------
# packed (padding-free) rows: cst_rows_pack / cst_rows_unpack and packed self-attention against the dense padded calls
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
def test_rows_pack_unpack(K, dt):
    k, L = K
    B, T, C = 5, 77, 136
    n = torch.tensor([77, 60, 33, 77, 1], dtype=torch.int32)
    off = torch.zeros(B + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(n, 0)
    rows = int(off[-1])
    x = rnd(B, T, C, dt=dt, seed=1)
    offd = off.cuda()
    packed = k.rows_pack(x, offd, rows, tail_sum=False)
    for b in range(B):
        assert torch.equal(packed[off[b]:off[b + 1]], x[b, :n[b]])
    back = k.rows_unpack(packed, offd, B, T, tail_broadcast=True)
    back0 = k.rows_unpack(packed, offd, B, T, tail_broadcast=False)
    for b in range(B):
        nb = int(n[b])
        assert torch.equal(back[b, :nb], x[b, :nb]) and torch.equal(back0[b, :nb], x[b, :nb])
        if nb < T:
            assert torch.equal(back[b, nb:], x[b, nb - 1].expand(T - nb, C)) and float(back0[b, nb:].float().abs().max()) == 0.0
    g = rnd(B, T, C, dt=dt, seed=2)
    gs = k.rows_pack(g, offd, rows, tail_sum=True)
    assert torch.equal(gs, k.rows_pack(g, offd, rows, tail_sum=True))  # fixed summation order
    for b in range(B):
        nb = int(n[b])
        assert torch.equal(gs[off[b]:off[b] + nb - 1], g[b, :nb - 1])
        check(gs[off[b] + nb - 1], g[b, nb - 1:].float().sum(0), dt, "tail sum b=%d" % b)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("p", [0.0, 0.15])
def test_attention_packed_rows_equal_the_padded_call_bit_for_bit(K, dt, p):
    """Self-attention over packed rows (cst_attn_desc.seq_offsets: queries = the kept rows of each sequence, keys = its real frames)
    against the dense call with a key padding mask on the same data: identical bits for the kept rows — outputs, lse, and the
    three gradients — including the dropout masks (the packed call indexes the mask space of the padded one)."""
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    k, L = K
    B, T, H, D = 4, 300, 3, 64
    C = H * D
    lens = torch.tensor([300, 251, 130, 17])
    margin = 9
    pm = (torch.arange(T)[None, :] >= lens[:, None]).cuda()
    plan = CF.plan_packed_rows(pm, margin)
    assert plan.longest == 300 and plan.rows == int(torch.clamp(lens + margin + 1, max=T).sum())
    qkv = rnd(B, T, 3 * C, dt=dt, seed=11)
    do = rnd(B, T, C, dt=dt, seed=12)
    key = 777
    # dense padded call
    q, kk, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    kpm = pm.to(torch.uint8).contiguous()
    o = torch.empty(B, T, C, dtype=dt, device="cuda"); lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
    d = k.attn_desc(q, kk, v, o, lse, H, D, kpm, False, D ** -0.5, "bt", "bt", p, key, None)
    k.attn_fwd_desc(d)
    dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    k.attn_bwd_fill(d, do, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], delta, D)
    k.attn_bwd_desc(d)
    # packed call on the kept rows
    off = plan.offsets
    pq = k.rows_pack(qkv, off, plan.rows, False).unsqueeze(0)
    pdo = k.rows_pack(do, off, plan.rows, False).unsqueeze(0)
    po = torch.empty(1, plan.rows, C, dtype=dt, device="cuda"); plse = torch.empty(B, H, plan.longest, dtype=torch.float32, device="cuda")
    d2 = k.attn_desc(pq[..., :C], pq[..., C:2 * C], pq[..., 2 * C:], po, plse, H, D, None, False, D ** -0.5, "bt", "bt", p, key, plan.kv_len, (off, plan.longest))
    k.attn_fwd_desc(d2)
    pdqkv = torch.empty_like(pq); pdelta = torch.empty_like(plse)
    k.attn_bwd_fill(d2, pdo, pdqkv[..., :C], pdqkv[..., C:2 * C], pdqkv[..., 2 * C:], pdelta, D)
    k.attn_bwd_desc(d2)
    offh = off.cpu()
    for b in range(B):
        nb, lb = int(offh[b + 1] - offh[b]), int(lens[b])
        sl = slice(int(offh[b]), int(offh[b + 1]))
        assert torch.equal(po[0, sl], o[b, :nb]), "output rows of sequence %d" % b
        assert torch.equal(plse[b, :, :nb], lse[b, :, :nb])
        # gradients: the dense call also receives dO on the rows the packed call dropped, which changes dK / dV of the real frames;
        # compare against a dense backward whose dO is zero behind the kept rows
    do_kept = do.clone()
    for b in range(B):
        do_kept[b, int(offh[b + 1] - offh[b]):] = 0
    dqkv2 = torch.empty_like(qkv)
    k.attn_bwd_fill(d, do_kept, dqkv2[..., :C], dqkv2[..., C:2 * C], dqkv2[..., 2 * C:], delta, D)
    k.attn_bwd_desc(d)
    for b in range(B):
        nb = int(offh[b + 1] - offh[b])
        sl = slice(int(offh[b]), int(offh[b + 1]))
        assert torch.equal(pdqkv[0, sl], dqkv2[b, :nb]), "gradient rows of sequence %d" % b


def test_gemm_reserved_cus_is_a_pure_speed_switch(K):
    """cst_gemm_reserve_cus (CUs the persistent GEMM leaves to a concurrent all-reduce): a smaller persistent grid walks the same work
    items — identical bits — and the call returns the previous setting."""
    k, L = K
    lib = L.load()
    M, N, Kd = 4096, 768, 1024
    A, B = rnd(M, Kd, dt=torch.bfloat16, seed=1), rnd(N, Kd, dt=torch.bfloat16, seed=2)
    bias = rnd(N, dt=torch.bfloat16, seed=3)
    outs = []
    assert lib.cst_gemm_reserve_cus(-1) == 0
    for n in (0, 64, 200):
        prev = lib.cst_gemm_reserve_cus(n)
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        k.gemm(A, B, C, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, bias=bias, act=L.ACT_GELU)
        outs.append(C)
        assert lib.cst_gemm_reserve_cus(-1) == n and prev in (0, 64)
    lib.cst_gemm_reserve_cus(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(768, 48, 128), (64, 16, 16), (33, 7, 8)])
def test_weight_norm_last_dim(K, dt, shape):
    """cst_weight_norm_fwd/bwd against nn.utils.weight_norm(conv, dim=2) semantics (wav2vec2.py:773-779): w = v g / ||v||, one norm per
    kernel tap; forward and both gradients, and the same bits on a second call (fixed summation order)."""
    k, L = K
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    v = rnd(*shape, dt=dt, seed=1).requires_grad_(True)
    g = (1.0 + 0.1 * rnd(1, 1, shape[-1], dt=torch.float32, seed=2)).to(dt).requires_grad_(True)
    dw = rnd(*shape, dt=dt, seed=3)
    w = CF.weight_norm_last_dim(v, g)
    w.backward(dw)
    vr, gr = v.detach().float().requires_grad_(True), g.detach().float().requires_grad_(True)
    wr = vr * (gr / vr.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
    wr.backward(dw.float())
    check(w, wr, dt, "weight_norm fwd")
    check(v.grad, vr.grad, dt, "weight_norm dv", scale=float(vr.grad.abs().max()))
    check(g.grad, gr.grad, dt, "weight_norm dg", scale=float(gr.grad.abs().max()))
    v2, g2 = v.detach().clone().requires_grad_(True), g.detach().clone().requires_grad_(True)
    w2 = CF.weight_norm_last_dim(v2, g2)
    w2.backward(dw)
    assert torch.equal(w, w2) and torch.equal(v.grad, v2.grad) and torch.equal(g.grad, g2.grad)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(3072, 768), (768, 2304), (64, 64), (72, 136), (8, 264), (336, 8)])
def test_transpose2d_is_exact(K, dt, shape):
    """cst_transpose2d (the W^T copy the dX GEMMs read k-major): pure data movement, bit-exact incl. partial 64 x 64 tiles; sides that
    are no multiple of 8 are refused (functional._want_wt never asks for them)."""
    k, L = K
    x = torch.randn(*shape, device="cuda").to(dt)
    assert torch.equal(k.transpose2d(x), x.t().contiguous())
    with pytest.raises(RuntimeError, match="multiples of"):
        k.transpose2d(torch.zeros(65, 128, device="cuda", dtype=dt))


def test_dx_through_transposed_weight_equals_m_major_read():
    """functional._linear_backward reads W^T k-major above CST_WT_MIN_ROWS rows; the GEMM accumulates the same products in the same
    K order either way, so dX has the same bits as the m-major read it replaces."""
    import os
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    torch.manual_seed(5)
    x = torch.randn(4608, 768, device="cuda").bfloat16().requires_grad_(True)
    w = (torch.randn(1536, 768, device="cuda") / 28).bfloat16().requires_grad_(True)
    dy = torch.randn(4608, 1536, device="cuda").bfloat16()
    grads = []
    for no_wt in ("", "1"):
        os.environ["CST_NO_WT"] = no_wt
        try:
            y = CF.linear(x, w, None)
            gx, gw = torch.autograd.grad(y, (x, w), dy)
        finally:
            os.environ.pop("CST_NO_WT", None)
        grads.append((gx, gw))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


def test_weight_transposes_refresh_in_one_launch_after_an_optimizer_step(K):
    """functional.WEIGHT_TRANSPOSES: the W^T copies survive across backward passes, every copy is refreshed by ONE multi-matrix launch
    after a fused optimizer step (raw-pointer update: no autograd version bump), a torch in-place write re-transposes that weight
    alone, and the copies always equal W^T bit for bit."""
    from importlib import import_module
    k, L = K
    CF = import_module("chimera-st_amd.functional")
    optim = import_module("chimera-st_amd.optim")
    cache = CF._WeightTransposes()
    ws = [torch.nn.Parameter(torch.randn(r, c, device="cuda").bfloat16(), requires_grad=False)
          for r, c in ((768, 2304), (3072, 768), (72, 136), (512, 512))]
    for w in ws:
        assert torch.equal(cache.get(w), w.t().contiguous())
    assert cache.refreshes == 0 and len(cache.entries) == 4
    tmp = torch.randn(256, 512, device="cuda").bfloat16()  # a temporary (torch.cat fallback, padded weight): transposed, never cached
    assert torch.equal(cache.get(tmp), tmp.t().contiguous()) and len(cache.entries) == 4
    first = [cache.get(w) for w in ws]
    assert all(a is cache.get(w) for a, w in zip(first, ws))  # cached: no launch, same tensor
    # a fused-optimizer-style update: the library writes the storage through raw pointers (no autograd version bump), epoch + 1
    for w in ws:
        vers = w._version
        k.transpose2d(torch.randn(w.shape[1], w.shape[0], device="cuda").to(w.dtype), out=w)
        assert w._version == vers
        assert not torch.equal(cache.entries[(w.data_ptr(), tuple(w.shape), w.dtype)][1], w.t().contiguous())  # stale until asked for
    optim.PARAM_EPOCH[0] += 1
    got = cache.get(ws[2])
    assert cache.refreshes == 1
    for w in ws:
        assert torch.equal(cache.get(w), w.t().contiguous())
    assert cache.refreshes == 1 and got is first[2]
    ws[1].mul_(2.0)  # a torch op: version bump, this weight alone
    assert torch.equal(cache.get(ws[1]), ws[1].t().contiguous()) and cache.refreshes == 1
    # a dead Parameter does not pin its copy: the entry goes with the next whole-table refresh
    key3 = (ws[3].data_ptr(), tuple(ws[3].shape), ws[3].dtype)
    del first, got, w
    ws.pop()
    optim.PARAM_EPOCH[0] += 1
    cache.get(ws[0])
    assert key3 not in cache.entries and len(cache.entries) == 3


def test_weight_transposes_follow_the_flat_buffer_and_the_stacked_view(K):
    """Views of a FlatParamBuffers storage are cached while the buffers live; the stacked q | k | v view notices an in-place edit of k
    or v alone (it shares only q's version counter); a restore through FusedAdam.load_state_dict (a copy into the flat buffer: no
    per-parameter version bump) refreshes the copies; dropping the buffers drops the entries."""
    import gc
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    optim = import_module("chimera-st_amd.optim")

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attention = True
            self.k_proj, self.v_proj, self.q_proj = (torch.nn.Linear(64, 64) for _ in range(3))

    net = Attn().cuda().bfloat16()
    buf = optim.FlatParamBuffers(net.parameters(), adjacent=optim.qkv_groups(net))
    cache = CF.WEIGHT_TRANSPOSES
    cache.invalidate()
    cat = lambda: torch.cat((net.q_proj.weight, net.k_proj.weight, net.v_proj.weight), 0).detach()
    w = CF.stacked_rows(net.q_proj.weight, net.k_proj.weight, net.v_proj.weight)
    assert w.data_ptr() == net.q_proj.weight.data_ptr()
    assert torch.equal(cache.get(w), cat().t().contiguous()) and len(cache.entries) == 1
    with torch.no_grad():
        net.v_proj.weight.mul_(3.0)  # v alone: q's version counter does not move
    w = CF.stacked_rows(net.q_proj.weight, net.k_proj.weight, net.v_proj.weight)
    assert torch.equal(cache.get(w), cat().t().contiguous())
    opt = optim.FusedAdam(None, buffers=buf)
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()}
    sd["master"].mul_(0.5)
    opt.load_state_dict(sd)
    w = CF.stacked_rows(net.q_proj.weight, net.k_proj.weight, net.v_proj.weight)
    assert torch.equal(cache.get(w), cat().t().contiguous())
    del w, opt, buf, net
    gc.collect()
    assert len(cache.entries) == 0


# --------------------------------------------------------------------------------------------
# The kernels bench.py times, at the bench's shapes, against fp32 torch (round 4: VERDICT "what's weak" #1)
# --------------------------------------------------------------------------------------------
def test_gemm8p_at_the_bench_shape_with_its_epilogues(K):
    """The persistent 256 x 256 kernel at the wav2vec2 FFN shape of the step (31 760 packed rows): fc1 forward with bias + GELU +
    pre-activation output (MODE 1), fc2 forward with bias + residual (MODE 2), the dX GEMM with the GELU' epilogue (MODE 2) and a
    row-limited launch (m_live: dead 256-row tiles) — every output element against fp32 torch on the same bf16 inputs."""
    k, L = K
    dt = torch.bfloat16
    M, N, K_ = 31760, 3072, 768
    x, w1, b1 = rnd(M, K_, dt=dt, seed=1), rnd(N, K_, dt=dt, seed=2, scale=K_ ** -0.5), rnd(N, dt=dt, seed=3)
    h, z = torch.empty(M, N, dtype=dt, device="cuda"), torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(x, w1, h, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, bias=b1, act=L.ACT_GELU, aux_out=z, ld_aux_out=N, split_k=1)
    zr = x.float() @ w1.float().t() + b1.float()
    check(z, zr, dt, "fc1 pre-activation")
    check(h, F.gelu(zr), dt, "fc1 bias + GELU")
    del zr
    w2, b2, res = rnd(K_, N, dt=dt, seed=4, scale=N ** -0.5), rnd(K_, dt=dt, seed=5), rnd(M, K_, dt=dt, seed=6)
    y = torch.empty(M, K_, dtype=dt, device="cuda")
    k.gemm(h, w2, y, M, K_, N, a_kmajor=1, b_kmajor=1, lda=N, ldb=N, ldc=K_, bias=b2, resid=res, ld_resid=K_, split_k=1)
    check(y, h.float() @ w2.float().t() + b2.float() + res.float(), dt, "fc2 bias + residual")
    # dz1 = (dy W2) * GELU'(z1) through the transposed-weight copy (both operands k-major, as functional._FFNFn.backward launches it)
    dy = rnd(M, K_, dt=dt, seed=7)
    w2t = w2.t().contiguous()  # [N, K_] = W2^T: dz1[m, n] = sum_c dy[m, c] W2[c, n]
    dz = torch.empty(M, N, dtype=dt, device="cuda")
    k.gemm(dy, w2t, dz, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, dact=L.ACT_GELU, aux_in=z, ld_aux_in=N, split_k=1)
    zz = z.float()
    dgelu = 0.5 * (1 + torch.erf(zz / math.sqrt(2))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2 * math.pi)
    check(dz, (dy.float() @ w2.float()) * dgelu, dt, "fc1 dX * GELU'")
    del zz, dgelu


@pytest.mark.parametrize("packed", [False, True], ids=["padded+kv_len", "packed"])
def test_attention_dma_staged_kernels_at_the_bench_sequence_length(K, packed):
    """fa_fwd / fa_dq / fa_dkv (bf16, head dim 64: csrc/attention_fast.inc) at T = 1499 — 24 key tiles per sequence, ragged lengths,
    12 heads — against the fp32 torch restatement of modules/multihead_attention.py:326-361: the padded form (key-padding words +
    kv_len) and the packed form the wav2vec2 stack runs (seq_offsets: row offsets beyond 2 k, queries = kept rows, keys = real frames)."""
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    k, L = K
    dt = torch.bfloat16
    B, T, H, D = 4, 1499, 12, 64
    C = H * D
    lens = torch.tensor([1499, 1203, 850, 501])
    pm = (torch.arange(T)[None, :] >= lens[:, None]).cuda()
    kpm = pm.to(torch.uint8).contiguous()
    qkv = rnd(B, T, 3 * C, dt=dt, seed=21)
    do = rnd(B, T, C, dt=dt, seed=22)
    plan = CF.plan_packed_rows(pm, 6)
    offh = plan.offsets.cpu()
    kept = [int(offh[b + 1] - offh[b]) for b in range(B)]
    do_kept = do.clone()
    for b in range(B):
        do_kept[b, kept[b]:] = 0  # the packed form has no rows behind the kept ones: the reference gets zero gradient there
    scale = D ** -0.5
    qr, kr, vr = (qkv[..., i * C:(i + 1) * C].float().detach().requires_grad_(True) for i in range(3))
    ref = attn_ref(qr, kr, vr, H, kpm, False, scale)
    ref.backward(do_kept.float())
    if not packed:
        q, kk, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        o = torch.empty(B, T, C, dtype=dt, device="cuda"); lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
        d = k.attn_desc(q, kk, v, o, lse, H, D, kpm, False, scale, "bt", "bt", 0.0, 0, lens.to(torch.int32).cuda())
        k.attn_fwd_desc(d)
        dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
        k.attn_bwd_fill(d, do_kept, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], delta, D)
        k.attn_bwd_desc(d)
        for b in range(B):
            n = kept[b]
            check(o[b, :n], ref[b, :n], dt, "o[%d]" % b)
            check(dqkv[b, :n, :C], qr.grad[b, :n], dt, "dq[%d]" % b)
            check(dqkv[b, :n, C:2 * C], kr.grad[b, :n], dt, "dk[%d]" % b)
            check(dqkv[b, :n, 2 * C:], vr.grad[b, :n], dt, "dv[%d]" % b)
        return
    assert plan.rows > 2048 and plan.longest == T
    pq = k.rows_pack(qkv, plan.offsets, plan.rows, False).unsqueeze(0)
    pdo = k.rows_pack(do_kept, plan.offsets, plan.rows, False).unsqueeze(0)
    po = torch.empty(1, plan.rows, C, dtype=dt, device="cuda"); plse = torch.empty(B, H, plan.longest, dtype=torch.float32, device="cuda")
    d2 = k.attn_desc(pq[..., :C], pq[..., C:2 * C], pq[..., 2 * C:], po, plse, H, D, None, False, scale, "bt", "bt", 0.0, 0, plan.kv_len, (plan.offsets, plan.longest))
    k.attn_fwd_desc(d2)
    pdqkv = torch.empty_like(pq); pdelta = torch.empty_like(plse)
    k.attn_bwd_fill(d2, pdo, pdqkv[..., :C], pdqkv[..., C:2 * C], pdqkv[..., 2 * C:], pdelta, D)
    k.attn_bwd_desc(d2)
    for b in range(B):
        sl = slice(int(offh[b]), int(offh[b + 1]))
        n = kept[b]
        check(po[0, sl], ref[b, :n], dt, "packed o[%d]" % b)
        check(pdqkv[0, sl, :C], qr.grad[b, :n], dt, "packed dq[%d]" % b)
        check(pdqkv[0, sl, C:2 * C], kr.grad[b, :n], dt, "packed dk[%d]" % b)
        check(pdqkv[0, sl, 2 * C:], vr.grad[b, :n], dt, "packed dv[%d]" % b)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n_out,k_in,tokens", [(512, 512, 4064), (1536, 512, 7901), (512, 2048, 7901), (768, 3072, 31760), (72, 136, 300)])
def test_gemm_colsum_is_the_bias_gradient_next_to_dw(K, dt, n_out, k_in, tokens):
    """cst_gemm_desc.colsum on the weight-gradient layout (both operands mn-major): dW = dY^T X and db = column sums of dY from ONE
    call — a by-product of the 4-wave kernels (with and without split-K, with live-tile stamps), the separate two-launch column sum
    behind the 16-wave kernel — against fp32 torch; bit-reproducible; dW has the bits of the launch without the by-product."""
    k, L = K
    if dt == torch.float32 and tokens > 20000:
        pytest.skip("bf16-only shape")
    dy = rnd(tokens, n_out, dt=dt, seed=61)
    dy[tokens // 2: tokens // 2 + 700] = 0  # dead rows (padded frames)
    x = rnd(tokens, k_in, dt=dt, seed=62)
    fused = k.dw_colsum_is_fused(n_out, k_in, tokens, dt)
    if dt == torch.bfloat16:  # (the 16-wave configuration takes the wav2vec2-sized products only; fp32 splits K differently)
        assert fused == (n_out < 256 or k_in < 256 or tokens < 20000)
    outs = []
    for rep in range(2):
        dw, db = torch.empty(n_out, k_in, dtype=dt, device="cuda"), torch.empty(n_out, dtype=dt, device="cuda")
        k.gemm(dy, x, dw, n_out, k_in, tokens, a_kmajor=0, b_kmajor=0, lda=n_out, ldb=k_in, ldc=k_in, split_k=-1, colsum=db)
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dw0 = torch.empty(n_out, k_in, dtype=dt, device="cuda")
    k.gemm(dy, x, dw0, n_out, k_in, tokens, a_kmajor=0, b_kmajor=0, lda=n_out, ldb=k_in, ldc=k_in, split_k=-1)
    assert torch.equal(dw0, outs[0][0])
    ref = dy.float().sum(0)
    check(outs[0][1], ref, dt, "db", scale=float(ref.abs().max()))
    check(outs[0][0], dy.float().t() @ x.float(), dt, "dW")
    if tokens % 64 == 0 or True:
        # live-tile stamps (the dead rows' 64-row blocks skipped): same sums
        rows64 = (tokens + 63) // 64
        live_rows = (dy.float().abs().sum(1) > 0)
        pad = rows64 * 64 - tokens
        lr = torch.cat([live_rows, torch.zeros(pad, dtype=torch.bool, device="cuda")]).view(rows64, 64).any(1)
        epoch = 0x1234567
        stamps = torch.where(lr, torch.full((rows64,), epoch, dtype=torch.int32, device="cuda"), torch.zeros(rows64, dtype=torch.int32, device="cuda"))
        dw2, db2 = torch.empty(n_out, k_in, dtype=dt, device="cuda"), torch.empty(n_out, dtype=dt, device="cuda")
        k.gemm(dy, x, dw2, n_out, k_in, tokens, a_kmajor=0, b_kmajor=0, lda=n_out, ldb=k_in, ldc=k_in, split_k=-1, k_live=(stamps, epoch), colsum=db2)
        check(db2, ref, dt, "db with stamps", scale=float(ref.abs().max()))
        check(dw2, dy.float().t() @ x.float(), dt, "dW with stamps")


def test_reduce_multi_finishes_many_reductions_in_one_launch(K):
    """cst_reduce_multi: dst[i] = sum_p src[p * stride + i] for a table of items of different lengths, partial counts, strides and
    output types in ONE launch (more than 64 items: two launches), against fp64 sums; run twice: the same bits."""
    k, L = K
    torch.manual_seed(3)
    specs = [(768, 1024, 2 * 768, torch.bfloat16), (768, 1024, 2 * 768, torch.bfloat16), (512 * 512, 15, 512 * 512, torch.bfloat16),
             (512, 15, 512, torch.bfloat16), (8, 1, 8, torch.float32), (2048 * 512, 4, 2048 * 512, torch.float32), (1000 * 8, 7, 9000, torch.bfloat16)]
    specs = specs + [(64 * (i + 1), 3 + i % 9, 64 * (i + 1) + 8, torch.bfloat16) for i in range(70)]
    srcs, dsts, items = [], [], []
    for n, (Lr, P, stride, dt) in enumerate(specs):
        src = torch.randn(P * stride + 64, device="cuda")
        dst = torch.full((Lr,), float("nan"), dtype=dt, device="cuda")
        srcs.append(src); dsts.append(dst)
        items.append((src.data_ptr(), dst.data_ptr(), stride, Lr, P, L.dtype_code(dt), 1 if n < 2 or n % 3 == 0 else 0))
    k.reduce_multi(items)
    first = [d.clone() for d in dsts]
    k.reduce_multi(items)
    for (Lr, P, stride, dt), src, dst, f in zip(specs, srcs, dsts, first):
        ref = torch.as_strided(src, (P, Lr), (stride, 1)).double().sum(0)
        check(dst, ref.float(), dt, "reduce_multi L=%d P=%d" % (Lr, P), scale=float(ref.abs().max()))
        assert torch.equal(dst, f)


def test_deferred_reductions_give_the_gradients_of_the_immediate_route():
    """kernels.DEFER (split-K slabs of the small weight-gradient GEMMs + their bias-gradient slices + LayerNorm dgamma / dbeta
    partials finished by one cst_reduce_multi launch at the end of the backward pass): the same gradients as the launch-each route to
    fp32 summation order, every deferred destination written (the suite runs with CST_DEFER_POISON=1: an unwritten one is NaN),
    nothing deferred for a parameter that is shared, has a gradient already, or outside the mode."""
    from importlib import import_module
    load_pkg()
    CF = import_module("chimera-st_amd.functional")
    Kk = import_module("chimera-st_amd.kernels")
    torch.manual_seed(0)
    dt = torch.bfloat16
    lin1, lin2 = torch.nn.Linear(512, 2048).cuda().to(dt), torch.nn.Linear(2048, 512).cuda().to(dt)
    ln = torch.nn.LayerNorm(512).cuda().to(dt)
    proj = torch.nn.Linear(512, 512).cuda().to(dt)
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(ln.parameters()) + list(proj.parameters())
    x = torch.randn(4064, 512, device="cuda").to(dt)

    def run(defer):
        for p in params:
            p.grad = None
        with Kk.deferred_reductions(defer):
            h = CF.layer_norm(x, ln.weight, ln.bias)
            y = CF.ffn(h, lin1.weight, lin1.bias, lin2.weight, lin2.bias, "relu", resid=h)
            z = CF.linear(y, proj.weight, proj.bias)
            n0 = Kk.DEFER.flushes
            z.float().pow(2).sum().backward()
            pending = len(Kk.DEFER.items)
        return [p.grad.clone() for p in params], pending, Kk.DEFER.flushes - n0

    g0, pend0, fl0 = run(False)
    g1, pend1, fl1 = run(True)
    assert pend0 == 0 and fl0 == 0 and pend1 >= 6 and fl1 == 1  # 3 split weight gradients (+ bias slices) + dgamma / dbeta, one flush
    for a, b, p in zip(g0, g1, params):  # the same bits: cst_reduce_multi adds in the order of the kernel each item stands in for
        assert torch.isfinite(b.float()).all() and torch.equal(a, b), "deferred gradient %s" % (tuple(p.shape),)
    # a parameter that already holds a gradient (a later micro-batch) or is marked shared takes the immediate route
    proj.weight._cst_shared = True
    with Kk.deferred_reductions(True):
        h = CF.layer_norm(x, ln.weight, ln.bias)  # ln.weight.grad is not None: accumulation
        z = CF.linear(h, proj.weight, proj.bias)
        z.float().pow(2).sum().backward()
        assert len(Kk.DEFER.items) == 0
    assert all(torch.isfinite(p.grad.float()).all() for p in params)


@pytest.mark.parametrize("M,N,K_,epi", [(31760, 768, 3072, "resid"), (31760, 768, 768, "bias"), (31760, 2304, 768, "bias"), (24100, 768, 1024, "dact"),
                                        (31744 + 16, 768, 512, "plain")])
def test_gemm8p_half_height_tail_items_keep_the_bits(K, M, N, K_, epi):
    """The persistent kernel cuts the tiles of a thinly filled last round into two 128-row items (gemm8p.hip, 'Tail'): the K order per
    output element is unchanged, so C (and a pre-activation output) have the bits of the whole-tile walk (CST_GEMM8P_NO_HALVES=1) —
    on the N = 768 family of the wav2vec2 layers, with each epilogue kind, incl. a last tile that ends inside its upper half."""
    import os
    k, L = K
    dt = torch.bfloat16
    A, B = rnd(M, K_, dt=dt, seed=71), rnd(N, K_, dt=dt, seed=72, scale=K_ ** -0.5)
    kw = {}
    if epi == "resid":
        kw = dict(bias=rnd(N, dt=dt, seed=73), resid=rnd(M, N, dt=dt, seed=74), ld_resid=N, drop_p=0.1, drop_key=99)
    elif epi == "bias":
        kw = dict(bias=rnd(N, dt=dt, seed=73), act=L.ACT_GELU, aux_out=torch.empty(M, N, dtype=dt, device="cuda"), ld_aux_out=N)
    elif epi == "dact":
        kw = dict(dact=L.ACT_GELU, aux_in=rnd(M, N, dt=dt, seed=75), ld_aux_in=N)
    outs = []
    for no_halves in ("", "1"):
        if no_halves:
            os.environ["CST_GEMM8P_NO_HALVES"] = "1"
        try:
            C = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
            if "aux_out" in kw:
                kw["aux_out"] = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
            k.gemm(A, B, C, M, N, K_, a_kmajor=1, b_kmajor=1, lda=K_, ldb=K_, ldc=N, split_k=1, **kw)
            outs.append((C, kw.get("aux_out")))
        finally:
            os.environ.pop("CST_GEMM8P_NO_HALVES", None)
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])
    if epi == "plain":
        check(outs[0][0], A.float() @ B.float().t(), dt, "half-height tail items")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_attention_kv_packed_projection_equals_separate_k_and_v(K, dt, drop):
    """functional.attention_kv (k | v as the two channel halves of ONE projection; dk | dv written into one buffer) against
    functional.attention on separate k and v tensors: the same kernels on the same values through different strides — outputs and
    all gradients bit for bit, with a key padding mask and with probability dropout (the mask depends on (row, key) only)."""
    from importlib import import_module
    CF = import_module("chimera-st_amd.functional")
    rng = import_module("chimera-st_amd.rng")
    B, Tq, Tk, H, D = 3, 40, 150, 4, 64
    C = H * D
    q0, kv0 = rnd(B, Tq, C, dt=dt, seed=81), rnd(B, Tk, 2 * C, dt=dt, seed=82)
    do = rnd(B, Tq, C, dt=dt, seed=83)
    lens = torch.tensor([150, 97, 64])
    kpm = (torch.arange(Tk)[None] >= lens[:, None]).cuda()
    outs = []
    for packed in (True, False):
        rng.reseed(7)
        q = q0.clone().requires_grad_(True)
        kv = kv0.clone().requires_grad_(True)
        if packed:
            o = CF.attention_kv(q, kv, H, kpm, dropout_p=drop)
        else:
            k, v = kv[..., :C].contiguous(), kv[..., C:].contiguous()
            o = CF.attention(q, k, v, H, kpm, dropout_p=drop)
        gq, gkv = torch.autograd.grad(o, (q, kv), do)
        outs.append((o.detach(), gq, gkv))
    for a, b, name in zip(outs[0], outs[1], ("o", "dq", "dkv")):
        assert torch.equal(a, b), name


def test_four_wave_instantiations_of_the_dma_gemm_keep_the_bits():
    """The experiment-hook configurations big4 / big4n (gemm_glds_kernel with four waves and 128-row wave tiles: pinned issue order,
    per-wave epilogue, swapped accumulator layout — DESIGN 5.1, round 4) against whatever the dispatcher picks: plain bf16 stores of
    the same products in the same K order are the same bits, ragged M / N edges and a K tail included.  And gemm4w.hip (the lean
    kernel of the same shape: the N = 768 long-K launches of the step) against the persistent 8-wave kernel with bias + dropout + residual, residual
    + live-tile stamps, rows beyond M: the same bits.  (Own process: the hook's environment switch is read once per process.)"""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import importlib, os, sys, torch
sys.path.insert(0, %r)
K = importlib.import_module("chimera-st_amd.kernels")
for (m, n, k) in [(1000, 768, 512), (2304, 700, 1096), (513, 384, 200)]:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    outs = []
    for cfg in ("", "big4", "big4n"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        c = torch.full((m, n), float("nan"), device="cuda", dtype=torch.bfloat16)
        K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1)
        outs.append(c)
    ref = (a.float() @ w.float().t())
    assert torch.isfinite(outs[0].float()).all() and (outs[0].float() - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (m, n, k)
# gemm4w.hip against the persistent 8-wave kernel, with the step's epilogues and live-tile stamps: the same bits
L = importlib.import_module("chimera-st_amd.lib")
m, n, k = 24000, 768, 1536
a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
a[20000:] = 0
stamps = torch.zeros((m + 63) // 64, dtype=torch.int32, device="cuda"); stamps[: 20000 // 64 + 1] = 7
r = torch.randn(m, n, device="cuda").bfloat16(); b = torch.randn(n, device="cuda").bfloat16()
for kw in (dict(), dict(resid=r, ld_resid=n), dict(resid=r, ld_resid=n, bias=b, drop_p=0.1, drop_key=4242), dict(resid=r, ld_resid=n, m_live=(stamps, 7))):
    outs = []
    for cfg in ("8p", "4w"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        c = torch.full((m, n), float("nan"), device="cuda", dtype=torch.bfloat16)
        K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1, **kw)
        outs.append(c)
    assert torch.isfinite(outs[1].float()).all() and torch.equal(outs[0], outs[1]), sorted(kw)
print("ok")
""" % ROOT
    e = dict(os.environ, CST_GEMM_EXPERIMENT="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_gemm8p_row_limited_batches_deal_their_tile_groups_round_robin(K):
    """Batched launches with per-batch row limits (cst_gemm_desc.m_len: the conv stack) hand the m-tile groups of a batch to the XCDs
    round-robin instead of in contiguous runs (gemm8p.hip setup(): the tiles behind a limit are the tail of the batch's rows, and the
    contiguous runs gave all of them to the last XCDs).  The map must stay a bijection of the tiles: every output row below its
    batch's limit holds the product (implicit-GEMM operand: overlapping rows, as in the conv stack), every row behind it is either
    untouched by a live tile's store or zero, nothing is left unwritten among the live tiles — for row counts with and without a
    partial last group and group counts that are / are not multiples of 8."""
    k, L = K
    dt = torch.bfloat16
    # 19 full groups + a partial one; exactly 19 groups; 5 groups + a partial one (too few for the deal: batch z walks its tiles
    # rotated by z * ntiles / 8 positions instead)
    for Lout, nb in ((16384 + 256 * 4 * 3 + 130, 3), (256 * 4 * 19, 2), (256 * 4 * 5 + 100, 6)):
        Cin, Cout, taps, stride = 64, 512, 3, 2
        Lin = (Lout - 1) * stride + taps
        x = rnd(nb, Lin, Cin, dt=dt, seed=5)
        w = rnd(Cout, taps * Cin, dt=dt, seed=6, scale=(taps * Cin) ** -0.5)
        lim = torch.tensor([Lout, Lout // 3 + 77, 5, Lout - 300, Lout // 2, 1000][:nb], dtype=torch.int32, device="cuda")
        y = torch.full((nb, Lout, Cout), float("nan"), dtype=dt, device="cuda")
        k.gemm(x, w, y, Lout, Cout, taps * Cin, a_kmajor=1, b_kmajor=1, lda=stride * Cin, ldb=taps * Cin, ldc=Cout, batch0=nb,
               sa=(Lin * Cin, 0), sb=(0, 0), sc=(Lout * Cout, 0), split_k=1, m_len=lim)
        A = torch.as_strided(x, (nb, Lout, taps * Cin), (Lin * Cin, stride * Cin, 1)).float()
        ref = A @ w.float().t()
        for b in range(nb):
            n_live = int(lim[b])
            check(y[b, :n_live], ref[b, :n_live], dt, "rows below the limit, batch %d" % b)
            tail = y[b, (n_live + 255) // 256 * 256:]          # whole tiles behind the limit: zeros (the skipped K loop's epilogue)
            assert bool((tail == 0).all()), "rows of dead tiles: %d non-zero / NaN" % int((tail != 0).sum())
        assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("dt", DT)
def test_pos_conv_frame_limits_keep_every_bit(K, dt):
    """functional.pos_conv_gelu_residual with `lens` / `grad_rows` (round 5): the input is zero from lens[b] on (wav2vec2.py:820-821), so
    the output tiles whose windows lie in the padding skip their products and their epilogue on zero accumulators, x + GELU(bias), IS
    the value of those frames; the gradient of the output is zero from grad_rows[b] on (what pack_rows' backward leaves), which bounds
    the dX GEMM the same way.  Output, dx, dw and db have the bits of the unlimited call — utterances shorter than the convolution's
    reach, a full-length one and a zero-length one included."""
    CF = __import__("importlib").import_module("chimera-st_amd.functional")
    k, L = K
    B_, T, C, G, Kp = 5, 1100, 768, 16, 128
    lens = torch.tensor([1100, 700, 333, 40, 0], dtype=torch.int32, device="cuda")
    keep = torch.clamp(lens + 7, max=T)                      # rows a packing plan with margin 6 keeps
    t = torch.arange(T, device="cuda")[None, :, None]
    x0 = rnd(B_, T, C, dt=dt, seed=64, scale=0.5) * (t < lens[:, None, None]).to(dt)
    w0 = rnd(C, C // G, Kp, dt=dt, seed=65, scale=0.02)
    b0 = rnd(C, dt=dt, seed=66, scale=0.1)
    dy = rnd(B_, T, C, dt=dt, seed=67) * (t < keep[:, None, None]).to(dt)
    outs = []
    for limited in (False, True, "host"):   # "host": the K-block stamps from the plan's host integers (one pinned copy) instead of device ops
        x, w, bias = (v.clone().requires_grad_(True) for v in (x0, w0, b0))
        k.STATS.clear()
        y = CF.pos_conv_gelu_residual(x, w, bias, G, lens if limited else None, keep if limited else None,
                                      keep.tolist() if limited == "host" else None)
        y.backward(dy)
        outs.append((y.detach(), x.grad, w.grad, bias.grad))
        if limited:
            assert k.STATS.get("gemm_m_len", 0) == 2, dict(k.STATS)   # the forward GEMM and the dX GEMM took their limits
            assert k.STATS.get("gemm_k_live", 0) == 1, dict(k.STATS)  # ... and the weight-gradient GEMM its live K blocks
    for other in outs[1:]:
        for a, b, what in zip(outs[0], other, ("y", "dx", "dw", "db")):
            assert torch.isfinite(b.float()).all(), what
            assert torch.equal(a, b), "%s differs on %d elements" % (what, int((a != b).sum()))
