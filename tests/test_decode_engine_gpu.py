"""GPU tests of the device-resident decode loop (decode.hip + decode_engine.py):
  * cst_beam_init/step against the oracle's search loop (oracle.beam_search_with) on a synthetic next-token table at the
    real vocabulary size — bit-exact token ids, finalisation order, scores to fp32 rounding;
  * cst_dec_self_attn / cst_dec_embed against plain torch fp32 restatements, with a shuffled ancestry table;
  * the captured-graph engine against the reference's SequenceGenerator fixtures (bit-exact ids) and against the
    module-by-module mirror loop on an s2t_transformer with ragged encoder padding."""
import ctypes
import math
from argparse import Namespace
from importlib import import_module

import pytest
import torch

from conftest import golden_sample, load_golden, load_pkg
from test_model_gpu import assert_close, build_from_golden, to_cuda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def KL():
    load_pkg()
    return import_module("chimera-st_amd.kernels"), import_module("chimera-st_amd.lib")


def _beam_state(L, bsz, beam, V, max_len, min_len, dtype, logits, unk_penalty=0.0, len_penalty=1.0, temperature=1.0,
                pad=1, unk=3, eos=2):
    dev = "cuda"
    bbsz, L1, LT = bsz * beam, max_len + 1, max_len + 2
    z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=dev)
    st = dict(step=z(1, dt=torch.int32), num_remaining=z(1, dt=torch.int32), tokens=z(2, bbsz, LT, dt=torch.int64),
              scores=z(2, bbsz, L1, dt=torch.float32), anc=z(2, bbsz, L1, dt=torch.int32), ignore=z(bsz, beam, dt=torch.uint8),
              finished=z(bsz, dt=torch.uint8), nfinal=z(bsz, dt=torch.int32), fin_tokens=z(bsz, beam, L1, dt=torch.int64),
              fin_pos=z(bsz, beam, L1, dt=torch.float32), fin_score=z(bsz, beam, dt=torch.float32), fin_len=z(bsz, beam, dt=torch.int32))
    d = L.BeamDesc()
    d.dtype = L.dtype_code(dtype)
    d.bsz, d.beam, d.vocab, d.max_len = bsz, beam, V, max_len
    d.pad, d.unk, d.eos, d.min_len = pad, unk, eos, min_len
    d.unk_penalty, d.len_penalty, d.temperature, d.normalize_scores = unk_penalty, len_penalty, temperature, 1
    d.logits, d.ld_logits = logits.data_ptr(), logits.stride(0)
    d.step, d.tokens, d.scores, d.anc = (st[k].data_ptr() for k in ("step", "tokens", "scores", "anc"))
    d.cands_to_ignore, d.finished, d.nfinal = st["ignore"].data_ptr(), st["finished"].data_ptr(), st["nfinal"].data_ptr()
    d.num_remaining = st["num_remaining"].data_ptr()
    d.fin_tokens, d.fin_pos, d.fin_score, d.fin_len = (st[k].data_ptr() for k in ("fin_tokens", "fin_pos", "fin_score", "fin_len"))
    st["ws"] = torch.zeros(L.load().cst_beam_workspace(bsz, beam), dtype=torch.uint8, device=dev)
    d.workspace = st["ws"].data_ptr()
    return st, d


@pytest.mark.parametrize("beam,V,dtype", [(5, 10000, torch.float32), (1, 10000, torch.float32), (5, 10000, torch.bfloat16),
                                          (10, 997, torch.float32), (3, 61, torch.float32)])
def test_beam_step_matches_oracle_search(KL, beam, V, dtype):
    """A synthetic "decoder": logits of the next token depend on (sentence, step, last token) through a seeded table, with
    the eos logit boosted as the step grows so hypotheses finish at different steps.  The device loop (cst_beam_step) and
    the oracle's restatement of sequence_generator.py/search.py must produce the same finalized hypotheses."""
    k, L = KL
    from oracle import chimera_oracle as O
    bsz, max_len, min_len, NT = 6, 14, 2, 64  # NT = table rows: logits row = table[(sentence * 7 + step * 13 + token) % NT]
    g = torch.Generator().manual_seed(beam * 1000 + V)
    table = (torch.randn(NT, V, generator=g) * 2.0).to(dtype)
    eos_boost = torch.linspace(-2.0, 6.0, max_len + 1)
    Vp = (V + 7) // 8 * 8
    logits = torch.zeros(bsz * beam, Vp, dtype=dtype, device="cuda")
    st, d = _beam_state(L, bsz, beam, V, max_len, min_len, dtype, logits, unk_penalty=0.5, len_penalty=1.3)
    lib = L.load()
    table_d, boost_d = table.cuda(), eos_boost.cuda()
    L.check(lib.cst_beam_init(ctypes.byref(d), L.stream_ptr()), "cst_beam_init")
    sent = torch.arange(bsz, device="cuda").repeat_interleave(beam)
    for step in range(max_len + 1):
        last = st["tokens"][step & 1, :, step]
        row = (sent * 7 + step * 13 + last) % NT
        lg = table_d[row].float()
        lg[:, 2] += boost_d[step]
        logits[:, :V] = lg.to(dtype)
        L.check(lib.cst_beam_step(ctypes.byref(d), L.stream_ptr()), "cst_beam_step")
        if int(st["num_remaining"].item()) == 0:
            break
    assert int(st["num_remaining"].item()) == 0
    assert int(st["step"].item()) == step + 1

    def lp(b, tokens):
        s_ = tokens.size(1) - 1
        row = (b * 7 + s_ * 13 + tokens[:, -1]) % NT
        lg = table[row].float()
        lg[:, 2] += eos_boost[s_]
        return torch.log_softmax(lg.to(dtype).float(), dim=-1)

    ref = O.beam_search_with(lp, bsz, beam, max_len, min_len, unk_penalty=0.5, len_penalty=1.3)
    fin_len, nfinal = st["fin_len"].cpu(), st["nfinal"].cpu()
    for b in range(bsz):
        assert int(nfinal[b]) == len(ref[b]) == beam
        hyps = [(float(st["fin_score"][b, r]), st["fin_tokens"][b, r, :int(fin_len[b, r])].cpu(), st["fin_pos"][b, r, :int(fin_len[b, r])].cpu())
                for r in range(beam)]
        hyps.sort(key=lambda h: -h[0])
        for r in range(beam):
            if dtype == torch.float32:  # bf16-quantised logits can tie exactly; tied candidates have equal scores
                assert hyps[r][1].tolist() == ref[b][r]["tokens"].tolist(), (b, r)
            assert len(hyps[r][1]) == len(ref[b][r]["tokens"])
            assert abs(hyps[r][0] - ref[b][r]["score"]) < 2e-5 * max(1.0, abs(ref[b][r]["score"]))
            assert float((hyps[r][2] - ref[b][r]["positional_scores"]).abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,D", [(8, 64), (3, 32)])
def test_dec_self_attn_and_embed(KL, dtype, H, D):
    k, L = KL
    lib = L.load()
    rows, max_len, C = 37, 21, H * D
    L1, LT = max_len + 1, max_len + 2
    g = torch.Generator().manual_seed(H * 100 + D)
    kc = torch.randn(rows, H, L1, D, generator=g).to(dtype).cuda()  # head-major caches
    vc = torch.randn(rows, H, L1, D, generator=g).to(dtype).cuda()
    kc0, vc0 = kc.clone(), vc.clone()
    qkv = torch.randn(rows, 3 * C, generator=g).to(dtype).cuda()
    out = torch.zeros(rows, C, dtype=dtype, device="cuda")
    scale = D ** -0.5
    for s in (0, 1, 7, max_len):
        anc = torch.randint(0, rows, (2, rows, L1), generator=g).int().cuda()
        anc[s & 1, :, s] = torch.arange(rows, dtype=torch.int32, device="cuda")
        step = torch.tensor([s], dtype=torch.int32, device="cuda")
        kc.copy_(kc0); vc.copy_(vc0)
        L.check(lib.cst_dec_self_attn(L.ptr(qkv), L.ptr(kc), L.ptr(vc), L.ptr(anc), L.ptr(step), L.ptr(out), rows, H, D, max_len,
                                      scale, L.dtype_code(dtype), L.stream_ptr()), "cst_dec_self_attn")
        # reference: gather the ancestry rows, append the new k/v, plain softmax attention in fp32
        a = anc[s & 1].long()
        pos = torch.arange(s + 1, device="cuda")
        Kr = kc0[a[:, :s + 1], :, pos].float().reshape(rows, s + 1, C)  # [rows, s+1, H, D] -> [rows, s+1, C]
        Vr = vc0[a[:, :s + 1], :, pos].float().reshape(rows, s + 1, C)
        Kr[:, s] = qkv[:, C:2 * C].float()
        Vr[:, s] = qkv[:, 2 * C:].float()
        q = qkv[:, :C].float().view(rows, H, D)
        sc = torch.einsum("rhd,rjhd->rhj", q, Kr.view(rows, s + 1, H, D)) * scale
        ref = torch.einsum("rhj,rjhd->rhd", torch.softmax(sc, -1), Vr.view(rows, s + 1, H, D)).reshape(rows, C)
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), s
        # the new key / value landed in slot s of the hypothesis' own row; nothing else was touched
        assert torch.equal(kc[:, :, s].reshape(rows, C), qkv[:, C:2 * C]) and torch.equal(vc[:, :, s].reshape(rows, C), qkv[:, 2 * C:])
        m = torch.ones(L1, dtype=torch.bool); m[s] = False
        assert torch.equal(kc[:, :, m], kc0[:, :, m]) and torch.equal(vc[:, :, m], vc0[:, :, m])
    # embed: scale * E[token at the step] + P[pad + 1 + step]
    V, pad = 50, 1
    E = torch.randn(V, C, generator=g).to(dtype).cuda()
    P = torch.randn(max_len + 4, C, generator=g).cuda()
    tokens = torch.randint(0, V, (2, rows, LT), generator=g).cuda()
    x = torch.zeros(rows, C, dtype=dtype, device="cuda")
    for s in (0, 5, max_len):
        step = torch.tensor([s], dtype=torch.int32, device="cuda")
        L.check(lib.cst_dec_embed(L.ptr(tokens), L.ptr(step), L.ptr(E), L.ptr(P), 3.25, pad, L.ptr(x), rows, C, max_len, P.shape[0],
                                  L.dtype_code(dtype), L.stream_ptr()), "cst_dec_embed")
        ref = (3.25 * E[tokens[s & 1, :, s]].float() + P[pad + 1 + s]).to(dtype)
        assert torch.equal(x, ref), s


@pytest.mark.parametrize("M,N,K,act", [(160, 3072, 1024, "none"), (160, 1024, 1024, "none"), (160, 4096, 1024, "relu"),
                                       (37, 1000, 512, "gelu"), (160, 10000, 1024, "none")])
def test_dec_ln_linear_matches_layernorm_then_linear(KL, M, N, K, act):
    """cst_dec_ln_linear (LayerNorm folded into the projection: raw rows x gamma-scaled weights, row statistics gathered in the
    kernel) against F.layer_norm -> F.linear -> activation evaluated in fp32 on the same bf16 parameters.  Rows get a large common
    offset (mean 3 sigma) so that the mean-correction term  - mean * sum_k Wg[n,k]  is exercised for real.  Tolerance: the bf16
    rounding of the folded weights and of the result (2 x 2^-8 of the operand scale)."""
    Kk, L = KL
    g = torch.Generator().manual_seed(N + K)
    dt = torch.bfloat16
    x = (torch.randn(M, K, generator=g) + 3.0 * torch.randn(M, 1, generator=g)).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    ln = torch.nn.LayerNorm(K).cuda()
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.3 * torch.randn(K, generator=g))
        ln.bias.copy_(0.2 * torch.randn(K, generator=g))
    ln = ln.to(dt)
    eng = import_module("chimera-st_amd.decode_engine").BeamDecodeEngine
    wg, sg, sb, eps = eng._fold_ln(ln, W, b)
    y = torch.empty(M, N, dtype=dt, device="cuda")
    code = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "gelu": L.ACT_GELU}[act]
    L.check(L.load().cst_dec_ln_linear(L.ptr(x), L.ptr(wg), L.ptr(sg), L.ptr(sb), eps, None, L.ptr(y), M, N, K, K, 0, N, code, None, 0,
                                       L.dtype_code(dt), L.stream_ptr()), "cst_dec_ln_linear")
    h = torch.nn.functional.layer_norm(x.float(), (K,), ln.weight.float(), ln.bias.float(), ln.eps)
    pre = h @ W.float().t() + b.float()
    ref = pre
    if act == "relu":
        ref = torch.relu(pre)
    elif act == "gelu":
        ref = torch.nn.functional.gelu(pre)
    err = (y.float() - ref).abs()
    scale = float(pre.abs().max())  # the roundings act on the pre-activation
    assert float(err.max()) <= 2.0 * 2.0 ** -8 * max(scale, 1.0), (float(err.max()), scale)
    assert float(err.mean()) <= 2.0 ** -8 * max(float(pre.abs().mean()), 0.1)


@pytest.mark.parametrize("M,N,K,act,has_bias,has_resid", [
    (160, 1024, 1024, "none", True, True),    # out_proj / fc2-shaped: one 16-column tile per workgroup
    (160, 3072, 1024, "none", True, False),   # packed q | k | v projection
    (160, 4096, 1024, "relu", True, False),   # fc1 + activation: two column tiles per workgroup
    (160, 1024, 4096, "gelu", True, True),    # long K
    (160, 10000, 1024, "none", False, False),  # vocabulary projection: four column tiles per workgroup
    (37, 1000, 512, "relu", True, True),      # ragged rows and columns (N % 16 != 0)
    (5, 520, 512, "none", False, True),
])
def test_dec_linear_matches_fp32_reference(KL, M, N, K, act, has_bias, has_resid):
    """cst_dec_linear (decode.hip) against act(x W^T + b) + resid evaluated in fp32 on the same bf16 operands: the kernel accumulates
    in fp32 and rounds once, so it must agree to one bf16 rounding of the result (2^-8 relative) plus the summation-order noise of
    fp32 (K <= 4096 terms)."""
    Kk, L = KL
    g = torch.Generator().manual_seed(M * 7 + N)
    dt = torch.bfloat16
    x = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda() if has_bias else None
    r = torch.randn(M, N, generator=g).to(dt).cuda() if has_resid else None
    ldy = N + 8  # a padded output row (the logits buffer of the engine is one)
    y = torch.full((M, ldy), 7.0, dtype=dt, device="cuda")
    code = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "gelu": L.ACT_GELU}[act]
    L.check(L.load().cst_dec_linear(L.ptr(x), L.ptr(W), L.ptr(b) if b is not None else None, L.ptr(r) if r is not None else None,
                                    L.ptr(y), M, N, K, K, N if r is not None else 0, ldy, code, None, 0, L.dtype_code(dt),
                                    L.stream_ptr()), "cst_dec_linear")
    ref = x.float() @ W.float().t()
    if b is not None:
        ref = ref + b.float()
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if r is not None:
        ref = ref + r.float()
    got = y[:, :N].float()
    err = (got - ref).abs()
    assert float((err - (2.0 ** -8) * ref.abs()).max()) <= 2e-3, float(err.max())
    assert float((y[:, N:].float() - 7.0).abs().max()) == 0.0  # nothing written behind the row
    # deterministic: a second launch gives the same bits
    y2 = torch.full((M, ldy), 7.0, dtype=dt, device="cuda")
    L.check(L.load().cst_dec_linear(L.ptr(x), L.ptr(W), L.ptr(b) if b is not None else None, L.ptr(r) if r is not None else None,
                                    L.ptr(y2), M, N, K, K, N if r is not None else 0, ldy, code, None, 0, L.dtype_code(dt),
                                    L.stream_ptr()), "cst_dec_linear")
    assert torch.equal(y, y2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,D,beam,S", [(8, 64, 5, 375), (2, 32, 1, 64), (4, 64, 10, 130), (16, 64, 3, 33), (4, 64, 32, 97), (2, 64, 1, 1),
                                        (16, 64, 5, 750), (2, 64, 33, 70)])
def test_dec_cross_attn(KL, dtype, H, D, beam, S):
    """Shared per-sentence K/V, beam queries each, ragged key padding — against plain fp32 softmax attention.  bf16 / head dim 64 /
    beam <= 32 runs the matrix-core kernel whose four waves split the keys (fewer tiles than waves, a ragged last tile, 32 query rows and
    the bench's own shape are in the list); beam 33, head dim 32 and fp32 run the VALU kernel."""
    k, L = KL
    lib = L.load()
    bsz, C = 3, H * D
    g = torch.Generator().manual_seed(H * 1000 + S)
    q = torch.randn(bsz * beam, C, generator=g).to(dtype).cuda()
    kx = torch.randn(bsz, S, C, generator=g).to(dtype).cuda()
    vx = torch.randn(bsz, S, C, generator=g).to(dtype).cuda()
    lens = torch.tensor([S, max(1, S // 2), max(1, S // 3 + 1)])
    kpm = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
    out = torch.zeros(bsz * beam, C, dtype=dtype, device="cuda")
    scale = D ** -0.5
    for mask in (kpm, None):
        for stepv, expect_write in ((3, True), (99, False)):
            out.zero_()
            step = torch.tensor([stepv], dtype=torch.int32, device="cuda")
            kxh = kx.view(bsz, S, H, D).transpose(1, 2).contiguous()  # head-major [bsz, H, S, D]
            vxh = vx.view(bsz, S, H, D).transpose(1, 2).contiguous()
            L.check(lib.cst_dec_cross_attn(L.ptr(q), L.ptr(kxh), L.ptr(vxh), L.ptr(mask), L.ptr(out), L.ptr(step), 20, bsz, beam, H, D, S,
                                           scale, L.dtype_code(dtype), L.stream_ptr()), "cst_dec_cross_attn")
            if not expect_write:  # past max_len: a no-op
                assert float(out.abs().max()) == 0.0
                continue
            qf = q.float().view(bsz, beam, H, D)
            sc = torch.einsum("bqhd,bjhd->bhqj", qf, kx.float().view(bsz, S, H, D)) * scale
            if mask is not None:
                sc = sc.masked_fill(mask.bool()[:, None, None, :], float("-inf"))
            ref = torch.einsum("bhqj,bjhd->bqhd", torch.softmax(sc, -1), vx.float().view(bsz, S, H, D)).reshape(bsz * beam, C)
            tol = 2e-5 if dtype == torch.float32 else 2e-2
            assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("H,beam,S", [(16, 5, 750), (8, 1, 40), (8, 32, 97), (16, 3, 33)])
def test_dec_ln_q_cross_attn(KL, H, beam, S):
    """LayerNorm + query projection + cross attention of one decode step as ONE launch (cst_dec_ln_q_cross_attn) against plain fp32
    torch — LayerNorm, Linear, softmax attention over the sentence's keys with ragged key padding — and against the two-launch path
    it replaces (cst_dec_ln_linear, then cst_dec_cross_attn), whose stored bf16 q it reproduces up to the summation order over K."""
    k, L = KL
    lib = L.load()
    eng = import_module("chimera-st_amd.decode_engine").BeamDecodeEngine
    dt, D, bsz = torch.bfloat16, 64, 3
    C = H * D
    g = torch.Generator().manual_seed(H * 100 + S)
    x = (torch.randn(bsz * beam, C, generator=g) * 1.5 + 0.3).to(dt).cuda()
    ln = torch.nn.LayerNorm(C).cuda()
    lin = torch.nn.Linear(C, C).cuda()
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(C, generator=g))
        ln.bias.copy_(0.1 * torch.randn(C, generator=g))
        lin.weight.copy_(torch.randn(C, C, generator=g) / C ** 0.5)
        lin.bias.copy_(0.1 * torch.randn(C, generator=g))
    ln, lin = ln.to(dt), lin.to(dt)
    wg, sg, sb, eps = eng._fold_ln(ln, lin.weight, lin.bias)
    wfrag = eng.fragment_major(wg, H)  # the fused launch's weight layout (include/cst.h)
    kx = torch.randn(bsz, S, C, generator=g).to(dt).cuda()
    vx = torch.randn(bsz, S, C, generator=g).to(dt).cuda()
    kxh = kx.view(bsz, S, H, D).transpose(1, 2).contiguous()
    vxh = vx.view(bsz, S, H, D).transpose(1, 2).contiguous()
    lens = torch.tensor([S, max(1, S // 2), max(1, S // 3 + 1)])
    kpm = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8).cuda()
    scale = D ** -0.5
    step = torch.tensor([3], dtype=torch.int32, device="cuda")
    for mask in (kpm, None):
        out = torch.zeros(bsz * beam, C, dtype=dt, device="cuda")
        L.check(lib.cst_dec_ln_q_cross_attn(L.ptr(x), C, L.ptr(wfrag), L.ptr(sg), L.ptr(sb), eps, L.ptr(kxh), L.ptr(vxh), L.ptr(mask), L.ptr(out),
                                            L.ptr(step), 20, bsz, beam, H, D, S, scale, L.dtype_code(dt), L.stream_ptr()), "cst_dec_ln_q_cross_attn")
        # the two-launch path
        q2 = torch.empty(bsz * beam, C, dtype=dt, device="cuda")
        out2 = torch.zeros_like(out)
        L.check(lib.cst_dec_ln_linear(L.ptr(x), L.ptr(wg), L.ptr(sg), L.ptr(sb), eps, None, L.ptr(q2), bsz * beam, C, C, C, 0, C, L.ACT_NONE, None, 0,
                                      L.dtype_code(dt), L.stream_ptr()), "cst_dec_ln_linear")
        L.check(lib.cst_dec_cross_attn(L.ptr(q2), L.ptr(kxh), L.ptr(vxh), L.ptr(mask), L.ptr(out2), L.ptr(step), 20, bsz, beam, H, D, S, scale,
                                       L.dtype_code(dt), L.stream_ptr()), "cst_dec_cross_attn")
        # fp32 torch on the same bf16 parameters
        qf = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps),
                                        lin.weight.float(), lin.bias.float()).view(bsz, beam, H, D)
        sc = torch.einsum("bqhd,bjhd->bhqj", qf, kx.float().view(bsz, S, H, D)) * scale
        if mask is not None:
            sc = sc.masked_fill(mask.bool()[:, None, None, :], float("-inf"))
        ref = torch.einsum("bhqj,bjhd->bqhd", torch.softmax(sc, -1), vx.float().view(bsz, S, H, D)).reshape(bsz * beam, C)
        top = max(1.0, float(ref.abs().max()))
        assert float((out.float() - ref).abs().max()) < 3e-2 * top
        assert float((out2.float() - ref).abs().max()) < 3e-2 * top
        assert float((out.float() - out2.float()).abs().max()) < 2e-2 * top   # same roundings of q; only the K summation order differs
    # past max_len: a no-op
    out.zero_()
    late = torch.tensor([99], dtype=torch.int32, device="cuda")
    L.check(lib.cst_dec_ln_q_cross_attn(L.ptr(x), C, L.ptr(wfrag), L.ptr(sg), L.ptr(sb), eps, L.ptr(kxh), L.ptr(vxh), None, L.ptr(out), L.ptr(late), 20,
                                        bsz, beam, H, D, S, scale, L.dtype_code(dt), L.stream_ptr()), "cst_dec_ln_q_cross_attn")
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("beam", [1, 5])
@pytest.mark.parametrize("use_graph", [False, True])
def test_engine_matches_reference_generator(beam, use_graph):
    """decode_tiny.npz = hypotheses of the REAL reference SequenceGenerator (tools/ref_harness/make_goldens.py)."""
    g = load_golden("decode_tiny.npz")
    model, task, args = build_from_golden(g, "chimera", torch.float32)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    eng_mod = import_module("chimera-st_amd.decode_engine")
    assert eng_mod.BeamDecodeEngine.supported(model.decoder)
    gen = SG([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=12, min_len=1, use_graph=use_graph)
    sample = to_cuda(golden_sample(g))
    for rep in range(2):  # the second call replays the cached graph on re-initialised state
        hyps = gen.generate([model], sample)
        assert gen._engine is not None and gen._engine.use_graph == use_graph
        for b in range(len(hyps)):
            for r in range(min(beam, 3)):
                key = "gen/beam%d/b%d/r%d/" % (beam, b, r)
                assert hyps[b][r]["tokens"].tolist() == g[key + "tokens"].tolist(), key  # bit-exact token ids
                assert abs(float(hyps[b][r]["score"]) - float(g[key + "score"])) < 1e-3
                assert_close(hyps[b][r]["positional_scores"], g[key + "pos_scores"], 1e-3, key + "pos_scores")


def _build_s2t(dtype, d=256, heads=4, layers=2, V=500, seed=3, tied=True):
    load_pkg()
    s2t = import_module("chimera-st_amd.s2t_transformer")
    tasks = import_module("chimera-st_amd.tasks")
    torch.manual_seed(seed)
    task = tasks.SpeechToTextTask(Namespace(data=None, synthetic_vocab_size=V))
    args = Namespace(encoder_embed_dim=d, encoder_ffn_embed_dim=4 * d, encoder_attention_heads=heads, decoder_attention_heads=heads,
                     encoder_layers=layers, decoder_layers=layers, dropout=0.0, conv_channels=2 * d, share_decoder_input_output_embed=tied)
    model = s2t.S2TTransformerModel.build_model(args, task)
    with torch.no_grad():  # sharpen the output distribution (a tied random-init model repeats one token; an untied one wanders)
        model.decoder.output_projection.weight.mul_(4.0)
    return model.to("cuda", dtype).eval(), task


@pytest.mark.parametrize("cross_kernel", ["flash", "flash_hm", "shared"])
@pytest.mark.parametrize("beam", [1, 4])
def test_engine_equals_mirror_loop_ragged_batch(beam, cross_kernel):
    """fp32 s2t_transformer (fbank input, ragged lengths -> encoder_padding_mask): the captured-graph device loop and the
    module-by-module host loop (fused=False: index_select reorder, torch.topk) give identical hypotheses."""
    model, task = _build_s2t(torch.float32, tied=False)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    g = torch.Generator().manual_seed(11)
    B, T = 5, 97
    src = torch.randn(B, T, 80, generator=g).cuda()
    lens = torch.tensor([97, 80, 64, 33, 20]).cuda()
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens}}
    fused = SG([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=24, cross_kernel=cross_kernel)
    mirror = SG([model], task.target_dictionary, beam_size=beam, max_len_a=0, max_len_b=24, fused=False)
    h1, h2 = fused.generate([model], sample), mirror.generate([model], sample)
    assert fused._engine is not None and mirror._engine is None
    toks_seen = set()
    for b in range(B):
        assert len(h1[b]) == len(h2[b]) == beam
        for r in range(beam):
            assert h1[b][r]["tokens"].tolist() == h2[b][r]["tokens"].tolist(), (b, r)
            assert abs(float(h1[b][r]["score"]) - float(h2[b][r]["score"])) < 1e-4
            assert_close(h1[b][r]["positional_scores"], h2[b][r]["positional_scores"].cpu().numpy(), 1e-3, "pos")
            toks_seen.update(h1[b][r]["tokens"].tolist())
    assert len(toks_seen) > 8, "degenerate test: the hypotheses repeat a handful of tokens"


def test_engine_bf16_large_dims_runs_and_agrees_on_first_tokens():
    """s2t_transformer_l decoder dims (d 1024, 16 heads, ffn 4096) in bf16, batch 8 x beam 5: the engine terminates, scores are
    finite and ordered; the first token of the best hypothesis agrees with the mirror loop (bf16 near-ties may diverge later)."""
    model, task = _build_s2t(torch.bfloat16, d=1024, heads=16, layers=2, V=10000)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    g = torch.Generator().manual_seed(5)
    src = torch.randn(8, 120, 80, generator=g).cuda().to(torch.bfloat16)
    lens = torch.tensor([120, 120, 100, 90, 77, 60, 41, 30]).cuda()
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens}}
    h1 = SG([model], task.target_dictionary, beam_size=5, max_len_a=0, max_len_b=20).generate([model], sample)
    h2 = SG([model], task.target_dictionary, beam_size=5, max_len_a=0, max_len_b=20, fused=False).generate([model], sample)
    agree = 0
    for b in range(8):
        sc = [float(h["score"]) for h in h1[b]]
        assert len(sc) == 5 and all(math.isfinite(s) for s in sc) and sc == sorted(sc, reverse=True)
        agree += int(h1[b][0]["tokens"][0]) == int(h2[b][0]["tokens"][0])
    assert agree >= 6


def test_engine_repacks_after_a_fused_optimizer_step():
    """The fused Adam kernel writes parameters through raw pointers (no autograd version bump, same data_ptr).  An engine kept across
    training updates (validation BLEU between epochs) must not decode with the folded / packed weight copies of the previous update:
    its hypotheses equal those of an engine built after the update."""
    g = load_golden("chimera_tiny.npz")
    model, task, args = build_from_golden(g, "chimera", torch.float32)
    crit = import_module("chimera-st_amd.criterions").TripletSTMTContrastiveCriterion(task, False, 0.1, [1.0, 1.0, 1.0], 0.1)
    Trainer = import_module("chimera-st_amd.trainer").Trainer
    targs = Namespace(bf16=True, lr=[5e-2], adam_betas="(0.9, 0.98)", adam_eps=1e-8, weight_decay=0.0, clip_norm=0.0,
                      warmup_updates=1, warmup_init_lr=5e-2, seed=1)
    tr = Trainer(targs, task, model, crit, device="cuda")
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    sample = golden_sample(g)
    dec_sample = to_cuda(golden_sample(g))
    kept = SG([tr.model], task.target_dictionary, beam_size=5, max_len_a=0, max_len_b=12, min_len=1)
    before = kept.generate([tr.model], dec_sample)
    assert kept._engine is not None
    packed_before = kept._engine._packed[1]["layers"][0]["wqkv"].clone()
    for _ in range(3):
        tr.train_step([sample])
    tr.model.eval()
    after_kept = kept.generate([tr.model], dec_sample)
    fresh = SG([tr.model], task.target_dictionary, beam_size=5, max_len_a=0, max_len_b=12, min_len=1)
    after_fresh = fresh.generate([tr.model], dec_sample)
    assert not torch.equal(packed_before, kept._engine._packed[1]["layers"][0]["wqkv"]), "lr 5e-2 x 3 updates must move the weights"
    for b in range(len(after_fresh)):
        for r in range(3):
            assert after_kept[b][r]["tokens"].tolist() == after_fresh[b][r]["tokens"].tolist()
            assert float(after_kept[b][r]["score"]) == float(after_fresh[b][r]["score"])
    del before


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_engine_lanes_give_the_hypotheses_of_the_undivided_batch(monkeypatch, dtype):
    """The engine cuts the batch into groups of sentences that decode side by side on their own streams (CST_DEC_LANES).  Sentences
    never interact in beam search and every kernel's per-row arithmetic is independent of the number of rows, so 1, 2 and 3 lanes
    return the same token ids and bit-identical scores — also on the second call (graph replay on re-initialised state) and
    when the groups finish at different steps (ragged source lengths)."""
    model, task = _build_s2t(dtype, d=512, heads=8, layers=2, V=2000, tied=False)
    SG = import_module("chimera-st_amd.sequence_generator").SequenceGenerator
    g = torch.Generator().manual_seed(7)
    B = 7
    src = torch.randn(B, 150, 80, generator=g).cuda().to(dtype)
    lens = torch.tensor([150, 150, 131, 90, 77, 41, 30]).cuda()
    sample = {"net_input": {"src_tokens": src, "src_lengths": lens}}
    out = {}
    for lanes in (1, 2, 3):
        monkeypatch.setenv("CST_DEC_LANES", str(lanes))
        gen = SG([model], task.target_dictionary, beam_size=4, max_len_a=0, max_len_b=30)
        for rep in range(2):
            h = gen.generate([model], sample)
            assert gen._engine is not None and gen._engine.lanes == lanes and len(gen._engine._state) == lanes
            out[(lanes, rep)] = [[(x["tokens"].tolist(), float(x["score"]), x["positional_scores"].tolist()) for x in hb] for hb in h]
    ref = out[(1, 0)]
    assert len(ref) == B and all(len(hb) == 4 for hb in ref)
    assert len({tuple(hb[0][0]) for hb in ref}) > 1, "degenerate test: every sentence decodes to the same tokens"
    for k, v in out.items():
        assert v == ref, k
