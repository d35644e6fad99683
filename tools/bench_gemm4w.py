#!/usr/bin/env python3
"""Correctness + timing of the eligible bf16 GEMM launches under the current CST_GEMM4W mode (0 = gemm8p / 16-wave kernels only,
1 = default policy, 2 = gemm4w for every eligible launch).  Run once per mode; prints one line per case."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
rng = importlib.import_module("chimera-st_amd.rng")
dt = torch.bfloat16
mode = os.environ.get("CST_GEMM4W", "1")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def gelu(x):
    return torch.nn.functional.gelu(x)


def dgelu(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5


def case(name, M, N, Kd, bk=1, bias=False, act=None, aux=False, dact=False, resid=False, drop=0.0, check=True, time=True):
    g = torch.Generator(device="cpu").manual_seed(M + N + Kd)
    A = (torch.randn(M, Kd, generator=g) * 0.5).to(dt).cuda()
    B = (torch.randn(N, Kd, generator=g) * 0.05).to(dt).cuda() if bk else (torch.randn(Kd, N, generator=g) * 0.05).to(dt).cuda()
    b = (torch.randn(N, generator=g) * 0.1).to(dt).cuda() if bias else None
    C = torch.empty(M, N, dtype=dt, device="cuda")
    z = torch.empty(M, N, dtype=dt, device="cuda") if aux else None
    zin = (torch.randn(M, N, generator=g)).to(dt).cuda() if dact else None
    r = (torch.randn(M, N, generator=g)).to(dt).cuda() if resid else None
    A_ = L.ACT_GELU if act == "gelu" else (L.ACT_RELU if act == "relu" else L.ACT_NONE)
    key = 1234567

    def run():
        K.gemm(A, B, C, M, N, Kd, a_kmajor=1, b_kmajor=bk, lda=Kd, ldb=Kd if bk else N, ldc=N, bias=b, act=A_, aux_out=z, ld_aux_out=N,
               dact=L.ACT_GELU if dact else L.ACT_NONE, aux_in=zin, ld_aux_in=N, resid=r, ld_resid=N, split_k=1, drop_p=drop, drop_key=key)

    run()
    err = -1.0
    if check:
        ref = A.float() @ (B.float().t() if bk else B.float())
        if bias:
            ref = ref + b.float()
        zref = ref.to(dt).float()
        if act == "gelu":
            ref = gelu(zref)
        elif act == "relu":
            ref = torch.relu(zref)
        else:
            ref = zref
        if drop > 0:
            keep = torch.from_numpy(rng.keep_mask_numpy(key, M * N, drop)).view(M, N).cuda()
            ref = torch.where(keep, ref / (1 - drop), torch.zeros_like(ref))
        if dact:
            ref = ref * dgelu(zin.float())
        if resid:
            ref = ref + r.float()
        err = float((C.float() - ref).abs().max() / ref.abs().max())
        if aux:
            err = max(err, float((z.float() - zref).abs().max() / zref.abs().max()))
    ms = timeit(run) if time else 0.0
    print("mode %s | %-34s M=%6d N=%5d K=%5d | rel err %.2e | %.3f ms %.0f TF/s" % (mode, name, M, N, Kd, err, ms, 2.0 * M * N * Kd / max(ms, 1e-9) / 1e9), flush=True)
    assert err < 3e-2 or not check, name


M = 47968
small = dict(time=False)
case("odd tile edges", 1000, 520, 264, bias=True, act="relu", **small)
case("odd tile edges, B mn-major", 1000, 520, 264, bk=0, resid=True, **small)
case("K not a multiple of 32", 700, 264, 200, bias=True, aux=True, act="gelu", **small)
case("one tile", 256, 128, 64, **small)
case("dropout epilogue", 3608, 3592, 136, bias=True, act="relu", resid=True, drop=0.25, **small)
case("dX*act'(aux_in)+resid, B mn-major", 2000, 776, 1032, bk=0, dact=True, resid=True, **small)
case("fc1 plain", M, 3072, 768)
case("fc1 bias+gelu+aux_out", M, 3072, 768, bias=True, act="gelu", aux=True)
case("fc1 bias+gelu+aux_out+dropout", M, 3072, 768, bias=True, act="gelu", aux=True, drop=0.1, check=False)
case("fc2 bias+resid", M, 768, 3072, bias=True, resid=True)
case("qkv bias", M, 2304, 768, bias=True)
case("out_proj bias+resid", M, 768, 768, bias=True, resid=True)
case("dX fc1 (B mn-major)+resid", M, 768, 3072, bk=0, resid=True)
case("dX fc2 * gelu'(aux_in) (B mn-major)", M, 3072, 768, bk=0, dact=True)
case("dX qkv (B mn-major)", M, 768, 2304, bk=0)
case("enc fc1 d512 relu", 12000, 2048, 512, bias=True, act="relu", aux=True)
case("enc fc2 d512", 12000, 512, 2048, bias=True, resid=True)
case("conv-like K=1536", 32 * 11999, 512, 1536, check=False)
