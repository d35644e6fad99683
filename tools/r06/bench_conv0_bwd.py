#!/usr/bin/env python3
"""conv0 backward at the bench's frame counts (B = 32, lengths uniform 10-30 s): launch time of cst_conv0_gn_gelu_bwd.
CST_CONV0_NO_MFMA=1 selects the register kernel (A/B in separate processes)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("chimera-st_amd.kernels")
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
B, smax = 32, 480000
audio = sorted([int(torch.randint(160000 // 320, smax // 320 + 1, (1,), generator=g)) * 320 for _ in range(B)], reverse=True)
audio[0] = smax
S, C, kk, st = smax, 512, 10, 5
L = (S - kk) // st + 1
wav = torch.zeros(B, S, device="cuda")
for i, s in enumerate(audio):
    wav[i, :s] = 0.1 * torch.randn(s, generator=g).cuda()
w = (0.5 * torch.randn(C, kk, generator=g)).to(dt).cuda()
ga, be = (1 + 0.1 * torch.randn(C, generator=g)).to(dt).cuda(), (0.1 * torch.randn(C, generator=g)).to(dt).cuda()
y, mean, rstd, gram = K.conv0_fwd(wav, w, ga, be, kk, st)
lim0 = torch.tensor([(s_ - kk) // st + 1 for s_ in audio], dtype=torch.int32, device="cuda")
for _ in range(3):
    K.conv0_fwd(wav, w, ga, be, kk, st, frame_limit=lim0)
torch.cuda.synchronize()
f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
f0.record()
for _ in range(10):
    K.conv0_fwd(wav, w, ga, be, kk, st, frame_limit=lim0)
f1.record()
torch.cuda.synchronize()
print("forward, frame limits: %.3f ms" % (f0.elapsed_time(f1) / 10))
dy = torch.randn(B, L, C, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda").to(dt)
lim = torch.tensor([(s - kk) // st + 1 for s in audio], dtype=torch.int32, device="cuda")
for i in range(B):
    dy[i, int(lim[i]):] = 0
for name, fl in (("all frames", None), ("frame limits", lim)):
    for _ in range(3):
        out = K.conv0_bwd(dy, wav, w, ga, be, mean, rstd, gram, kk, st, frame_limit=fl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = K.conv0_bwd(dy, wav, w, ga, be, mean, rstd, gram, kk, st, frame_limit=fl)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.3f ms  dw[0,:3]=%s" % (name, e0.elapsed_time(e1) / 10, out[0][0, :3].tolist()))
torch.save([o.float().cpu() for o in out], os.environ.get("OUT", "/tmp/conv0_bwd_out.pt"))
