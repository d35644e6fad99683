import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
B, T, H, D = 32, 1499, 12, 64
dt = torch.bfloat16
q = torch.randn(B, T, H * D, device="cuda").to(dt); k = torch.randn_like(q); v = torch.randn_like(q)
o, lse = K.attn_fwd(q, k, v, H, D, None, False, 0.125)
do = torch.randn_like(o)
for _ in range(5):
    K.attn_fwd(q, k, v, H, D, None, False, 0.125)
    K.attn_bwd(do, q, k, v, o, lse, H, D, None, False, 0.125)
torch.cuda.synchronize()
