import importlib, os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
CF = importlib.import_module("chimera-st_amd.functional")
dt = torch.bfloat16
B, T, C, G, Kp = 32, 1499, 768, 16, 128
x = torch.randn(B, T, C, device="cuda").to(dt).requires_grad_(True)
w = (torch.randn(C, C // G, Kp, device="cuda") * 0.02).to(dt).requires_grad_(True)
b = torch.zeros(C, device="cuda", dtype=dt, requires_grad=True)
dy = torch.randn(B, T, C, device="cuda").to(dt)
def run():
    y = CF.pos_conv_gelu_residual(x, w, b, G); y.backward(dy); x.grad = None; w.grad = None; b.grad = None
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print("pos-conv fwd+bwd %.3f ms  (narrow: %s)" % (s.elapsed_time(e) / 10, "off" if os.environ.get("CST_GEMM_NO_NARROW") else "on"))
