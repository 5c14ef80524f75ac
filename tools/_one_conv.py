import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops
B, T = 64, 401
x = torch.randn(B, T, 32, 20, device="cuda"); w = torch.randn(64, 32, 2, 3, device="cuda") * 0.1; b = torch.zeros(64, device="cuda")
for _ in range(4):
    ops.conv_gather(x, w, b, B, T, 32, 20, 64, 10, KT=2, S=2, pad=1, prec="bf16")
e = torch.randn(B, T, 16, 40, device="cuda"); w3 = torch.randn(32, 16, 2, 3, device="cuda") * 0.1; b3 = torch.zeros(32, device="cuda")
for _ in range(4):
    ops.conv_gather(e, w3, b3, B, T, 16, 40, 32, 20, KT=2, S=2, pad=1, prec="bf16")
torch.cuda.synchronize()
