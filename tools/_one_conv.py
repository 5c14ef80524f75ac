import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops
B, T = 64, 401
ch = (1, 8, 16, 32, 64); Fk = [160, 80, 40, 20, 10]
for k in (4, 2):
    x = torch.randn(B, T, ch[k - 1], Fk[k - 1], device="cuda"); dy = torch.randn(B, T, ch[k], Fk[k], device="cuda")
    dw = torch.zeros(ch[k], ch[k - 1], 2, 3, device="cuda")
    for _ in range(4):
        ops.conv_wgrad(dy, x, dw, B, T, ch[k], Fk[k], ch[k - 1], Fk[k - 1], KT=2, S=2, pad=1, prec="bf16")
torch.cuda.synchronize()
