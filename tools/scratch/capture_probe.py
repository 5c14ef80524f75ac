"""which part of the config-5 step refuses HIP-graph capture?"""
import sys, torch, traceback
sys.path.insert(0, '.')
from model import mtfaa as M
from cruse_amd.nn_generic import to_f16, to_f32
torch.manual_seed(0)
dev = "cuda"
stft = M.STFT(320, 160, 320, "hann")
pe = M.PhaseEncoder(4, 1).to(dev)
tfcm = M.TFCM(24, (3, 3), 6).to(dev)
x = (0.1 * torch.randn(8, 64000)).to(dev)
params = [q for mod in (pe, tfcm) for q in mod.parameters()]


def parts():
    c = stft.transform(x)
    yield "stft", c
    a = pe([c])
    yield "pe", a
    h = torch.cat([a] * 12, dim=1)
    yield "cat", h
    h16 = to_f16(h)
    yield "to_f16", h16
    y = to_f32(tfcm(h16))
    yield "tfcm", y
    for q in params:
        q.grad = None
    (y.square().mean() * 65536.0).backward()
    yield "backward", y


for _ in range(2):
    for _n, _v in parts():
        pass
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    last = "start"
    try:
        g.capture_begin()
        for name, v in parts():
            last = name
            print("captured", name, flush=True)
        g.capture_end()
        print("capture ok")
    except Exception as ex:
        print("FAILED after", last, repr(ex)[:300])
        traceback.print_exc(limit=12)
