import sys, time, torch
sys.path.insert(0, '.')
from cruse_amd.data import synth_batch
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet_2
torch.manual_seed(0)
m = unet_2(rnn_groups=int(sys.argv[1]) if len(sys.argv) > 1 else 1, precision="bf16").cuda()
eng = TrainEngine(m, use_graph=False)
pool = [synth_batch(64, 64000, "cuda", s) for s in range(2)]
for i in range(5): eng.step(*pool[i % 2])
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(30):
    a = time.perf_counter(); eng.step(*pool[i % 2]); ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ts.sort()
print(f"host enqueue per step: median {ts[15]*1e3:.2f} ms, min {ts[0]*1e3:.2f}, max {ts[-1]*1e3:.2f}; loop {1e3*(t1-t0)/30:.2f} ms/step; drain after loop {1e3*(t2-t1):.2f} ms; total {1e3*(t2-t0)/30:.2f} ms/step")
