import sys, time, torch
sys.path.insert(0, '.')
from cruse_amd import ops
from cruse_amd.data import synth_batch
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet_2
for g, graph in ((1, False), (1, True), (4, True), (4, False)):
    torch.manual_seed(0)
    m = unet_2(rnn_groups=g, precision="bf16").cuda()
    eng = TrainEngine(m, lr=1e-3, use_graph=graph, clip_grad_norm=5.0)
    pool = [synth_batch(64, 64000, "cuda", 100 + s) for s in range(8)]
    losses = []
    t0 = time.perf_counter()
    for i in range(400):
        ls = eng.step(*pool[i % 8])
        if i % 50 == 0 or i == 399:
            losses.append(round(eng.loss_value(ls), 5))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"g={g} graph={graph}: 400 steps {dt/400*1e3:.2f} ms/step, losses {losses}, gru_status {ops.gru_status()}, skipped {eng.skipped_steps()}", flush=True)
    assert ops.gru_status() == 0 and eng.skipped_steps() == 0 and losses[-1] < losses[0]
