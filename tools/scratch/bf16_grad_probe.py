import sys, torch
sys.path.insert(0, '.')
from cruse_amd.model.cruse_net import unet_2
from cruse_amd.model.cruse import CRUSE4MagAddSkipUpsample
from cruse_amd import config


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(cls, **kw):
    torch.manual_seed(5)
    big = cls(precision="bf16", **kw).cuda(); ref = cls(precision="f32", **kw).cuda()
    ref.load_state_dict(big.state_dict())
    xb, wb = torch.rand(8, 1, 401, 160).cuda() + 0.05, torch.randn(8, 1, 401, 160).cuda()
    m, mr = big(xb), ref(xb)
    (m * wb).sum().backward(); (mr * wb).sum().backward()
    pairs = [(n, a.grad, b.grad) for (n, a), (_, b) in zip(big.named_parameters(), ref.named_parameters())
             if a.grad is not None and float(b.grad.norm()) > 1e-3]
    allg = rel(torch.cat([a.flatten() for _, a, _ in pairs]), torch.cat([b.flatten() for _, _, b in pairs]))
    worst = sorted(((rel(a, b), n) for n, a, b in pairs), reverse=True)[:4]
    print(cls.__name__, kw, f"mask {rel(m, mr):.2e} all {allg:.2e}", worst)


run(unet_2, rnn_groups=1)
run(CRUSE4MagAddSkipUpsample, rnn_groups=1)
with config.use(config.get().copy(bf16_dy=False)):
    run(CRUSE4MagAddSkipUpsample, rnn_groups=1)
with config.use(config.get().copy(conv_bwd_x3=True)):
    run(CRUSE4MagAddSkipUpsample, rnn_groups=1)
