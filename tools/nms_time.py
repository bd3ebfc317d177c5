import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsgp_repre_amd as N
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
def rpn_like(n=4741, levels=5):
    c = torch.rand(n, 2, generator=g) * torch.tensor([1344., 800.])
    wh = torch.rand(n, 2, generator=g) * 200 + 16
    # clustered: many near-duplicates
    c[: n // 2] = c[n // 2: n // 2 * 2][torch.randint(0, n // 2, (n // 2,), generator=g)] + torch.randn(n // 2, 2, generator=g) * 8
    return torch.cat([c - wh / 2, c + wh / 2], -1).to(dev), torch.rand(n, generator=g).to(dev), torch.randint(0, levels, (n,), generator=g).to(dev)
for name, (n, lv, thr, keep) in dict(rpn=(4741, 5, 0.7, 1000), roi=(3000, 15, 0.5, 100), big=(20000, 5, 0.7, 2000)).items():
    b, s, idx = rpn_like(n, lv)
    for _ in range(3): k = N.ops.nms(b, s, thr, idx, keep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): k = N.ops.nms(b, s, thr, idx, keep)
    torch.cuda.synchronize(); print(name, n, 'kept', k.numel(), 'ms per call (incl. sort + host sync)', (time.perf_counter() - t0) / 20 * 1e3)
