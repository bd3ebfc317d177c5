"""Soak (GPU box): 120 hooked forwards of a small net with inputs of changing size through the grouped covariance pass (plans per geometry, one shared
workspace: memory flat, covariances finite and bit-symmetric), then 300 optimizer steps with fresh gradients over a common and a wide (r = 180) rank class
(memory flat, parameters finite).  Usage: python tools/soak.py"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import torch.nn as nn
import nsgp_repre_amd as N
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 1), nn.ReLU(), nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(),
                    nn.Conv2d(128, 64, 3, stride=2, padding=1)).to(dev)
col = N.runner.CovarianceCollector(net, [], grouped=True).register()
sizes = [(96, 160), (128, 128), (96, 160), (112, 144), (64, 200), (128, 128)]
mem = []
with torch.no_grad():
    for it in range(120):
        h, w = sizes[it % len(sizes)]
        x = torch.randn(2, 64, h, w, device=dev).abs()
        net(x)
        col.flush()
        if it % 20 == 19:
            torch.cuda.synchronize()
            mem.append((it, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, len(col._plans)))
print("covariance soak (MB allocated, reserved, plans):", mem)
fin = all(torch.isfinite(v).all().item() for v in col.fea_in.values())
sym = all(torch.equal(v, v.t().contiguous()) for v in col.fea_in.values())
print("finite", fin, "bit-symmetric", sym)
col.remove(); col.close()
assert fin and sym and mem[-1][1] <= mem[1][1] + 8, mem
# optimizer soak: 300 steps, fresh grads, memory flat, parameters finite
ps = [nn.Parameter(torch.randn(256, 2304, device=dev) * 0.01), nn.Parameter(torch.randn(512, 4608, device=dev) * 0.01), nn.Parameter(torch.randn(64, device=dev))]
opt = N.SGDNSCL(ps, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
opt.param_groups[0]["names"] = ["backbone.a.weight", "neck.b.weight", "x.bias"]
for p, n, r in ((ps[0], "backbone.a.weight", 40), (ps[1], "neck.b.weight", 180)):
    Q, _ = torch.linalg.qr(torch.randn(p.shape[1], 256, device=dev))
    V = torch.zeros(p.shape[1], p.shape[1], device=dev); V[:, :256] = Q
    opt.set_basis(n, V, r)
m0 = None
for it in range(300):
    for p in ps:
        p.grad = torch.randn_like(p) * 0.1
    opt.step(); opt.zero_grad()
    if it == 20:
        torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
torch.cuda.synchronize()
print("optimizer soak: allocated MB at step 20 / 300:", m0 >> 20, torch.cuda.memory_allocated() >> 20, "lowrank layers", opt.lowrank_stats()[0], "finite", all(torch.isfinite(p).all().item() for p in ps))
assert torch.cuda.memory_allocated() <= m0 + (8 << 20)
print("soak ok")
