"""Host time of optimizer.step() over the 162-tensor R-50-FPN table when every gradient is a NEW tensor each step (what backward() after
zero_grad(set_to_none=True) produces), with and without projectors; cProfile of the slower case."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import nsgp_oracle as O  # noqa: E402
import nsgp_repre_amd as N  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
table = O.resnet_fpn_step_table(50) if hasattr(O, "resnet_fpn_step_table") else None
layers = O.resnet_fpn_projected_layers(50)
params, names = [], []
for n, cout, D in layers:
    params.append(torch.nn.Parameter(torch.randn(cout, D, device=dev) * 0.01))
    names.append(n)
for i in range(112):      # the un-projected tensors (norms, biases, heads): small
    params.append(torch.nn.Parameter(torch.randn(256, device=dev)))
    names.append(f"other.{i}.bias")
for with_proj in (False, True):
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    opt.param_groups[0]["names"] = names
    if with_proj:
        cache = {}
        for (n, cout, D), p in zip(layers, params):
            if D not in cache:
                cache[D] = bench.make_basis(D, dev, 1000 + D)
            opt.set_basis(n, cache[D][0], cache[D][1])

    def fresh():
        for p in params:
            p.grad = torch.empty_like(p)
    for _ in range(5):
        fresh(); opt.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        fresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); opt.step(); ts.append(time.perf_counter() - t0)
    print(f"projectors {with_proj}: host time of step() median {1e3 * sorted(ts)[15]:.3f} ms  min {1e3 * min(ts):.3f}", flush=True)
    if with_proj:
        pr = cProfile.Profile()
        for _ in range(50):
            fresh()
            pr.enable(); opt.step(); pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
