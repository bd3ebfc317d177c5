"""Timing of the projected step on the R-50-FPN / R-101-FPN tables with projectors from SURVEY 8d's seeded covariances
(eigh -> elbow -> set_basis): the default low-rank form against the dense fp16-split GEMM with the same projectors.
Usage: python tools/lowrank_bench.py [50|101] [steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nsgp_oracle as O  # noqa: E402  (layer table only)
import nsgp_repre_amd as N  # noqa: E402
from nsgp_repre_amd.optim.threshold import elbow_index  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    layers = O.resnet_fpn_projected_layers(depth)
    if os.environ.get("LR_SHAPES"):      # study aid: "rows,D,count[;rows,D,count...]" replaces the table (1x1 convolutions)
        layers = []
        for spec in os.environ["LR_SHAPES"].split(";"):
            r, D, c = (int(x) for x in spec.split(","))
            layers += [(f"neck.lateral_convs.{len(layers) + i}.conv.weight", r, D) for i in range(c)]
    basis = {}
    for n, cout, D in layers:
        if D in basis:
            continue
        gen = torch.Generator(device=dev).manual_seed(2000 + D)
        X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
        lam, Q = torch.linalg.eigh((X.t() @ X).contiguous())
        sv = lam.abs()
        order = torch.argsort(sv, descending=True, stable=True)
        basis[D] = (Q[:, order].contiguous(), int(elbow_index(sv[order].cpu().numpy(), 0.0, "sgd")))
        del X, lam, Q
    print("removed directions per width:", {D: b[1] for D, b in sorted(basis.items())}, flush=True)
    out = {}
    for kind in (("sgd",) if os.environ.get("LR_SHAPES") else ("sgd", "adamw")):
        for low in ((True,) if os.environ.get("LR_SHAPES") else (True, False)):
            gen = torch.Generator(device=dev).manual_seed(7)
            params, names = [], []
            for n, cout, D in layers:
                k = 3 if (("conv2" in n or "fpn_convs" in n) and not os.environ.get("LR_SHAPES")) else 1
                params.append(torch.nn.Parameter(torch.randn(cout, D // (k * k), k, k, device=dev, generator=gen) * 0.02))
                names.append(n)
            for i in range(112):   # the un-projected tensors of the table (BN, biases, heads): ~14.6 M elements
                params.append(torch.nn.Parameter(torch.randn(130000, device=dev, generator=gen) * 0.02))
                names.append(f"plain.{i}")
            opt = (N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True) if kind == "sgd"
                   else N.AdamWNSCL(params, lr=1e-3, weight_decay=0.05, svd=True))
            opt.param_groups[0]["names"] = names
            opt.low_rank = low
            for n, cout, D in layers:
                opt.set_basis(n, *basis[D])
            for p in params:
                p.grad = torch.randn(p.shape, device=dev, generator=gen) * 1e-3
            for _ in range(5):
                opt.step()
            torch.cuda.synchronize()
            opt.profile_begin(steps)
            for _ in range(steps):
                opt.step()
            torch.cuda.synchronize()
            _, u_ms, g_ms = opt.profile_end()
            rec = dict(update_kernel_ms=u_ms, projection_launches_ms=g_ms, nsgp_step_ms=u_ms + g_ms, lowrank=opt.lowrank_stats(),
                       dense_tiles=opt.tile_counts(), algorithmic_dense_flops=opt.plan_stats()[0])
            out[f"{kind}_{'low_rank' if low else 'dense_f16x2'}"] = rec
            print(kind, "low_rank" if low else "dense", json.dumps(rec), flush=True)
            opt.close()
            del opt, params
            torch.cuda.empty_cache()
    rd = os.environ.get("NSGP_REPORT_DIR")
    if rd:
        os.makedirs(rd, exist_ok=True)
        json.dump(out, open(os.path.join(rd, f"lowrank_bench_r{depth}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
