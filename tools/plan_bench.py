"""Time the library's grouped projection GEMM on synthetic layer tables (GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import nsgp_repre_amd as N
import nsgp_oracle as O

dev = torch.device("cuda:0")

def run(layers, label, steps=12):
    params, names, cache = [], [], {}
    for i, (cout, D) in enumerate(layers):
        params.append(torch.nn.Parameter(torch.randn(cout, D, 1, 1, device=dev) * 0.02))
        names.append(f"backbone.l{i}.weight")
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    opt.param_groups[0]["names"] = names
    for n, (cout, D) in zip(names, layers):
        if D not in cache:
            cache[D] = torch.randn(D, D, device=dev) / D ** 0.5
        opt.transforms[n] = cache[D]
    grads = [torch.randn_like(p) for p in params]
    for it in range(3 + steps):
        for p, g in zip(params, grads):
            p.grad = g
        if it == 3:
            opt.profile_begin(steps)
        opt.step()
    n, u, g = opt.profile_end()
    fl, by, tiles, nl = opt.plan_stats()
    print(f"{label:34s} tiles {tiles:5d}  update {u*1e3:7.1f} us  gemm {g*1e3:8.1f} us  {fl/g/1e9:6.1f} TF", flush=True)

run([(4096, 4096)], "1x(4096,4096)")
run([(512, 4608)] * 3, "3x(512,4608)")
run([(256, 2304)] * 10, "10x(256,2304)")
run([(256, 2304)] * 27, "27x(256,2304)")
run([(256, 1024)] * 23, "23x(256,1024)")
run([(1024, 256)] * 23, "23x(1024,256)")
run([(1024, 256)] * 64, "64x(1024,256)")
run([(2048, 1024)] * 8, "8x(2048,1024)")
r50 = [(c, d) for _, c, d in O.resnet_fpn_projected_layers(50)]
run(r50, "R-50-FPN table")
r101 = [(c, d) for _, c, d in O.resnet_fpn_projected_layers(101)]
run(r101, "R-101-FPN table")
run([x for x in r101 if x[1] >= 1024], "R-101 layers with D >= 1024")
run([x for x in r101 if x[1] < 1024], "R-101 layers with D < 1024")
sys.exit(0)

# --- the single-launch API on one big problem (same kernel body, 2-D grid, pointers as kernel args)
from nsgp_repre_amd import ops
for rows, cols in ((4096, 4096), (512, 4096)):
    a = torch.randn(rows, cols, device=dev)
    P = torch.randn(cols, cols, device=dev) / cols ** 0.5
    out = torch.zeros_like(a)
    for _ in range(3):
        ops.project(a, P, scale=-0.02, out=out, accumulate=True)
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.project(a, P, scale=-0.02, out=out, accumulate=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"nsgp_project single launch ({rows},{cols}) tiles {rows//128*cols//128:5d}  {ts[len(ts)//2]*1e3:8.1f} us  {2.0*rows*cols*cols/ts[len(ts)//2]/1e9:6.1f} TF", flush=True)

# --- clock hypothesis: the same single launch, preceded each time by an HBM-bound elementwise kernel
rows, cols = 4096, 4096
a = torch.randn(rows, cols, device=dev)
P = torch.randn(cols, cols, device=dev) / cols ** 0.5
out = torch.zeros_like(a)
big = torch.randn(64 * 1024 * 1024, device=dev)
for label, pre in (("alone", None), ("after 0.5 GB copy", lambda: big.add_(1.0)), ("after idle 2 ms", "sleep")):
    ts = []
    for it in range(10):
        if pre == "sleep":
            torch.cuda.synchronize(); import time; time.sleep(0.002)
        elif pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.project(a, P, scale=-0.02, out=out, accumulate=True); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    print(f"single launch (4096,4096) {label:20s} {ts[len(ts)//2]*1e3:8.1f} us  {2.0*rows*cols*cols/ts[len(ts)//2]/1e9:6.1f} TF", flush=True)
