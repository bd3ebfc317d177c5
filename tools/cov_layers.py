"""Per-layer timing of the covariance SYRK paths on the R-50-FPN hooked convs at 800x1344 (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import bench
from nsgp_repre_amd import ops

dev = torch.device("cuda:0")
layers = bench.r50_fpn_hooked_convs()
seen, rows = {}, []
g = torch.Generator(device=dev).manual_seed(1)
for n, cin, k, s, p, h, w in layers:
    key = (cin, k, s, p, h, w)
    seen.setdefault(key, []).append(n)
tot = {0: 0.0, 2: 0.0, 3: 0.0, "best": 0.0}
print(f"{'layer shape':40s} {'count':>5s} {'L':>7s} {'D':>5s} {'GF(ref)':>8s} | {'fp32':>8s} {'f16 gen1':>8s} {'f16 gen2':>8s} ms per call")
for (cin, k, s, p, h, w), names in seen.items():
    x = torch.randn(1, cin, h, w, device=dev, generator=g).abs()
    ws = torch.empty(ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p)), dtype=torch.uint8, device=dev)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    L, D = ho * wo, cin * k * k
    t = {}
    for mode in (0, 3, 2):
        prev = ops.cov_set_split_mfma(mode)
        cov = None
        for _ in range(2):
            cov = ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), cov, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cov = ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), cov, ws)
        e1.record(); torch.cuda.synchronize()
        t[mode] = e0.elapsed_time(e1) / 5
        ops.cov_set_split_mfma(prev)
        tot[mode] += t[mode] * len(names)
    tot["best"] += min(t.values()) * len(names)
    print(f"{names[0][:40]:40s} {len(names):5d} {L:7d} {D:5d} {2.0*L*D*D/1e9:8.1f} | {t[0]:8.3f} {t[3]:8.3f} {t[2]:8.3f}", flush=True)
print(f"sum over the 61 convs: fp32 {tot[0]:.2f} ms, f16 gen-1 everywhere {tot[3]:.2f} ms, f16 forced (gen-2 where D >= 512) {tot[2]:.2f} ms, best per layer {tot['best']:.2f} ms")
