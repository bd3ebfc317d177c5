"""One hooked covariance forward of R-50-FPN / R-101-FPN at 800x1344 on 1 .. 8 side HIP streams (CovarianceStreams):
wall time per forward and a bitwise comparison of the covariances against the single-stream result.
Usage: python tools/cov_streams.py [50|101]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nsgp_repre_amd import ops  # noqa: E402
from nsgp_repre_amd.runner.nullspace import CovarianceStreams  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    layers = bench.r50_fpn_hooked_convs(depth=depth)
    acts = {}
    for n, cin, k, s, p, h, w in layers:
        if (cin, h, w) not in acts:
            acts[(cin, h, w)] = torch.randn(1, cin, h, w, device=dev, generator=g).abs()
    ref = None
    for ns in (1, 2, 4, 8):
        side = CovarianceStreams(ns)
        covs = {}

        def forward():
            for slot, (n, cin, k, s, p, h, w) in enumerate(layers):
                x = acts[(cin, h, w)]
                nb = ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p))
                covs[n] = side.run(slot, x, lambda wsf, x=x, k=k, s=s, p=p, n=n, nb=nb: ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), covs.get(n), wsf(nb)))
            side.join()
        forward()
        torch.cuda.synchronize()
        first = {n: c.clone() for n, c in covs.items()}
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); forward(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = first
        same = all(torch.equal(first[n], ref[n]) for n in ref)
        print(f"R-{depth}: {ns} stream(s): {sorted(ts)[2]:.3f} ms per forward (min {min(ts):.3f}); first-pass covariances bitwise equal to 1 stream: {same}", flush=True)


if __name__ == "__main__":
    main()
