"""Kernel-level view of the hooked covariance forward (R-50-FPN / R-101-FPN, 800x1344): run under rocprofv3 --kernel-trace --stats.
The grouped pass is run 4 x (1 warm-up + 3 timed) by bench._covariance_forward_ms, then round 2's hook-time launches (4 streams, 1
stream) the same number of times: divide the per-kernel call counts accordingly.
Usage (GPU box): rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o trace -- python3 tools/cov_forward_trace.py [50|101]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ms, ms4, ms1, flops, n, st = bench._covariance_forward_ms(dev, depth, only_grouped=os.environ.get("COV_ONLY_GROUPED") == "1")
print(f"R-{depth}: grouped {ms:.3f} ms, hook-time 4 streams {ms4:.3f} ms, 1 stream {ms1:.3f} ms per hooked forward, {n} convs, {st}")
