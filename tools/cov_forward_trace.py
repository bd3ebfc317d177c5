"""Kernel-level view of one hooked covariance forward (R-50-FPN, 800x1344): run under rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ms, flops, n = bench._covariance_forward_ms(dev, depth)
print(f"R-{depth}: {ms:.3f} ms per hooked forward, {n} convs")
