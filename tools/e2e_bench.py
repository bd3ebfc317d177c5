"""The end-to-end section of bench.py on its own (for profiling): python3 tools/e2e_bench.py [--f32] [--steps N] [--batch B]."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--graphs", action="store_true")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True: MIOpen searches for the fastest solver per convolution")
    args = ap.parse_args()
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    import nsgp_repre_amd as N
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = bench.end_to_end_training(N, dev, 1, 0, {}, args.steps, 3, not args.f32, args.batch, args.channels_last, args.graphs)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
