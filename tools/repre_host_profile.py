"""Where the HOST time of one fused replay pass goes (cProfile over 200 passes; the GPU work is ~0.2 ms, the pass is host-bound)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import nsgp_repre_amd as N  # noqa: E402

dev = torch.device("cuda:0")
split = [0, 15, 20]
head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=20, task_split=split, task_id=2).to(dev)


class Replay(N.roi_heads.PrototypeReplay):
    pass


rp = Replay()
rp.bbox_head, rp.task_split, rp.task_id, rp.replay = head, split, 2, True
rp.bbox_featss = torch.relu(torch.randn(150, 12544, device=dev))
rp.tmp_label = torch.randint(0, 15, (150,), device=dev)


def one():
    head.zero_grad(set_to_none=True)
    rp.add_replay_loss({})["replay_loss_cls"].backward()


for _ in range(10):
    one()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200):
    one()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / 200:.4f} ms per pass; with the final sync {1e3 * (t2 - t0) / 200:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
