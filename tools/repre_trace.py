"""The fused RePRE replay pass alone, for a rocprofv3 kernel trace: K = 150 (or argv[1]) prototypes, 25 forward + backward passes.
Usage (GPU box): rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_repre -o trace -- python3 tools/repre_trace.py [K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch  # noqa: E402
import nsgp_repre_amd as N  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 150
split = {150: [0, 15, 20], 100: [0, 10, 20], 400: [0, 40, 80]}.get(K, [0, 15, 20])
dev = torch.device("cuda:0")
torch.manual_seed(7)
head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=split[-1], task_split=split, task_id=2).to(dev)


class Replay(N.roi_heads.PrototypeReplay):
    pass


rp = Replay()
rp.bbox_head, rp.task_split, rp.task_id, rp.replay = head, split, 2, True
rp.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev))
rp.tmp_label = torch.randint(0, split[1], (K,), device=dev)
for _ in range(5):
    head.zero_grad(set_to_none=True)
    rp.add_replay_loss({})["replay_loss_cls"].backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    head.zero_grad(set_to_none=True)
    rp.add_replay_loss({})["replay_loss_cls"].backward()
e1.record()
torch.cuda.synchronize()
print(f"K = {K}: {e0.elapsed_time(e1) / 20:.4f} ms per fused pass, back to back (steady state needs more passes: see bench.py)")
