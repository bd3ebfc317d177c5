"""Where does one end-to-end training step go?  Runs the task-2 step of bench.end_to_end_training piece by piece with
a device synchronisation after each piece (so pieces do not overlap; the sum exceeds the pipelined step) and prints
wall ms per piece.  python3 tools/e2e_sections.py [--f32]"""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f32", action="store_true")
    args = ap.parse_args()
    amp = not args.f32
    import nsgp_repre_amd as N
    from nsgp_repre_amd.detection import StandaloneRoIHead, build_faster_rcnn, synthetic_batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = build_faster_rcnn(task_id=2, task_split=[0, 15, 20]).to(dev)
    head = model.roi_head
    head.replay = True
    head.bbox_featss = torch.relu(torch.randn(150, 12544, device=dev))
    head.tmp_label = torch.randint(0, 15, (150,), device=dev)
    mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
    mix.task_id = 2
    mix.attach_teacher(model)
    model.train()
    x, samples = synthetic_batch(1, (15, 20), dev, seed=0)
    T = {}

    def tick(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return out

    for it in range(6):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            t = model.teacher_model
            tx = tick("teacher backbone+fpn", lambda: t.extract_feat(x))
            with torch.no_grad():
                tp = tick("teacher rpn predict (convs, top-k, decode, NMS)", lambda: t.rpn_head.predict(tx, samples))
                tr = tick("teacher roi predict (RoIAlign, head, per-class NMS)", lambda: t.roi_head.predict(tx, tp, samples))
            tick("pseudo-label filter", lambda: N.detectors.filter_pseudo_labels(tr[0].bboxes, tr[0].scores, samples[0].gt_instances.bboxes, 0.5, 0.7))
            sx = tick("student backbone+fpn fwd", lambda: model.extract_feat(x))
            rl, props = tick("student rpn loss + proposals", lambda: model.rpn_head.loss_and_predict(sx, copy.deepcopy(samples)))
            roi = tick("student roi loss (sample, RoIAlign, head, CE/L1)", lambda: StandaloneRoIHead.loss(head, sx, props, samples))
            rep = tick("replay loss (bank through the head, fused CE)", lambda: head.add_replay_loss({}))
        loss = sum(v for k, v in {**rl, **roi, **rep}.items() if "loss" in k)
        tick("backward", lambda: loss.backward())
        model.zero_grad()
    for k, v in T.items():
        print(f"{sum(v[2:]) / len(v[2:]):8.2f} ms  {k}")
    print(f"{sum(sum(v[2:]) / len(v[2:]) for v in T.values()):8.2f} ms  sum (serialised)")


if __name__ == "__main__":
    main()
