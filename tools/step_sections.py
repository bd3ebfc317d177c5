"""Where one task-2 training step spends its time (measurement aid, not part of the product or the bench).

Runs the stand-alone R-50-FPN detector step of bench.py section by section, twice: with a device synchronisation after every
section (GPU time per section) and with none (host time to ISSUE the section: when the sum of these approaches the step time
the step is launch-bound).  Sections follow faster_rcnn_roi_replay.py:44-142: teacher predict (features, RPN predict, RoI
predict), pseudo-label filter, student features, RPN loss + proposals, RoI loss (+ replay), backward, optimizer step."""
import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsgp_repre_amd as N  # noqa: E402
from nsgp_repre_amd.detection import build_faster_rcnn, relocate_segment_final_weights, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--f32", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(4321)
    model = build_faster_rcnn(depth=50, num_classes=20, task_id=2, task_split=[0, 15, 20]).to(dev)
    relocate_segment_final_weights(model)      # guard against a stock MIOpen over-read (profiles/README.md, incident analysis)
    head = model.roi_head
    head.replay = True
    head.bbox_featss = torch.relu(torch.randn(150, 12544, device=dev))
    head.tmp_label = torch.randint(0, 15, (150,), device=dev)
    mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
    mix.task_id = 2
    mix.attach_teacher(model)
    opt = N.SGDNSCL(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    N.runner.nullspace.wire_param_names(opt, model)
    model.train()
    batches = [synthetic_batch(1, (15, 20), dev, seed=i) for i in range(4)]
    names = ["teacher.features", "teacher.rpn_predict", "teacher.roi_predict", "pseudo_label_filter", "student.features",
             "student.rpn_loss_and_predict", "student.roi_loss", "student.replay_loss", "backward", "optimizer.step", "optimizer.zero_grad"]

    def step(i, sync, acc):
        x, samples = batches[i % 4]
        samples = copy.deepcopy(samples)
        t = [time.perf_counter()]

        def mark():
            if sync:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.f32):
            teacher = model.teacher_model
            with torch.no_grad():
                teacher.eval()
                tx = teacher.extract_feat(x); mark()
                ts = copy.deepcopy(samples)
                props = teacher.rpn_head.predict(tx, ts, rescale=False); mark()
                preds = teacher.roi_head.predict(tx, props, ts, rescale=False); mark()
                rpn_samples = copy.deepcopy(samples)
                from nsgp_repre_amd.detectors.faster_rcnn_roi_replay import filter_pseudo_labels
                for inst, gt_sample, rpn_sample in zip(preds, samples, rpn_samples):
                    if len(inst) == 0:
                        continue
                    to_rpn, to_roi = filter_pseudo_labels(inst.bboxes, inst.scores, gt_sample.gt_instances.bboxes, model.rpn_thresh, model.roi_thresh)
                    pseudo = inst[:]
                    pseudo.__delattr__("scores")
                    rpn_sample.gt_instances = rpn_sample.gt_instances.cat([rpn_sample.gt_instances, pseudo[to_rpn]])
                    gt_sample.gt_instances = gt_sample.gt_instances.cat([gt_sample.gt_instances, pseudo[to_roi]])
                mark()
            sx = model.extract_feat(x); mark()
            for s in rpn_samples:
                s.gt_instances.labels = torch.zeros_like(s.gt_instances.labels)
            rpn_losses, proposals = model.rpn_head.loss_and_predict(sx, rpn_samples, proposal_cfg=None); mark()
            from nsgp_repre_amd.detection.roi_parts import StandaloneRoIHead
            roi_losses = StandaloneRoIHead.loss(head, sx, proposals, samples); mark()
            roi_losses = head.add_replay_loss(roi_losses); mark()
        loss = sum(v for k, v in {**rpn_losses, **roi_losses}.items() if "loss" in k)
        loss.backward(); mark()
        opt.step(); mark()
        opt.zero_grad(); mark()
        if not sync:
            torch.cuda.synchronize()
            t.append(time.perf_counter())
        if acc is not None:
            for k, (a, b) in enumerate(zip(t[:-1], t[1:])):
                acc[k] = acc.get(k, 0.0) + (b - a) * 1e3
    for i in range(4):
        step(i, True, None)
    out = {}
    for sync in (True, False):
        acc = {}
        for i in range(args.steps):
            step(i, sync, acc)
        key = "gpu_ms_per_section (sync after each)" if sync else "host_issue_ms_per_section (no sync; last entry = drain at step end)"
        labels = names + ([] if sync else ["drain"])
        out[key] = {labels[k]: round(v / args.steps, 3) for k, v in sorted(acc.items())}
        out[key]["total"] = round(sum(acc.values()) / args.steps, 3)
    out["detector_dtype"] = "f32" if args.f32 else "bf16 autocast"
    print(json.dumps(out, indent=1))
    opt.close()


if __name__ == "__main__":
    main()
