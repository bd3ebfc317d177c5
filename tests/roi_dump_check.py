"""Shared by the CPU and the GPU test of the RoI dump: run the stand-alone head's ``get_bbox_stuff`` on the G8 inputs under the
G8 torch seed and compare everything with what the REFERENCE's ``get_bbox_stuff`` (head:106-202, driving its own
MaxIoUAssigner / RandomSampler / DeltaXYWHBBoxCoder / BBoxHead.get_targets) returned for the same inputs
(tests/golden/make_golden.py ``run_roi_dump``).  mmcv's RoIAlign is absent from the image, so both sides use the closed-form
stand-in extractor ``inputs.g8_extract``: the dumped feature rows then identify the RoIs they came from."""
import numpy as np
import torch
import torch.nn as nn

import inputs as I


class _Extractor(nn.Module):
    num_inputs = 4

    def forward(self, feats, rois):
        return I.g8_extract(rois)


class _Head(nn.Module):
    num_classes = I.G8_NUM_CLASSES

    def get_mid_features(self, f):
        return f.flatten(1)


def check_roi_dump(N, golden_dir, device):
    from nsgp_repre_amd.detection import roi_parts
    from nsgp_repre_amd.detection.structures import DetSample, Instances
    G = np.load(f"{golden_dir}/g8_roi_dump.npz")
    from nsgp_repre_amd.roi_heads.replay_head import StandardRoIReplayHead
    head = StandardRoIReplayHead(bbox_roi_extractor=_Extractor(), bbox_head=_Head())
    c = head.TRAIN
    for ci, (case, seed, _) in enumerate(I.G8_CASES):
        imgs = I.g8_case(ci)
        rpn = [Instances(bboxes=torch.from_numpy(im["proposals"]).to(device), scores=torch.from_numpy(im["scores"]).to(device)) for im in imgs]
        samples = [DetSample(Instances(bboxes=torch.from_numpy(im["gt_bboxes"]).to(device), labels=torch.from_numpy(im["gt_labels"]).to(device)),
                             img_shape=I.G8_CANVAS) for im in imgs]
        # the pieces, against the reference's intermediate results
        for i, im in enumerate(imgs):
            props, gt = rpn[i].bboxes, samples[i].gt_instances.bboxes
            assigned = roi_parts.assign_max_iou(props, gt, c["pos_iou_thr"], c["neg_iou_thr"], c["min_pos_iou"], c["match_low_quality"])
            assert np.array_equal(assigned.cpu().numpy(), G[f"{case}__img{i}__gt_inds"]), (case, i)
            rpn_style = roi_parts.assign_max_iou(props, gt, 0.7, 0.3, 0.3, True)      # the RPN's setting: low-quality matches on
            assert np.array_equal(rpn_style.cpu().numpy(), G[f"{case}__img{i}__rpn_gt_inds"]), (case, i)
        # the whole call, same seed: the sampler's two permutations per image, then the five-row selection's
        x = [torch.zeros(len(imgs), 1, 1, 1, device=device)] * 4
        torch.manual_seed(seed)
        feats, cls_t, cls_w, box_t, box_w, rois = head.get_bbox_stuff(x, rpn, samples)
        assert feats.shape == (5, I.G8_FEAT_C * 49) or case == "few_fg" and feats.shape[0] == 5
        assert np.array_equal(cls_t.cpu().numpy(), G[f"{case}__cls_t"]), case                    # integer labels: bit-exact
        assert np.array_equal(rois.cpu().numpy(), G[f"{case}__rois"]), case                      # copied boxes: bit-exact
        assert np.array_equal(cls_w.cpu().numpy(), G[f"{case}__cls_w"]) and np.array_equal(box_w.cpu().numpy(), G[f"{case}__box_w"]), case
        ref_t = G[f"{case}__box_t"]
        assert np.abs(box_t.cpu().numpy() - ref_t).max() <= 1e-5 * max(np.abs(ref_t).max(), 1.0), case    # fp32 encode (log, divide)
        ref_f = G[f"{case}__feats"]
        assert np.abs(feats.cpu().numpy() - ref_f).max() <= 1e-6 * np.abs(ref_f).max(), case
    # the sampler alone on the reference's assignment (gts in front, as BaseSampler.sample adds them: base_sampler.py:96-110)
    for ci, (case, seed, _) in enumerate(I.G8_CASES):
        torch.manual_seed(seed)
        for i, im in enumerate(I.g8_case(ci)):
            g = im["gt_bboxes"].shape[0]
            assigned = torch.cat([torch.arange(1, g + 1), torch.from_numpy(G[f"{case}__img{i}__gt_inds"])]).to(device)
            pos, neg = roi_parts.random_sample(assigned, c["num"], c["pos_fraction"])
            assert np.array_equal(pos.cpu().numpy(), G[f"{case}__img{i}__pos_inds"]), (case, i)
            assert np.array_equal(neg.cpu().numpy(), G[f"{case}__img{i}__neg_inds"]), (case, i)
