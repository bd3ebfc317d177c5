"""RoI-side stock parts: MaxIoU assignment, random sampling, multi-level RoIAlign, and the
``StandardRoIHead`` recipe (loss / predict) that the fork's replay heads extend
(standard_roi_replay_head.py: ``super().loss`` + ``replay_loss``; ``get_bbox_stuff`` :106-202)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .boxes import bbox2delta, box_iou, delta2bbox
from .structures import Instances


def assign_max_iou(priors: torch.Tensor, gt_bboxes: torch.Tensor, pos_iou_thr: float, neg_iou_thr: float, min_pos_iou: float,
                   match_low_quality: bool) -> torch.Tensor:
    """MaxIoUAssigner -> assigned gt index + 1 per prior (0 = negative, -1 = ignore)."""
    n = priors.shape[0]
    assigned = priors.new_full((n,), -1, dtype=torch.int64)
    if gt_bboxes.shape[0] == 0:
        assigned[:] = 0
        return assigned
    overlaps = box_iou(gt_bboxes, priors)                                   # [G x N]
    max_ov, argmax = overlaps.max(dim=0)
    assigned[max_ov < neg_iou_thr] = 0
    pos = max_ov >= pos_iou_thr
    assigned[pos] = argmax[pos] + 1
    if match_low_quality:
        gt_max = overlaps.max(dim=1).values
        low = (overlaps == gt_max[:, None]) & (gt_max[:, None] >= min_pos_iou)
        ids = torch.arange(1, gt_bboxes.shape[0] + 1, device=priors.device)[:, None]
        best = (low * ids).max(dim=0).values                                # later gts override earlier ones
        assigned = torch.where(best > 0, best, assigned)
    return assigned


def random_sample(assigned: torch.Tensor, num: int, pos_fraction: float):
    """RandomSampler (neg_pos_ub = -1) -> (pos_inds, neg_inds), both ascending.  Same draws as mmdet's
    (task_modules/samplers/random_sampler.py:66-68, base_sampler.py:112-127): the permutation comes from the CPU default generator
    whatever device the boxes live on, and the chosen indices are sorted afterwards (``.unique()``), so a seeded run picks the same
    RoIs -- and hands ``select_five_rois`` the same row order -- as the reference."""
    pos = torch.nonzero(assigned > 0).flatten()
    neg = torch.nonzero(assigned == 0).flatten()
    n_pos = int(num * pos_fraction)
    if pos.numel() > n_pos:
        pos = pos[torch.randperm(pos.numel())[:n_pos].to(pos.device)].sort().values
    n_neg = num - pos.numel()
    if neg.numel() > n_neg:
        neg = neg[torch.randperm(neg.numel())[:n_neg].to(neg.device)].sort().values
    return pos, neg


class RoIAlignExtractor(nn.Module):
    """SingleRoIExtractor(RoIAlign 7x7, featmap_strides [4,8,16,32], finest_scale 56).  Bilinear sampling
    through ``grid_sample`` with a fixed 2x2 samples per bin (mmcv's ``sampling_ratio=0`` adapts the count to
    the RoI size -- a stock-op detail outside the path), ``aligned=True`` pixel model."""

    def __init__(self, output_size=7, featmap_strides=(4, 8, 16, 32), finest_scale=56, sampling=2, out_channels=256):
        super().__init__()
        self.out, self.strides, self.finest, self.s = output_size, list(featmap_strides), finest_scale, sampling
        self.num_inputs, self.out_channels = len(self.strides), out_channels

    def map_levels(self, rois):
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        return torch.floor(torch.log2(scale / self.finest + 1e-6)).clamp(0, len(self.strides) - 1).long()

    def forward(self, feats, rois):
        k, n = rois.shape[0], self.out * self.s
        out = feats[0].new_zeros(k, feats[0].shape[1], self.out, self.out, dtype=torch.float32)
        if k == 0:
            return out
        lvls = self.map_levels(rois)
        frac = (torch.arange(n, device=rois.device, dtype=torch.float32) + 0.5) / n          # sample positions in [0,1]
        for l, stride in enumerate(self.strides):
            idx = torch.nonzero(lvls == l).flatten()
            if idx.numel() == 0:
                continue
            r = rois[idx]
            f = feats[l].float()        # fp32 sampling under autocast too: the bf16 grid_sample backward (atomics) is 4x slower
            H, W = f.shape[2:]
            x1, y1 = r[:, 1] / stride - 0.5, r[:, 2] / stride - 0.5
            bw, bh = (r[:, 3] - r[:, 1]) / stride, (r[:, 4] - r[:, 2]) / stride
            xs = x1[:, None] + bw[:, None] * frac[None]                                        # [k_l x n] pixel-index coords
            ys = y1[:, None] + bh[:, None] * frac[None]
            gx = (2 * (xs + 0.5) / W - 1)[:, None, :].expand(-1, n, -1)
            gy = (2 * (ys + 0.5) / H - 1)[:, :, None].expand(-1, -1, n)
            grid = torch.stack([gx, gy], dim=-1)                                               # [k_l x n x n x 2]
            bidx = r[:, 0].long()
            res = []
            for b in torch.unique(bidx).tolist():
                sel = torch.nonzero(bidx == b).flatten()
                g = grid[sel].reshape(1, -1, n, 2).to(f.dtype)
                sm = F.grid_sample(f[b:b + 1], g, mode="bilinear", padding_mode="zeros", align_corners=False)
                sm = sm.reshape(f.shape[1], sel.numel(), n, n).permute(1, 0, 2, 3)
                res.append((sel, F.avg_pool2d(sm, self.s)))
            for sel, v in res:
                out[idx[sel]] = v.to(out.dtype)
        return out


def bbox2roi(bbox_list):
    return torch.cat([torch.cat([b.new_full((b.shape[0], 1), i), b[:, :4]], dim=-1) for i, b in enumerate(bbox_list)], dim=0)


class StandaloneRoIHead(nn.Module):
    """The StandardRoIHead recipe over a task bbox head (rcnn train_cfg: MaxIoU 0.5/0.5/0.5, RandomSampler 512 /
    0.25 / add_gt_as_proposals; CE + L1 with target stds [0.1,0.1,0.2,0.2]; test_cfg: score_thr 0.05, NMS 0.5,
    100 per image)."""

    TRAIN = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, num=512, pos_fraction=0.25)
    TEST = dict(score_thr=0.05, iou_threshold=0.5, max_per_img=100)
    STDS = (0.1, 0.1, 0.2, 0.2)

    def init_standalone(self, bbox_roi_extractor, bbox_head, train_cfg=None, test_cfg=None):
        from ..registry import MODELS
        if isinstance(bbox_head, dict):
            bbox_head = MODELS.build(bbox_head)
        if bbox_roi_extractor is None or isinstance(bbox_roi_extractor, dict):
            bbox_roi_extractor = RoIAlignExtractor()
        self.bbox_head, self.bbox_roi_extractor = bbox_head, bbox_roi_extractor
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.with_shared_head = False

    # -- shared pieces ------------------------------------------------------------------------
    def _sample(self, rpn_results_list, batch_data_samples):
        out = []
        c = self.TRAIN
        for props, sample in zip(rpn_results_list, batch_data_samples):
            gt = sample.gt_instances
            priors = torch.cat([gt.bboxes.to(props.bboxes), props.bboxes[:, :4]], dim=0)    # add_gt_as_proposals
            assigned = assign_max_iou(priors, gt.bboxes.to(priors), c["pos_iou_thr"], c["neg_iou_thr"], c["min_pos_iou"],
                                      c["match_low_quality"])
            pos, neg = random_sample(assigned, c["num"], c["pos_fraction"])
            out.append(dict(pos_priors=priors[pos], neg_priors=priors[neg], pos_gt_bboxes=gt.bboxes.to(priors)[assigned[pos] - 1],
                            pos_gt_labels=gt.labels.to(priors.device)[assigned[pos] - 1]))
        return out

    def get_roi_targets(self, sampling_results):
        """BBoxHead.get_targets: labels (bg = num_classes), label weights 1, encoded boxes for the positives."""
        nc = self.bbox_head.num_classes
        labels, lw, bt, bw = [], [], [], []
        for s in sampling_results:
            np_, nn_ = s["pos_priors"].shape[0], s["neg_priors"].shape[0]
            lab = s["pos_priors"].new_full((np_ + nn_,), nc, dtype=torch.int64)
            lab[:np_] = s["pos_gt_labels"]
            t = s["pos_priors"].new_zeros(np_ + nn_, 4)
            w = s["pos_priors"].new_zeros(np_ + nn_, 4)
            if np_ > 0:
                t[:np_] = bbox2delta(s["pos_priors"], s["pos_gt_bboxes"], stds=self.STDS)
                w[:np_] = 1.0
            labels.append(lab)
            lw.append(torch.ones_like(lab, dtype=torch.float32))
            bt.append(t)
            bw.append(w)
        return torch.cat(labels), torch.cat(lw), torch.cat(bt), torch.cat(bw)

    def _bbox_forward(self, x, rois):
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=feats)

    # -- training ------------------------------------------------------------------------------
    def loss(self, x, rpn_results_list, batch_data_samples) -> dict:
        sampling = self._sample(rpn_results_list, batch_data_samples)
        rois = bbox2roi([torch.cat([s["pos_priors"], s["neg_priors"]]) for s in sampling])
        res = self._bbox_forward(x, rois)
        labels, label_w, bbox_t, bbox_w = self.get_roi_targets(sampling)
        cls_score, bbox_pred = res["cls_score"].float(), res["bbox_pred"].float()
        n = max(labels.numel(), 1)
        losses = dict(loss_cls=F.cross_entropy(cls_score, labels, reduction="sum") / n)
        with torch.no_grad():
            losses["acc"] = (cls_score.argmax(-1) == labels).float().mean() * 100
        pos = torch.nonzero(labels < self.bbox_head.num_classes).flatten()
        if pos.numel() > 0:
            pred = bbox_pred.reshape(bbox_pred.shape[0], -1, 4)[pos, labels[pos]]
            losses["loss_bbox"] = (pred - bbox_t[pos]).abs().sum() / n
        else:
            losses["loss_bbox"] = bbox_pred.sum() * 0
        return losses

    # -- inference (the teacher runs this every training step, det:72-74) --------------------------
    @torch.no_grad()
    def predict(self, x, rpn_results_list, batch_data_samples, rescale=False):
        c = self.TEST
        rois = bbox2roi([p.bboxes for p in rpn_results_list])
        res = self._bbox_forward(x, rois)
        scores_all = res["cls_score"].float().softmax(-1)
        nc = self.bbox_head.num_classes
        out, start = [], 0
        for props, sample in zip(rpn_results_list, batch_data_samples):
            k = props.bboxes.shape[0]
            scores, deltas = scores_all[start:start + k, :nc], res["bbox_pred"][start:start + k].float()
            boxes = delta2bbox(props.bboxes[:, :4], deltas, stds=self.STDS, max_shape=sample.img_shape).reshape(k, nc, 4)
            start += k
            valid = scores > c["score_thr"]
            idx = torch.nonzero(valid)
            b, s, lab = boxes[idx[:, 0], idx[:, 1]], scores[idx[:, 0], idx[:, 1]], idx[:, 1]
            keep = ops.nms(b, s, c["iou_threshold"], idxs=lab, max_keep=c["max_per_img"]) if b.shape[0] else idx[:0, 0]
            out.append(Instances(bboxes=b[keep], scores=s[keep], labels=lab[keep]))
        return out

    # -- RePRE dump ---------------------------------------------------------------------------
    @torch.no_grad()
    def sampled_roi_stuff(self, x, rpn_results_list, batch_data_samples):
        """The stock half of ``get_bbox_stuff`` (head:133-161): sample, RoIAlign, mid features, targets."""
        sampling = self._sample(rpn_results_list, batch_data_samples)
        rois = bbox2roi([torch.cat([s["pos_priors"], s["neg_priors"]]) for s in sampling])
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        return (self.bbox_head.get_mid_features(feats),) + self.get_roi_targets(sampling) + (rois,)
