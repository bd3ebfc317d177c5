"""RPNHead of the stock recipe (faster-rcnn_r50_fpn.py): 3x3 conv 256 + ReLU, 3 anchors per location,
sigmoid objectness + L1 on DeltaXYWH targets; train_cfg.rpn: MaxIoU 0.7/0.3/0.3 with low-quality matches,
RandomSampler 256 / 0.5; proposals: top 2000 per level (train) / 1000 (test), level-aware NMS 0.7, 1000 kept."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .boxes import AnchorGenerator, bbox2delta, delta2bbox
from .roi_parts import assign_max_iou, random_sample
from .structures import Instances


class RPNHead(nn.Module):
    TRAIN = dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, num=256, pos_fraction=0.5)
    PROPOSAL_TRAIN = dict(nms_pre=2000, max_per_img=1000, iou_threshold=0.7)
    PROPOSAL_TEST = dict(nms_pre=1000, max_per_img=1000, iou_threshold=0.7)

    def __init__(self, in_channels=256, feat_channels=256, strides=(4, 8, 16, 32, 64)):
        super().__init__()
        self.anchors = AnchorGenerator(strides)
        a = self.anchors.num_base
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, a, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, a * 4, 1)
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)

    def forward(self, feats):
        cls, reg = [], []
        for f in feats:
            h = F.relu(self.rpn_conv(f))
            cls.append(self.rpn_cls(h))
            reg.append(self.rpn_reg(h))
        return cls, reg

    @staticmethod
    def _flatten(cls, reg):
        """per level [B, A, H, W] / [B, 4A, H, W] -> [B, H*W*A] / [B, H*W*A, 4] (location-major, anchor-minor)."""
        b = cls[0].shape[0]
        return ([c.permute(0, 2, 3, 1).reshape(b, -1).float() for c in cls],
                [r.permute(0, 2, 3, 1).reshape(b, -1, 4).float() for r in reg])

    def _loss(self, cls_l, reg_l, anchors_l, batch_data_samples):
        c = self.TRAIN
        anchors = torch.cat(anchors_l)
        cls = torch.cat(cls_l, dim=1)
        reg = torch.cat(reg_l, dim=1)
        loss_cls, loss_box, total = cls.new_zeros(()), cls.new_zeros(()), 0
        for i, sample in enumerate(batch_data_samples):
            gt = sample.gt_instances.bboxes.to(anchors)
            with torch.no_grad():
                assigned = assign_max_iou(anchors, gt, c["pos_iou_thr"], c["neg_iou_thr"], c["min_pos_iou"], c["match_low_quality"])
                pos, neg = random_sample(assigned, c["num"], c["pos_fraction"])
            total += pos.numel() + neg.numel()
            logits = torch.cat([cls[i, pos], cls[i, neg]])
            target = torch.cat([torch.ones_like(cls[i, pos]), torch.zeros_like(cls[i, neg])])
            loss_cls = loss_cls + F.binary_cross_entropy_with_logits(logits, target, reduction="sum")
            if pos.numel() > 0:
                t = bbox2delta(anchors[pos], gt[assigned[pos] - 1])
                loss_box = loss_box + (reg[i, pos] - t).abs().sum()
        total = max(total, 1)
        return dict(loss_rpn_cls=loss_cls / total, loss_rpn_bbox=loss_box / total + reg.sum() * 0)

    @torch.no_grad()
    def _proposals(self, cls_l, reg_l, anchors_l, img_shapes, cfg):
        out = []
        for i in range(cls_l[0].shape[0]):
            boxes, scores, lvls = [], [], []
            for l, (c, r, a) in enumerate(zip(cls_l, reg_l, anchors_l)):
                s = c[i].sigmoid()
                k = min(cfg["nms_pre"], s.numel())
                s, idx = s.topk(k)
                boxes.append(delta2bbox(a[idx], r[i, idx], max_shape=img_shapes[i]))
                scores.append(s)
                lvls.append(torch.full_like(idx, l))
            boxes, scores, lvls = torch.cat(boxes), torch.cat(scores), torch.cat(lvls)
            ok = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)          # min_bbox_size = 0
            boxes, scores, lvls = boxes[ok], scores[ok], lvls[ok]
            keep = ops.nms(boxes, scores, cfg["iou_threshold"], idxs=lvls, max_keep=cfg["max_per_img"])
            out.append(Instances(bboxes=boxes[keep], scores=scores[keep], labels=torch.zeros_like(keep)))
        return out

    def _run(self, x):
        cls, reg = self(x)
        sizes = [c.shape[2:] for c in cls]
        anchors = self.anchors.grid(sizes, cls[0].device)
        cls_l, reg_l = self._flatten(cls, reg)
        return cls_l, reg_l, anchors

    def loss_and_predict(self, x, batch_data_samples, proposal_cfg=None):
        cls_l, reg_l, anchors = self._run(x)
        losses = self._loss(cls_l, reg_l, anchors, batch_data_samples)
        shapes = [s.img_shape for s in batch_data_samples]
        return losses, self._proposals(cls_l, reg_l, anchors, shapes, proposal_cfg or self.PROPOSAL_TRAIN)

    def predict(self, x, batch_data_samples, rescale=False):
        cls_l, reg_l, anchors = self._run(x)
        return self._proposals(cls_l, reg_l, anchors, [s.img_shape for s in batch_data_samples], self.PROPOSAL_TEST)
