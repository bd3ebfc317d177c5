"""Per-task classification / regression heads (``ConvFCBBoxHeadTask`` / ``Shared2FCBBoxHeadTask``,
mmdet/models/roi_heads/bbox_heads/convfc_bbox_head_task.py:14-288, 516-529).

The task-specific part is small: one ``fc_cls`` Linear per task + one background Linear, one
``fc_reg`` Linear per task; heads of FUTURE tasks see a detached input and their logits are
forced to -inf (regression to 0) so that ``softmax`` ignores unseen classes (``:259-272``).  The
GEMMs stay on PyTorch-ROCm (hipBLASLt) -- SURVEY K8.

When mmdet is importable the class also inherits its ``BBoxHead`` (losses, targets, bbox coder)
so the reference configs build it unchanged; without mmdet it is a plain ``nn.Module`` that
carries exactly the forward path the replay loss needs.
"""
from typing import Sequence

import torch
import torch.nn as nn

from ..registry import MODELS, register

try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.roi_heads.bbox_heads.bbox_head import BBoxHead as _Base
    _HAVE_MMDET = True
except Exception:
    _Base = nn.Module
    _HAVE_MMDET = False


class TaskHeadMixin:
    """Builds and runs the per-task predictors on top of ``x_cls`` / ``x_reg`` features."""

    def _build_task_predictors(self, cls_in: int, reg_in: int, num_classes: int, task_split: Sequence[int],
                               task_id: int, reg_class_agnostic: bool = False, box_dim: int = 4,
                               extra_channels: int = 1):
        self.task_split = list(task_split)
        self.task_id = task_id
        self.reg_class_agnostic = reg_class_agnostic
        self.fc_cls = nn.ModuleList()
        for i in range(1, len(self.task_split)):
            self.fc_cls.append(nn.Linear(cls_in, self.task_split[i] - self.task_split[i - 1]))
        self.fc_cls.append(nn.Linear(cls_in, extra_channels))            # background (:108-112)
        self.background_nums = extra_channels
        self.fc_reg = nn.ModuleList()
        if reg_class_agnostic:
            self.fc_reg.append(nn.Linear(reg_in, box_dim))
        else:
            for i in range(1, len(self.task_split)):
                self.fc_reg.append(nn.Linear(reg_in, box_dim * (self.task_split[i] - self.task_split[i - 1])))
        # freeze the heads of tasks that have not arrived yet (:130-144); background stays live
        for i, m in enumerate(self.fc_cls):
            m.requires_grad_(i + 1 <= self.task_id or i + 1 == len(self.fc_cls))
        for i, m in enumerate(self.fc_reg):
            m.requires_grad_(i + 1 <= self.task_id or self.reg_class_agnostic)

    def _task_predict(self, x_cls: torch.Tensor, x_reg: torch.Tensor):
        preds = []
        for i, module in enumerate(self.fc_cls):
            future = (i + 1 > self.task_id) and (i + 1 != len(self.task_split))
            o = module(x_cls.detach() if future else x_cls)
            if future:
                o = torch.full_like(o, float("-inf"))     # exp(-inf) = 0: unseen classes vanish (:262-264)
            preds.append(o)
        cls_score = torch.cat(preds, dim=-1)
        preds = []
        for i, module in enumerate(self.fc_reg):
            future = (i + 1 > self.task_id) and not self.reg_class_agnostic
            o = module(x_reg.detach() if future else x_reg)
            if future:
                o = torch.zeros_like(o)
            preds.append(o)
        return cls_score, torch.cat(preds, dim=-1)


@register(MODELS)
class Shared2FCBBoxHeadTask(TaskHeadMixin, _Base):
    """flatten -> (Linear + ReLU) x 2 -> per-task cls || per-task reg  (Shared2FC shape,
    convfc_bbox_head_task.py:516-529 + forward :209-288 + get_mid_features :290-323)."""

    def __init__(self, in_channels: int = 256, fc_out_channels: int = 1024, roi_feat_size: int = 7,
                 num_classes: int = 80, task_split: Sequence[int] = (0, 10, 20), task_id: int = 1,
                 reg_class_agnostic: bool = False, **kwargs):
        if _HAVE_MMDET:  # pragma: no cover
            super().__init__(in_channels=in_channels, roi_feat_size=roi_feat_size, num_classes=num_classes,
                             reg_class_agnostic=reg_class_agnostic, **kwargs)
        else:
            nn.Module.__init__(self)
            self.in_channels, self.num_classes = in_channels, num_classes
        self.fc_out_channels = fc_out_channels
        self.roi_feat_area = roi_feat_size * roi_feat_size
        self.shared_fcs = nn.ModuleList([nn.Linear(in_channels * self.roi_feat_area, fc_out_channels),
                                         nn.Linear(fc_out_channels, fc_out_channels)])
        self.relu = nn.ReLU(inplace=True)
        self._build_task_predictors(fc_out_channels, fc_out_channels, num_classes, task_split, task_id,
                                    reg_class_agnostic)
        self.null_space = False

    def get_mid_features(self, x: torch.Tensor) -> torch.Tensor:
        """RoI features flattened, before the shared FCs: [N, 7*7*256] (:290-323)."""
        return x.flatten(1)

    def forward(self, x):
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = self.relu(fc(x))
        return self._task_predict(x, x)


# the reference's general class name resolves to the same implementation for the Shared2FC shape
register(MODELS, "ConvFCBBoxHeadTask")(Shared2FCBBoxHeadTask)
