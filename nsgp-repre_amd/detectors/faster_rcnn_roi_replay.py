"""``FasterRCNNRoIReplay`` -- the detector-side boundary of the hot path
(mmdet/models/detectors/faster_rcnn_roi_replay.py:14, forward :189-236).

Only the mode dispatch belongs to the path: ``mode='nullspace'`` (a loss pass without the
teacher, used under the covariance hooks by ``cal_fea_in``) and ``mode='roi_replay'`` (the RoI
feature dump used by ``cal_rois``).  The detector forward/backward itself is stock PyTorch-ROCm
(north_star), and the teacher pseudo-labelling inside ``loss`` (det:65-109) is SURVEY 8f-2
("next"), not built this round: ``loss`` here runs the stock two-stage loss.
"""
import torch.nn as nn

from ..registry import MODELS, register

try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.detectors.two_stage import TwoStageDetector as _Base
    _HAVE_MMDET = True
except Exception:
    _Base = nn.Module
    _HAVE_MMDET = False


class RoIReplayModes:
    """``forward(inputs, data_samples, mode)`` with the two extra modes of det:229-232."""

    def forward(self, inputs, data_samples=None, mode: str = "tensor"):
        if mode == "loss":
            return self.loss(inputs, data_samples)
        elif mode == "predict":
            return self.predict(inputs, data_samples)
        elif mode == "tensor":
            return self._forward(inputs, data_samples)
        elif mode == "nullspace":
            return self.loss(inputs, data_samples, use_teacher_student=False)
        elif mode == "roi_replay":
            return self.get_bbox_stuff(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')

    def get_bbox_stuff(self, batch_inputs, batch_data_samples):
        """det:146-186: features -> RPN proposals -> ``roi_head.get_bbox_stuff`` 6-tuple."""
        x = self.extract_feat(batch_inputs)
        rpn_results_list = self.rpn_head.predict(x, batch_data_samples, rescale=False)
        return self.roi_head.get_bbox_stuff(x, rpn_results_list, batch_data_samples)


@register(MODELS)
class FasterRCNNRoIReplay(RoIReplayModes, _Base):
    def __init__(self, *args, **kwargs):
        if _HAVE_MMDET:  # pragma: no cover
            super().__init__(*args, **kwargs)
        else:
            nn.Module.__init__(self)
            for k, v in kwargs.items():      # stand-alone: sub-modules are passed in ready-built
                setattr(self, k, v)
        self.rpn_thresh, self.roi_thresh = 0.5, 0.5   # set by the runner from rr_thresh (runner:439-440)

    def loss(self, batch_inputs, batch_data_samples, use_teacher_student=True):
        if not _HAVE_MMDET:
            raise RuntimeError("the stock two-stage loss needs mmdet")
        return _Base.loss(self, batch_inputs, batch_data_samples)  # pragma: no cover
