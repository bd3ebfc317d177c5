"""Stock two-stage detector parts in plain PyTorch-ROCm, for running the hot path end to end WITHOUT mmdet.

north_star: "the detector forward/backward runs on PyTorch-ROCm".  In the reference these parts come
from MMDetection / MMCV (ResNet, FPN, RPNHead, StandardRoIHead, RoIAlign, nms, assigners, samplers) and
the fork only plugs its optimizer, runner and RoI head into them.  Neither package exists in this image,
so this subpackage restates the published Faster R-CNN recipe of the reference's
``cl_faster_rcnn_cfgs/_base_/models/faster-rcnn_r50_fpn.py`` with the SAME module and parameter names
(``backbone.layer3.0.conv2.weight``, ``neck.fpn_convs.1.conv.weight`` ...) so that the names wiring,
ignore keys, covariance hooks and projector tables of the NSGP side see the real thing.  It is the
measurement harness for the end-to-end img/s of SURVEY 8(d) and the vehicle of the two-task test;
where mmdet is installed the registered classes inherit mmdet's and none of this is used.

Everything here is torch plumbing (MIOpen convolutions, hipBLASLt GEMMs) except NMS, which the teacher's
per-step ``predict`` needs and torch does not have: ``ops.nms`` (csrc/nms.hip).
"""
from .structures import DetSample, Instances
from .boxes import AnchorGenerator, bbox2delta, box_iou, delta2bbox
from .backbone import FPN, ResNet
from .rpn_head import RPNHead
from .roi_parts import RoIAlignExtractor, StandaloneRoIHead, assign_max_iou, random_sample
from .build import build_faster_rcnn, relocate_segment_final_weights, synthetic_batch

__all__ = ["DetSample", "Instances", "AnchorGenerator", "bbox2delta", "box_iou", "delta2bbox", "FPN", "ResNet", "RPNHead",
           "RoIAlignExtractor", "StandaloneRoIHead", "assign_max_iou", "random_sample", "build_faster_rcnn",
           "synthetic_batch", "relocate_segment_final_weights"]
