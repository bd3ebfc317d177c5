from .prototype_bank import build_prototype_bank, select_class_prototypes
from .replay_head import (PrototypeReplay, StandardMultiPrototypeReplayHead, StandardPrototypeReplayHead,
                          StandardRoIReplayHead, get_work_dir)
from .roi_dump import select_five_rois
from .task_bbox_head import ConvFCBBoxHeadTask, Shared2FCBBoxHeadTask

__all__ = ["build_prototype_bank", "select_class_prototypes", "PrototypeReplay", "StandardMultiPrototypeReplayHead",
           "ConvFCBBoxHeadTask", "Shared2FCBBoxHeadTask", "StandardPrototypeReplayHead", "StandardRoIReplayHead", "get_work_dir",
           "select_five_rois"]
