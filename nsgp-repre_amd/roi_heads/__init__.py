from .prototype_bank import build_prototype_bank, select_class_prototypes
from .replay_head import PrototypeReplay, StandardMultiPrototypeReplayHead, get_work_dir
from .task_bbox_head import Shared2FCBBoxHeadTask

__all__ = ["build_prototype_bank", "select_class_prototypes", "PrototypeReplay", "StandardMultiPrototypeReplayHead",
           "Shared2FCBBoxHeadTask", "get_work_dir"]
