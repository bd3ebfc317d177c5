from .faster_rcnn_roi_replay import FasterRCNNRoIReplay, RoIReplayModes

__all__ = ["FasterRCNNRoIReplay", "RoIReplayModes"]
