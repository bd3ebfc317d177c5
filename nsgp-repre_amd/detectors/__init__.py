from .faster_rcnn_roi_replay import FasterRCNNRoIReplay, RoIReplayModes, filter_pseudo_labels

__all__ = ["FasterRCNNRoIReplay", "RoIReplayModes", "filter_pseudo_labels"]
