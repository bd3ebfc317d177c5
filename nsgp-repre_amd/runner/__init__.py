from . import dist, ewc
from .br_nullspace_runner import BRNullSpaceRunner, NullSpaceTaskMixin
from .nullspace import (CovarianceCollector, cal_fea_in, cal_rois, find_checkpoint, full_ignore_keys, should_ignore,
                        update_optim_transforms, wire_param_names)

__all__ = ["dist", "ewc", "BRNullSpaceRunner", "NullSpaceTaskMixin", "CovarianceCollector", "cal_fea_in", "cal_rois", "find_checkpoint", "full_ignore_keys",
           "should_ignore", "update_optim_transforms", "wire_param_names"]
