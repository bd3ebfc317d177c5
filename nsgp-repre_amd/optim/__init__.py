from .adam_nscl import AdamNSCL, AdamWNSCL
from .sgd_nscl import SGDNSCL, SGDNSCLNA

__all__ = ["SGDNSCL", "SGDNSCLNA", "AdamWNSCL", "AdamNSCL"]
