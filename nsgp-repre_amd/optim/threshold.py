"""Elbow (rank) selection on a descending spectrum -- host-side, like the reference
(mmdet/engine/optimizers/SGD_NSCL.py:98-177 and AdamW_NSCL.py:105-133 use numpy +
``scipy.ndimage.gaussian_filter1d``).  It decides an INTEGER from a float spectrum, so it is
kept on the host in the same library calls: the smoothing is scipy's own routine."""
import numpy as np
import scipy.ndimage

SMOOTH_MIN_LEN = 128   # shorter spectra skip the smoothing (SGD_NSCL.py:137,152)
SMOOTH_SIGMA = 10
EDGE_DROP = 0.03       # fraction of points ignored, half on each side (SGD_NSCL.py:146-149)


def curvature_peak_value(points: np.ndarray) -> float:
    """Spectrum value at the peak of the second forward difference."""
    n = len(points)
    curve = scipy.ndimage.gaussian_filter1d(points, sigma=SMOOTH_SIGMA) if n >= SMOOTH_MIN_LEN else points
    slope = curve[:-1] - curve[1:]
    bend = slope[:-1] - slope[1:]
    if n >= SMOOTH_MIN_LEN:
        drop = int(n * EDGE_DROP / 2)
        assert n - drop >= 10
        bend = bend[drop:-drop]
    return points[int(np.argmax(bend)) + int((n - len(bend)) / 2)]


def elbow_index(points: np.ndarray, offset: float = 0.0, rule: str = "sgd") -> int:
    """``i_thres``: basis columns ``[i_thres, D)`` span the null space.

    ``rule='sgd'``  (SGD_NSCL.py:164-170, also standard_roi_replay_head.py:301-330):
        ``-1 <= offset <= 1`` shifts by ``int(offset * i)``, anything else by ``int(offset)``.
    ``rule='adam'`` (AdamW_NSCL.py:124-127, Adam_NSCL.py same):
        ``-1 < offset < 1`` shifts by ``int(offset * (D - i))``, anything else by ``int(offset)``.
    """
    points = np.asarray(points)
    assert points.ndim == 1
    n = len(points)
    cut = curvature_peak_value(points)
    i = int(np.flatnonzero(points >= cut).max())
    if rule == "sgd":
        shift = int(offset * i) if -1 <= offset <= 1 else int(offset)
    elif rule == "adam":
        shift = int(offset * (n - i)) if -1 < offset < 1 else int(offset)
    else:
        raise ValueError(f"unknown offset rule {rule!r}")
    return max(0, min(i + shift, n - 1))
