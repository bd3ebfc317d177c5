"""The identity behind the covariance pass's correlation form (csrc/covariance.hip), in fp64 on the CPU: for a 3x3 / stride 1 / padding 1
convolution the reference's X^T X (compute_cov, nsrunner_roi_replay.py:876-916: X = unfold of the batch mean) equals
    Cov_ext - Cov_ring,
Cov_ext[(c1,ky1,kx1),(c2,ky2,kx2)] = R[ky2-ky1, kx2-kx1][c1,c2], R[d][c1,c2] = sum_q X[c1][q] X[c2][q + d] (zero outside the image), and
Cov_ring = the covariance of the ring of positions y in {-1, H} / x in {-1, W}, which lives in four [3C x 3C] strip covariances.
The GPU tests check the kernels against the oracle; this one checks the algebra they implement, with the same index conventions."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _strip(x, H, W, which):
    """[3C x 3C] covariance of one strip: rows (c, k), k the free tap index; operand exactly as nsgp_cov_group_split_kernel builds it."""
    C = x.shape[0]
    L = W + 2 if which < 2 else H
    op = np.zeros((3 * C, L))
    for c in range(C):
        for k in range(3):
            for l in range(L):
                if which < 2:
                    iy, ix = (0 if which == 0 else H - 1), l + k - 2
                else:
                    iy, ix = l + k - 1, (0 if which == 2 else W - 1)
                if 0 <= iy < H and 0 <= ix < W:
                    op[3 * c + k, l] = x[c, iy, ix]
    return op @ op.T


@pytest.mark.parametrize("C,H,W", [(2, 4, 5), (3, 6, 4), (1, 3, 3), (2, 1, 7)])
def test_correlation_form_identity_fp64(C, H, W):
    rs = np.random.default_rng(C * 100 + H * 10 + W)
    x = rs.standard_normal((C, H, W))
    X = F.unfold(torch.from_numpy(x)[None], 3, padding=1)[0].t().numpy()      # [L x 9C], rows (c, ky, kx) as torch orders them
    cov = X.T @ X
    # flat zero-bordered image at pitch Wq >= W + 4, two zero rows above and below, and the shifted correlations as offsets in it
    Wq = W + 4
    flat = np.zeros((C, (H + 4) * Wq + 2 * (2 * Wq + 4)))
    base = 2 * Wq + 4                                                           # slack in front for negative offsets
    for r in range(H):
        flat[:, base + (r + 2) * Wq + 2: base + (r + 2) * Wq + 2 + W] = x[:, r]
    n = (H + 4) * Wq

    def R(dy, dx):
        off = dy * Wq + dx
        return flat[:, base: base + n] @ flat[:, base + off: base + off + n].T

    D = 9 * C
    s = [_strip(x, H, W, k) for k in range(4)]
    out = np.zeros((D, D))
    for d1 in range(D):
        c1, t1 = divmod(d1, 9)
        ky1, kx1 = divmod(t1, 3)
        for d2 in range(D):
            c2, t2 = divmod(d2, 9)
            ky2, kx2 = divmod(t2, 3)
            v = R(ky2 - ky1, kx2 - kx1)[c1, c2]
            if ky1 == 2 and ky2 == 2:
                v -= s[0][3 * c1 + kx1, 3 * c2 + kx2]
            if ky1 == 0 and ky2 == 0:
                v -= s[1][3 * c1 + kx1, 3 * c2 + kx2]
            if kx1 == 2 and kx2 == 2:
                v -= s[2][3 * c1 + ky1, 3 * c2 + ky2]
            if kx1 == 0 and kx2 == 0:
                v -= s[3][3 * c1 + ky1, 3 * c2 + ky2]
            out[d1, d2] = v
    assert np.abs(out - cov).max() <= 1e-12 * max(1.0, np.abs(cov).max())
    assert np.allclose(R(1, -2), R(-1, 2).T)                                    # R[-d] = R[d]^T: 13 products instead of 25
