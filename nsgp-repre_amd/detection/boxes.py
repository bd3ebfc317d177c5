"""Box arithmetic of the stock recipe: IoU, the DeltaXYWH coder, the anchor grid
(faster-rcnn_r50_fpn.py: AnchorGenerator scales=[8] ratios=[0.5,1,2] strides=[4..64];
DeltaXYWHBBoxCoder means 0, stds 1 (RPN) / [0.1,0.1,0.2,0.2] (RoI head))."""
import math

import torch


def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[N x 4], [M x 4] xyxy -> IoU [N x M]."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp(min=1e-6)


def bbox2delta(proposals: torch.Tensor, gt: torch.Tensor, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)) -> torch.Tensor:
    px, py = (proposals[:, 0] + proposals[:, 2]) * 0.5, (proposals[:, 1] + proposals[:, 3]) * 0.5
    pw, ph = proposals[:, 2] - proposals[:, 0], proposals[:, 3] - proposals[:, 1]
    gx, gy = (gt[:, 0] + gt[:, 2]) * 0.5, (gt[:, 1] + gt[:, 3]) * 0.5
    gw, gh = gt[:, 2] - gt[:, 0], gt[:, 3] - gt[:, 1]
    d = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return (d - d.new_tensor(means)) / d.new_tensor(stds)


def delta2bbox(rois: torch.Tensor, deltas: torch.Tensor, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip: float = 16 / 1000) -> torch.Tensor:
    """rois [N x 4], deltas [N x 4k] -> boxes [N x 4k]."""
    n = deltas.shape[0]
    d = deltas.reshape(n, -1, 4).float() * deltas.new_tensor(stds, dtype=torch.float32) + deltas.new_tensor(means, dtype=torch.float32)
    pxy = ((rois[:, :2] + rois[:, 2:]) * 0.5)[:, None, :]
    pwh = (rois[:, 2:] - rois[:, :2])[:, None, :]
    max_ratio = abs(math.log(wh_ratio_clip))
    dwh = d[..., 2:].clamp(min=-max_ratio, max=max_ratio)
    gxy = pxy + pwh * d[..., :2]
    gwh = pwh * dwh.exp()
    out = torch.cat([gxy - gwh * 0.5, gxy + gwh * 0.5], dim=-1)
    if max_shape is not None:
        out[..., 0::2] = out[..., 0::2].clamp(min=0, max=max_shape[1])
        out[..., 1::2] = out[..., 1::2].clamp(min=0, max=max_shape[0])
    return out.reshape(n, -1)


class AnchorGenerator:
    def __init__(self, strides=(4, 8, 16, 32, 64), ratios=(0.5, 1.0, 2.0), scales=(8,)):
        self.strides = list(strides)
        self.base = []
        for s in self.strides:
            r = torch.tensor(ratios, dtype=torch.float32)
            sc = torch.tensor(scales, dtype=torch.float32)
            ws = (s * (1.0 / r.sqrt())[:, None] * sc[None, :]).reshape(-1)
            hs = (s * r.sqrt()[:, None] * sc[None, :]).reshape(-1)
            self.base.append(torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], dim=-1))
        self.num_base = self.base[0].shape[0]
        self._cache = {}

    def grid(self, featmap_sizes, device):
        """Per level [H*W*A x 4], location-major / anchor-minor -- the order of a [A, H, W] -> [H, W, A] score map."""
        key = (tuple(tuple(s) for s in featmap_sizes), str(device))
        if key not in self._cache:
            out = []
            for (h, w), s, base in zip(featmap_sizes, self.strides, self.base):
                sx = torch.arange(w, device=device, dtype=torch.float32) * s
                sy = torch.arange(h, device=device, dtype=torch.float32) * s
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack([xx, yy, xx, yy], dim=-1).reshape(-1, 1, 4)
                out.append((shifts + base.to(device)[None]).reshape(-1, 4))
            self._cache[key] = out
        return self._cache[key]
