"""Coarse + fine prototype bank construction (RePRE).

Mirror of the bank-building block of ``StandardMultiPrototypeReplayHead.__init__``
(mmdet/models/roi_heads/standard_roi_replay_head.py:397-452).  Per old class:

* coarse prototype = mean of the class's RoI features (``:413``);
* cosine Gram >= 0.6 -> neighbour counts (``:417-421``) -- HIP kernel ``repre_sim_counts``:
  the N x N similarity never leaves the MFMA accumulators, the mask is bit-packed;
* rows in the bottom third by count are ineligible as centres (``:422-423``);
* greedy cover, at most ``max_prototype - 1`` fine prototypes, each the mean of the rows
  its centre's mask selects (``:430-446``) -- HIP kernel ``repre_masked_mean``.

The visiting order of rows comes from ``counts.sort(descending=True)`` on the HOST, i.e. the
very torch-CPU call the reference's CPU path makes (its unstable tie order included), on N
int64 values copied back from the GPU.  The greedy loop itself is O(9*N) bit operations on
the host and fetches at most 9 mask rows per class from the GPU.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops

SIM_THRESHOLD = 0.6  # standard_roi_replay_head.py:420


def _bits_to_bool(words_row: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(words_row.view(np.uint8), bitorder="little")[:n].astype(bool)


def select_class_prototypes(Fc: torch.Tensor, max_proto: int = 10, saved_masks: Optional[List[torch.Tensor]] = None,
                            order: Optional[torch.Tensor] = None):
    """One class.  ``Fc``: [N x D] fp32 GPU tensor.  Returns
    (coarse [1xD], fine list of [1xD], masks list of bool[N] CPU tensors, centre ids)."""
    N = Fc.shape[0]
    Fc = Fc.reshape(N, -1).contiguous()
    coarse = ops.masked_mean(Fc)
    counts, bitmask = ops.sim_counts(Fc, SIM_THRESHOLD)
    counts_cpu = counts.cpu()
    if order is None:
        cnt_sorted, order = counts_cpu.sort(dim=-1, descending=True)  # the reference's own call (:421)
    else:
        cnt_sorted = counts_cpu[order]
    thr_cnt = int(cnt_sorted[-N // 3])        # python precedence: (-N)//3  (:422)
    covered = (counts_cpu <= thr_cnt).numpy().copy()   # `potential_center` in the reference (:423)
    masks = list(saved_masks) if saved_masks is not None else []
    fine, centres = [], []
    order_l = order.tolist()
    for pc in range(max_proto - 1):
        for id_ in order_l:
            if pc < len(masks):
                m_bool = masks[pc].cpu().numpy().astype(bool)
                words = ops.pack_bool_mask(masks[pc]).to(Fc.device)
                cid = -1
            else:
                if covered[id_]:
                    continue
                words = bitmask[id_].contiguous()
                m_bool = _bits_to_bool(words.cpu().numpy(), N)
                masks.append(torch.from_numpy(m_bool.copy()))
                cid = id_
            covered |= m_bool
            fine.append(ops.masked_mean(Fc, words, int(m_bool.sum())))
            centres.append(cid)
            break
    return coarse, fine, masks, centres


def build_prototype_bank(feats: torch.Tensor, cls_targets: torch.Tensor, task_split: Sequence[int], task_id: int,
                         max_proto: int = 10, saved: Optional[list] = None, orders: Optional[dict] = None,
                         select_fn=None, shard: bool = True):
    """All old classes ``range(task_split[0], task_split[task_id-1])`` (head:405-449).
    Returns (bank [K x D], labels int64 [K], masks per class, centre ids per class).

    Under an initialised process group (``shard=True``) the classes -- independent units, SURVEY 8e -- are spread over the
    ranks by ``shard_by_cost`` with cost N_c^2 (the similarity Gram dominates: a 20,000-RoI COCO class is 10 TFLOP, a
    300-RoI VOC class 2 GFLOP), each rank builds its own classes, and one ragged all-gather of the bank rows plus one object
    all-gather of the small per-class records (masks, centre ids) gives every rank the complete bank, identical to the
    single-process result (the reference rebuilds every class on every rank).  ``select_fn`` replaces the per-class builder
    (the CPU tests pass the oracle's)."""
    from ..runner import dist as D
    select = select_fn or select_class_prototypes
    classes = list(range(task_split[0], task_split[task_id - 1]))
    rows = {c: (cls_targets == c) for c in classes}
    world, rank = (D.get_world_size(), D.get_rank()) if shard else (1, 0)
    owner = D.shard_by_cost([float(int(rows[c].sum())) ** 2 for c in classes], world) if world > 1 else [0] * len(classes)
    local = {}
    for c, o in zip(classes, owner):
        if o != rank:
            continue
        sm = saved[c] if (saved is not None and c < len(saved)) else None
        coarse, fine, masks, centres = select(feats[rows[c]], max_proto, saved_masks=sm, order=None if orders is None else orders.get(c))
        local[c] = (torch.cat([coarse] + list(fine), dim=0), masks, centres)
    if world > 1:
        import torch.distributed as dist
        mine = [c for c in classes if c in local]
        D_feat = feats.shape[1] if feats.dim() == 2 else int(np.prod(feats.shape[1:]))
        payload = torch.cat([local[c][0] for c in mine], dim=0) if mine else feats.new_zeros((0, D_feat))
        gathered = D.all_gather_different_shape(payload)                      # one ragged all-gather of the bank rows
        records = [None] * world
        dist.all_gather_object(records, [(c, int(local[c][0].shape[0]), [m.cpu() for m in local[c][1]], list(local[c][2])) for c in mine])
        for r, (rec, rows_r) in enumerate(zip(records, gathered)):
            off = 0
            for c, k, masks, centres in rec:
                local[c] = (rows_r[off:off + k].to(feats.device), masks, centres)
                off += k
    bank, labels, all_masks, all_centres = [], [], [], []
    for c in classes:
        b, masks, centres = local[c]
        bank.append(b)
        labels += [c] * int(b.shape[0])
        all_masks.append(masks)
        all_centres.append(centres)
    bank = torch.cat(bank, dim=0)
    return bank, torch.tensor(labels, dtype=torch.long, device=bank.device), all_masks, all_centres
