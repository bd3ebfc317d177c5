"""bench.py -- NSGP-RePRE on MI355X (see DESIGN.md section 'Measurement').

One "step" = ONE TRAINING ITERATION of BASELINE.json configs[1] (Faster R-CNN R-50-FPN, VOC 15+5 task 2, 1 synthetic
3x800x1344 image per GPU): teacher predict + pseudo-label filter, student forward (RPN + RoI losses + RePRE replay loss on the
K=150 prototype bank), backward, SGDNSCL.step (50 projected layers; the dense form is 118.3 GFLOP of projection against
0.584 GB of projectors) and zero_grad.  The projectors come from SURVEY 8d's seeded covariances through the product's own
get_eigens-style pipeline (eigh -> elbow -> set_basis), so the step runs the DEFAULT path: head-form projectors applied in the
low-rank form p += c (u - (u U) U^T), which makes the step HBM-bound.  `value` = whole-job training img/s over all ranks
(BASELINE's metric); `nsgp_step_ms` = the HIP launches of the projected optimizer step inside that very loop (HIP events recorded
by the library on the launch stream around every launch); `roofline` is the step's longest launch -- nsgp_update_lr_kernel, the
projected layers' elementwise update fused with T = u U, HBM-bound -- from its average duration over the same K timed steps, with
the other launches of the step as blocks beside it and the dense-GEMM kernel (still what externally assigned projectors run on)
under `hot_path`.

Multi-GPU: one process per GPU (torchrun); the detector is wrapped in DistributedDataParallel, so the bucketed RCCL all-reduce
of the 41.5 M fp32 gradients (overlapped with backward) IS inside the timed region for N > 1; the projected step itself is
replicated (identical gradients after the all-reduce, identical projectors) and has no exchange step of its own.

Beside the headline the same JSON line carries, measured in the same process (rank 0, N = 1 only, never part of `value`):
`hot_path` (the fork's additions alone -- SGDNSCL.step + replay loss on synthetic gradients: the default low-rank form, every
MFMA path of the dense projection, the AdamW flavour), `once_per_task` (covariance forward for R-50 and R-101, the 50-layer
get_eigens + get_transforms sweep, prototype-bank builds at three sizes, the R-101 step) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def r50_fpn_voc_parameter_table():
    """(name, shape, projected) for Faster R-CNN R-50-FPN with a 20-class task head -- the
    arithmetic of cl_faster_rcnn_cfgs/_base_/models/faster-rcnn_r50_fpn.py; frozen_stages=1
    drops conv1/bn1/layer1 from the optimizer; ignore_keys=['rpn','roi_head'] leaves their
    tensors un-projected (cl_faster_rcnn_nsgp_repre_15_5_2.py:18,39)."""
    import nsgp_oracle as O
    table = []
    for n, cout, D in O.resnet_fpn_projected_layers(50):
        k = 3 if (("conv2" in n) or ("fpn_convs" in n)) else 1
        table.append((n, (cout, D // (k * k), k, k), True))
        if n.startswith("backbone"):
            bn = n.replace("conv", "bn").replace("downsample.0", "downsample.1").replace(".weight", "")
            table.append((bn + ".weight", (cout,), False))
            table.append((bn + ".bias", (cout,), False))
        else:
            table.append((n.replace(".weight", ".bias"), (cout,), False))
    table += [("rpn_head.rpn_conv.weight", (256, 256, 3, 3), False), ("rpn_head.rpn_conv.bias", (256,), False),
              ("rpn_head.rpn_cls.weight", (3, 256, 1, 1), False), ("rpn_head.rpn_cls.bias", (3,), False),
              ("rpn_head.rpn_reg.weight", (12, 256, 1, 1), False), ("rpn_head.rpn_reg.bias", (12,), False)]
    return table


def r50_fpn_hooked_convs(H=800, W=1344, depth=50):
    """(name, cin, k, stride, pad, Hin, Win) of the convs cal_fea_in hooks on R-50-FPN (61) / R-101-FPN (112) for one
    800x1344 padded image (ignore_keys drop rpn/roi_head): sum 2*L*D^2 = 1.86 / 2.77 TFLOP."""
    out = [("backbone.conv1", 3, 7, 2, 3, H, W)]
    h, w, inpl = H // 4, W // 4, 64
    for li, (nb, planes, stride) in enumerate(((3, 64, 1), (4, 128, 2), (6 if depth == 50 else 23, 256, 2), (3, 512, 2)), start=1):
        for b in range(nb):
            st = stride if b == 0 else 1
            pre = f"backbone.layer{li}.{b}"
            out.append((pre + ".conv1", inpl, 1, 1, 0, h, w))
            out.append((pre + ".conv2", planes, 3, st, 1, h, w))
            h2, w2 = (h + 2 - 3) // st + 1, (w + 2 - 3) // st + 1
            out.append((pre + ".conv3", planes, 1, 1, 0, h2, w2))
            if b == 0:
                out.append((pre + ".downsample.0", inpl, 1, st, 0, h, w))
            inpl, h, w = planes * 4, h2, w2
    res = [(H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    for i, (c, (hh, ww)) in enumerate(zip((256, 512, 1024, 2048), res)):
        out.append((f"neck.lateral_convs.{i}.conv", c, 1, 1, 0, hh, ww))
    for i, (hh, ww) in enumerate(res):
        out.append((f"neck.fpn_convs.{i}.conv", 256, 3, 1, 1, hh, ww))
    return out


def _covariance_forward_ms(dev, depth, only_grouped=False):
    from nsgp_repre_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    layers = r50_fpn_hooked_convs(depth=depth)
    acts, covs, ws_bytes, ref_flops = {}, {}, 0, 0.0
    for n, cin, k, s, p, h, w in layers:
        if (cin, h, w) not in acts:
            acts[(cin, h, w)] = torch.randn(1, cin, h, w, device=dev, generator=g).abs()
        ws_bytes = max(ws_bytes, ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p)))
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        ref_flops += 2.0 * ho * wo * (cin * k * k) ** 2
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    from nsgp_repre_amd.runner.nullspace import CovarianceStreams
    side = CovarianceStreams(4)      # what CovarianceCollector uses by default: layer i on side stream i % 4

    def forward(streams):
        for slot, (n, cin, k, s, p, h, w) in enumerate(layers):
            x = acts[(cin, h, w)]
            if streams:
                nb = ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p))
                covs[n] = side.run(slot, x, lambda wsf, x=x, k=k, s=s, p=p, n=n, nb=nb: ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), covs.get(n), wsf(nb)))
            else:
                covs[n] = ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), covs.get(n), ws)
        if streams:
            side.join()
    # the grouped pass (what CovarianceCollector runs since round 3): every layer with D % 64 == 0 in ONE plan run (five launches),
    # the rest (the 7x7 stem) at hook time on a side stream
    geoms = [(1, cin, h, w, (k, k), (s, s), (p, p)) for n, cin, k, s, p, h, w in layers]
    plan = ops.CovGroupPlan(geoms, dev)
    gcov = [None] * len(layers)

    def forward_grouped():
        nonlocal gcov
        for slot, ((n, cin, k, s, p, h, w), ok) in enumerate(zip(layers, plan.routes)):
            if not ok:
                x = acts[(cin, h, w)]
                nb = ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p))
                covs[n] = side.run(slot, x, lambda wsf, x=x, k=k, s=s, p=p, n=n, nb=nb: ops.cov_accumulate_conv2d(x, (k, k), (s, s), (p, p), covs.get(n), wsf(nb)))
        gcov = plan.run([acts[(cin, h, w)] for n, cin, k, s, p, h, w in layers], gcov)
        side.join()
    out = []
    for fn in ((forward_grouped,) * 3 if only_grouped else (forward_grouped, lambda: forward(True), lambda: forward(False))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append(sorted(ts)[1])
    stats = dict(grouped_layers=plan.n_grouped, tiles=plan.n_tiles, upper_triangle_flops=plan.upper_flops, workspace_gb=plan.workspace_bytes / 1e9,
                 layers_in_correlation_form=plan.n_correlation_form, tile_steps=plan.tile_steps)
    plan.close()
    return out[0], out[1], out[2], ref_flops, len(layers), stats


def _bank_build(dev, n_classes, n_per_class, seed):
    """SURVEY 8d synthetic RoIs: per class 4 ReLU'd cluster centres + 0.6 * noise, ReLU'd; wall time of the whole build
    (similarity kernel, host-side greedy cover, masked means) and the similarity work by the reference's count 2 N^2 12544."""
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    g = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.relu(torch.randn(n_classes, 4, 12544, device=dev, generator=g))
    which = torch.randint(0, 4, (n_classes, n_per_class), device=dev, generator=g)
    feats = torch.empty(n_classes * n_per_class, 12544, device=dev)
    for c in range(n_classes):      # class by class: the 20,000-row class is 1 GB on its own
        feats[c * n_per_class:(c + 1) * n_per_class] = torch.relu(centres[c][which[c]] + 0.6 * torch.randn(n_per_class, 12544, device=dev, generator=g))
    cls = torch.arange(n_classes, device=dev).repeat_interleave(n_per_class)
    split = [0, n_classes, n_classes + 5]
    if n_per_class <= 2000:
        build_prototype_bank(feats, cls, split, 2, 10)       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bank, labels, _, _ = build_prototype_bank(feats, cls, split, 2, 10)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    flops = n_classes * 2.0 * n_per_class * n_per_class * 12544
    return {"classes": n_classes, "rois_per_class": n_per_class, "build_ms": ms, "bank_rows": int(bank.shape[0]),
            "similarity_reference_flops": flops, "tflops_by_reference_flops": flops / (ms * 1e-3) / 1e12}


def once_per_task_units(N, dev):
    """The once-per-task units of work of SURVEY 8d, each timed on its own (never part of `value`)."""
    out = {}
    for depth in (50, 101):
        ms, ms4, ms1, ref_flops, n, st = _covariance_forward_ms(dev, depth)
        up = st["upper_triangle_flops"]
        out[f"covariance_forward_r{depth}"] = {
            "ms": ms, "ms_hook_time_4_streams": ms4, "ms_hook_time_single_stream": ms1, "hooked_convs": n, **st,
            "reference_flops": ref_flops, "tflops_by_reference_flops": ref_flops / (ms * 1e-3) / 1e12,
            "roofline": {"bound": "mfma", "achieved": up / (ms * 1e-3) / 1e12, "peak": PEAK_16BIT_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": up / (ms * 1e-3) / 1e12 / PEAK_16BIT_MATRIX_TFLOPS,
                         "executed_mfma_flops": st["tile_steps"] * (256 * 128 * 32 * 2 * 3.0),
                         "executed_mfma_utilisation": st["tile_steps"] * (256 * 128 * 32 * 2 * 3.0) / (ms * 1e-3) / 1e12 / PEAK_16BIT_MATRIX_TFLOPS,
                         "note": "algorithmic = what the reference's X^T X needs: the upper triangles (diagonal blocks whole) of the grouped layers, sum L D (D + 128) FLOP, "
                                 "over the WHOLE hooked forward (mean / amax / operand split / SYRK / assemble + the stem at hook time).  executed = the k32 steps of "
                                 "256 x 128 tiles the SYRK launch runs x three fp16 MFMA products per fp32-equivalent product: the 3x3 / stride 1 / padding 1 layers "
                                 "on the large maps run in the correlation form (13 shifted C x C products instead of 40.5 blocks), so frac can exceed what the "
                                 "tile's own rate would give on the plain upper triangle"},
            "note": "one hooked forward at 800x1344 (fp32 activations), grouped pass: all layers with D % 64 == 0 in one plan run (mean, amax, operand split, "
                    "one SYRK tile table, ordered reduces of the long contractions, R reduce + assemble of the correlation-form layers), the 7x7 stem at hook "
                    "time on a side stream; ms_hook_time_* = round 2's per-layer launches (im2col form)"}
    torch.cuda.empty_cache()
    # a5 -> a7: spectra and projectors of all 50 projected layers from SURVEY 8d's seeded covariances (runner:635-662)
    import nsgp_oracle as O
    layers = O.resnet_fpn_projected_layers(50)
    params, names, fea_in = [], [], {}
    for idx, (n, cout, D) in enumerate(layers):
        params.append(torch.nn.Parameter(torch.empty(cout, D, device=dev)))
        names.append(n)
        gen = torch.Generator(device=dev).manual_seed(2000 + idx)
        X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
        fea_in[n] = (X.t() @ X).contiguous()
        del X
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    opt.param_groups[0]["names"] = names
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.get_eigens(fea_in)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    opt.get_transforms(offset=0.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ranks = {n: int(opt.eigens[n]["eigen_value"].shape[0] - round(float(torch.trace(opt.transforms[n]) ** 2 / (opt.transforms[n] ** 2).sum())))
             for n in names[:3]}
    out["eigens_and_transforms_r50"] = {
        "get_eigens_ms": (t1 - t0) * 1e3, "get_transforms_ms": (t2 - t1) * 1e3, "layers": len(layers),
        "eigensolver": "torch.linalg.eigh (rocSOLVER syevd, a library call), equal-width layers batched 16 per call",
        "projector_kernel": "nsgp_projector_head_kernel (P = I - U U^T from the orthonormalised removed directions; HIP, fp32 MFMA)",
        "note": "the reference runs torch.svd on every rank, twice (runner:554-555); under DDP the product shards the layers over the ranks (runner/dist.py)",
        "example_ranks_removed": ranks}
    opt.close()
    del opt, fea_in, params
    torch.cuda.empty_cache()
    # prototype-bank builds: VOC 15+5 sized, COCO 40+40 average class, COCO 'person'-sized stress class
    out["prototype_bank"] = [_bank_build(dev, 15, 300, 11), _bank_build(dev, 4, 1500, 12), _bank_build(dev, 1, 20000, 13)]
    torch.cuda.empty_cache()
    # R-101-FPN (configs[4]) projected step: 101 layers, 175.9 GFLOP
    layers = O.resnet_fpn_projected_layers(101)
    gen = torch.Generator(device=dev).manual_seed(77)
    params, names, cache = [], [], {}
    for n, cout, D in layers:
        k = 3 if ("conv2" in n or "fpn_convs" in n) else 1
        params.append(torch.nn.Parameter(torch.randn(cout, D // (k * k), k, k, device=dev, generator=gen) * 0.02))
        names.append(n)
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    opt.param_groups[0]["names"] = names
    for (n, cout, D) in layers:
        if D not in cache:
            cache[D] = make_basis(D, dev, 2000 + D)
        opt.set_basis(n, cache[D][0], cache[D][1])
    grads = [torch.randn(p.shape, device=dev, generator=gen) * 1e-3 for p in params]
    for p, g_ in zip(params, grads):
        p.grad = g_
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    opt.profile_begin(10)
    for _ in range(10):
        opt.step()
    torch.cuda.synchronize()
    _, u_ms, g_ms = opt.profile_end()
    flops = opt.plan_stats()[0]
    out["r101_projected_step"] = {"layers": len(layers), "elementwise_kernel_ms": u_ms, "projection_launches_ms": g_ms, "nsgp_step_ms": u_ms + g_ms,
                                  "dense_form_flops": flops, "dense_equivalent_tflops": flops / (g_ms * 1e-3) / 1e12,
                                  "layers_on_low_rank_form": opt.lowrank_stats()[0], "path": "low_rank (default)" if opt.lowrank_stats()[0] else (opt.uses_split_mfma() or "f32")}
    opt.close()
    return out


REPRE_CONFIGS = {"v15": dict(K=150, split=[0, 15, 20]), "v10": dict(K=100, split=[0, 10, 20]), "c40": dict(K=400, split=[0, 40, 80])}


def repre_step(N, dev, K, split, reps=30):
    """SURVEY 8(d) metric (ii): the per-step RePRE replay pass alone -- the K-row prototype bank through Shared2FCBBoxHeadTask
    (12544 -> 1024 -> 1024 -> per-task class heads) and CE(softmax(.)), forward + backward into the head's weight gradients
    (standard_roi_replay_head.py:468-501, convfc_bbox_head_task.py:235-276) -- HIP events around back-to-back passes on the current
    stream, median of `reps`.  Three implementations of the same arithmetic: the fused HIP path (default; fp32 MFMA), the
    module-by-module torch path in fp32 (what round 2 shipped) and that path under bf16 autocast (what the reference's AMP run does).
    `roofline`: ALGORITHMIC work of one pass = forward + weight gradients only (the bank is a constant: no dX) -- bytes W1 read +
    dW1 written + the bank read twice + W2 / dW2, FLOPs 2 * 2 K (12544 * 1024 + 1024^2 + C * 1024) -- against both bounds."""
    torch.manual_seed(7)
    head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=split[-1],
                                             task_split=list(split), task_id=2).to(dev)

    class Replay(N.roi_heads.PrototypeReplay):
        pass
    rp = Replay()
    rp.bbox_head, rp.task_split, rp.task_id, rp.replay = head, list(split), 2, True
    rp.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev))
    rp.tmp_label = torch.randint(0, split[1], (K,), device=dev)
    C_ = split[2] + 1
    fin, hid = 12544, 1024

    head_params = list(head.parameters())

    def one(fused, amp, backward=True):
        rp.fused_replay = fused
        for p_ in head_params:          # what zero_grad(set_to_none=True) does, without walking the module tree inside the timed region
            p_.grad = None
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = rp.replay_loss(rp.bbox_featss)["replay_loss"]["replay_loss_cls"]
        else:
            loss = rp.add_replay_loss({})["replay_loss_cls"]
        if backward:
            loss.backward()
        return loss

    def timed(fused, amp, backward=True):
        for _ in range(3):
            one(fused, amp, backward)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            one(fused, amp, backward)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]
    def steady(fused, amp, n=200):
        """n passes back to back (no synchronisation in between): HIP events around the run and the host's own issue time.  (Short bursts
        measure the clock ramp: 20 passes after 5 warm-ups read 0.33 ms on a box where 200 after 20 read 0.20.)"""
        for _ in range(20 if n >= 100 else 5):
            one(fused, amp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            one(fused, amp)
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, host
    # `repre_step_ms` = the pass in steady state (passes back to back, as inside a training loop that keeps the GPU busy); the isolated
    # figure (a synchronisation before every pass: launch latency and a cold clock on top) is reported beside it
    stream_ms, host_ms = steady(True, False)
    iso_ms, fused_fwd_ms = timed(True, False), timed(True, False, backward=False)

    def graph_replay_ms(n=200):
        """The same pass captured once in a HIP graph (torch.cuda.graph: the 11 launches of the C calls are issued on the capturing
        stream) and replayed n times: no Python between the launches, i.e. the GPU time of the pass whatever the host's speed."""
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    one(True, False)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                one(True, False)
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            del graph
            return ms
        except Exception as exc:      # a capture the runtime refuses is a missing number, not a failed bench
            torch.cuda.synchronize()
            return f"not captured: {type(exc).__name__}: {str(exc)[:120]}"
    graph_ms = graph_replay_ms()
    module_ms, _ = steady(False, False)
    bf16_ms, _ = steady(False, True, n=20)
    flops = 2.0 * 2.0 * K * (fin * hid + hid * hid + C_ * hid)
    nbytes = 4.0 * (2 * hid * fin + 2 * K * fin + 2 * hid * hid + 2 * C_ * hid)
    rp.fused_replay = True
    return {"K": K, "task_split": list(split), "kept_class_columns": C_, "repre_step_ms": stream_ms, "host_issue_ms": host_ms,
            "graph_replay_ms": graph_ms, "isolated_pass_ms": iso_ms, "isolated_forward_only_ms": fused_fwd_ms, "module_path_fp32_ms": module_ms, "module_path_bf16_autocast_ms": bf16_ms,
            "timing": "HIP events around 200 passes back to back after 20 warm-ups (head.zero_grad + forward + backward each): the pass is host-bound, so "
                      "repre_step_ms follows the box's host speed (host_issue_ms beside it); graph_replay_ms = the same pass captured in a HIP graph and "
                      "replayed 200 times = its GPU time; isolated = median of 30 passes with a synchronisation before each; module paths: the steady-state protocol",
            "launches": "forward: skinny split-K GEMM + slab reduce (x2), class scores + row CE terms, mean = 6; backward: CE, dZ2 (+ class-head gradients), "
                        "skinny GEMM, dZ1, grouped weight-gradient GEMM = 5 (csrc/replay_head.hip)",
            "roofline": {"algorithmic_flops": flops, "algorithmic_bytes": nbytes,
                         "mfma": {"bound": "mfma", "achieved": flops / (stream_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops / (stream_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS},
                         "hbm": {"bound": "hbm", "achieved": nbytes / (stream_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": nbytes / (stream_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                         "note": "over the whole pass (11 launches), back to back; fp32 MFMA bound 7.9 GFLOP / 157.3 TF = 50 us at K = 150, HBM bound 118 MB / 8 TB/s = 15 us; "
                                 "per-kernel durations: profiles/r03/repre_pass_kernels"}}


def make_basis(D, dev, seed):
    """(V, first) for one layer width, the way a run produces them: C = X^T X with X = [4D x D] ~ N(0,1) *
    diag(logspace(0,-3,D)) seeded, its eigenbasis by eigh on the GPU (descending), and the adaptive elbow index = the number
    of removed directions (21 .. 91 for the R-50 / R-101 widths).  ``set_basis(name, V, first)`` then builds the projector."""
    from nsgp_repre_amd.optim.threshold import elbow_index
    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.randn(4 * D, D, device=dev, generator=g) * torch.logspace(0, -3, D, device=dev)[None, :]
    lam, Q = torch.linalg.eigh((X.t() @ X).contiguous())
    sv = lam.abs()
    order = torch.argsort(sv, descending=True, stable=True)
    first = int(elbow_index(sv[order].cpu().numpy(), 0.0, "sgd"))
    return Q[:, order].contiguous(), max(1, first)


def host_cores():
    """Cores this process may actually use: min(affinity, cgroup cpu quota).  (On the GPU box
    os.cpu_count() says 256 while the container's share is 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    env = os.environ.get("NSGP_BENCH_CPU_THREADS")
    return int(env) if env else min(n, 64)


def cpu_baseline(table, seconds_budget=15.0):
    """The oracle's SGDNSCL step (the reference's arithmetic: a torch-CPU fp32 ``mm`` per projected
    layer inside a Python loop over every tensor) timed on this host's cores over the FULL
    162-tensor table, same shapes and hyper-parameters as the GPU run.  Reported, never the target."""
    import nsgp_oracle as O
    torch.set_num_threads(host_cores())
    g = torch.Generator().manual_seed(99)
    names = [n for n, _, _ in table]
    params = [torch.randn(s, generator=g) * 0.02 for _, s, _ in table]
    grads0 = [torch.randn(s, generator=g) * 1e-3 for _, s, _ in table]
    tr, cache = {}, {}
    for n, s, proj in table:
        if proj:
            D = s[1] * s[2] * s[3]
            if D not in cache:
                cache[D] = torch.randn(D, D, generator=g) / D ** 0.5
            tr[n] = cache[D]
    states = [dict() for _ in table]
    hp = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    times = []
    t_all = time.perf_counter()
    while True:
        grads = [x.clone() for x in grads0]
        t0 = time.perf_counter()
        O.sgd_nscl_step(names, params, grads, states, tr, **hp)
        times.append(time.perf_counter() - t0)
        if (time.perf_counter() - t_all > seconds_budget and len(times) >= 4) or len(times) >= 200:
            break
    timed = times[1:]   # first call = warm-up (thread pool, page faults)
    step_ms = sorted(timed)[len(timed) // 2] * 1e3
    # ---- the RePRE replay pass on the same cores: the oracle's task head + double-softmax CE, forward + backward (torch-CPU autograd),
    # K = 150 prototypes through 12544 -> 1024 -> 1024 -> (15 | 5 | 1) -- standard_roi_replay_head.py:468-501
    K, split = 150, [0, 15, 20]

    def lin(o, i):
        return (torch.randn(o, i, generator=g) * (1.0 / i ** 0.5)).requires_grad_(True), torch.zeros(o, requires_grad=True)
    shared = [lin(1024, 12544), lin(1024, 1024)]
    fc_cls = [lin(15, 1024), lin(5, 1024), lin(1, 1024)]
    fc_reg = [lin(60, 1024), lin(20, 1024)]
    bank = torch.relu(torch.randn(K, 12544, generator=g))
    labels = torch.randint(0, 15, (K,), generator=g)
    leaves = [t for pair in shared + fc_cls for t in pair]
    rtimes = []
    t_all = time.perf_counter()
    while True:
        for t in leaves:
            t.grad = None
        t0 = time.perf_counter()
        cls, _ = O.task_head_forward(bank, shared, fc_cls, fc_reg, 2, len(split))
        O.replay_loss_from_scores(cls, labels, split[2]).backward()
        rtimes.append(time.perf_counter() - t0)
        if (time.perf_counter() - t_all > seconds_budget / 2 and len(rtimes) >= 4) or len(rtimes) >= 200:
            break
    repre_ms = sorted(rtimes[1:])[len(rtimes[1:]) // 2] * 1e3
    return dict(repre_step_ms=repre_ms, nsgp_plus_repre_step_ms=step_ms + repre_ms,
                repre_sample=f"oracle task_head_forward + replay_loss_from_scores, forward + backward, K = 150: 1 warm-up + median of {len(rtimes) - 1} passes",
                **_cpu_nsgp_fields(step_ms, timed, seconds_budget))


def _cpu_nsgp_fields(step_ms, timed, seconds_budget):
    return dict(value=1e3 / step_ms, unit="NSGP projected steps/s (= images/s of the hot path alone at 1 image per step; the detector is not run on the CPU side)",
                cores=torch.get_num_threads(), kind="port", step_ms=step_ms,
                sample=f"oracle SGDNSCL.step over the full R-50-FPN table (162 tensors, 50 projected, 118.3 GFLOP): "
                       f"1 warm-up + median of {len(timed)} steps in ~{seconds_budget:.0f} s; the RePRE replay pass is timed beside it "
                       "(repre_step_ms)")


PEAK_16BIT_MATRIX_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)


SUSTAINED_HBM_GBS_3R2W = 5840.0     # measured, profiles/r03/hbm_peak_study.log (the higher of the two sizes)


def gbs_of(ms, nbytes):
    return nbytes / (ms * 1e-3) / 1e9 if ms else 0.0


def _pmc_file():
    for name in ("r03/traffic.json", "r02/traffic.json", "r01_traffic.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            return f
    return None


def roofline_block(split, flops, abytes_kernel, gemm_ms, update_ms, n_prof, numel, ntiles, nproj, detail=None, lowrank=None,
                   proj_numel=0, proj_bytes=0, shape=None):
    """`roofline` of the NSGP step's dominant kernel, from HIP events the library records around each launch of the timed steps.

    DEFAULT path (every projected layer on the low-rank form; `lowrank` = the plan's low-rank stats, `detail` = per-launch ms):
    the step is four launches, all HBM-bound.  The longest is the fused update + T launch of the 50 projected layers
    (`nsgp_update_lr_kernel`: g r+w, buf r+w, p r = 20 B per projected element -- the gradient is written back because the
    reference mutates p.grad in place (`grad.add_(wd, p)`, SGD_NSCL.py:400) and the optimizer mirrors that by default -- with
    T = u U computed on the fp32 MFMA while the stream is in flight) -> `bound: hbm`, `achieved` = those algorithmic bytes / its
    average duration, `peak` 8 TB/s, `traffic` = the HBM bytes rocprofv3's FETCH_SIZE / WRITE_SIZE counters saw per launch.  The
    other launches are blocks beside it, each against the same HBM peak with ITS algorithmic bytes (multi-tensor update of the
    un-projected tensors: 24 B per element; apply: reads the update, reads + writes p = 12 B per projected element), and
    `dense_equivalent_tflops` says what rate a dense u @ P (sum 2 Cout D^2 = 118.3 GFLOP) would have needed to
    finish in the time the low-rank projection takes (fused launch's share not counted: reduce + apply).

    DENSE path (projectors assigned from outside, or low_rank = False): the grouped projection GEMM dominates -> `bound: mfma`,
    `achieved` = ALGORITHMIC FLOP/s = flops / the launch's average duration, `peak` = the dense peak of the matrix unit the kernel
    runs on; on the split paths every fp32 product is evaluated as three fp16 (six bf16) MFMA products, reported as
    `executed_mfma_utilisation`, never as `frac`."""
    f = _pmc_file()
    tr = json.load(open(f)) if f else {}

    def hbm(kernel, ms, nbytes, note):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms else 0.0
        return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "kernel_ms": ms, "algorithmic_bytes": nbytes, "algorithmic_bytes_note": note}
    if lowrank and lowrank[0] == nproj and detail:
        plain_ms, fused_ms, dense_ms, reduce_ms, apply_ms = detail
        plain_numel = numel - proj_numel
        merged = bool(shape) and shape[2] > 0 and shape[0] == 0      # the un-projected tensors' chunks ride in the fused launch's grid
        fused_bytes = 20 * proj_numel + (24 * plain_numel if merged else 0)
        out = hbm("nsgp_update_lr_kernel<SGD> (the projected layers' elementwise update fused with T = u U on the exact fp32 MFMA"
                  + ("; the un-projected tensors' update rides at the end of its grid)" if merged else ")"),
                  fused_ms, fused_bytes, "g r+w (the reference's in-place grad.add_(wd, p), mirrored), momentum buffer r+w, p r = 20 B per element "
                                         "of the 50 projected layers (26.6 M elements)" + (" + g r+w, buf r+w, p r+w = 24 B per element of the un-projected "
                                                                                          "tensors (14.6 M elements)" if merged else ""))
        out.update({"profiled_steps": n_prof, "layers": nproj, "traffic": tr.get("nsgp_update_lr_kernel_hbm_bytes_per_launch"),
                    "frac_of_sustained_hbm": gbs_of(fused_ms, fused_bytes) / SUSTAINED_HBM_GBS_3R2W,
                    "sustained_hbm_note": f"a plain linear stream of this launch's access mix (3 reads + 2 writes, 16-byte non-temporal accesses) sustains "
                                          f"{SUSTAINED_HBM_GBS_3R2W / 1e3:.2f} TB/s on an MI355X box (5.51-5.84 across sizes; read-only 7.1-7.25, copy 6.0: "
                                          "tools/hbm_peak_bench.hip, profiles/r03/hbm_peak_study.log); `frac` stays priced against the 8 TB/s data-sheet figure",
                    "traffic_source": tr.get("source"),
                    "mfma_flops": lowrank[1] / 2, "mfma_frac_of_fp32_matrix_peak": lowrank[1] / 2 / (fused_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                    "nsgp_step_launches": ("" if merged else "nsgp_update_kernel -> ") + "nsgp_update_lr_kernel -> nsgp_lr_reduce_kernel -> nsgp_lr_apply_kernel",
                    "launch_shape_workgroups": dict(zip(("update", "fused_lowrank_units", "fused_plain_chunks", "reduce", "apply"), shape)) if shape else None,
                    "elementwise": None if merged else hbm("nsgp_update_kernel<SGD> (multi-tensor update of the un-projected tensors)", plain_ms, 24 * plain_numel,
                                                           "g r+w, buf r+w, p r+w = 24 B per element of the un-projected tensors (14.6 M elements)"),
                    "lowrank_reduce_ms": reduce_ms,
                    "lowrank_apply": hbm("nsgp_lr_apply_kernel<SGD> (p += c (u - T U^T), exact fp32 MFMA, K = r)", apply_ms, 12 * proj_numel,
                                         "reads the update, reads and writes p = 12 B per projected element"),
                    "step_bytes_moved_by_the_launches": 20 * proj_numel + 24 * plain_numel + 12 * proj_numel,
                    "step_bytes_moved_note": "what the four launches move: the apply launch re-reads the update and p of the projected layers (12 B / element) because T = u U "
                                             "must be complete first -- 32 B per projected element where SURVEY 8(d)'s algorithmic count is 24",
                    "step_launch_gbs": (20 * proj_numel + 24 * plain_numel + 12 * proj_numel) / ((plain_ms + fused_ms + reduce_ms + apply_ms) * 1e-3) / 1e9,
                    "step_8d_bytes": 24 * numel,
                    "step_frac_vs_8d_bytes": 24 * numel / ((plain_ms + fused_ms + reduce_ms + apply_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    "lowrank_apply_traffic": tr.get("nsgp_lr_apply_kernel_hbm_bytes_per_launch"),
                    "mfma_busy_fraction_pmc": tr.get("nsgp_update_lr_kernel_mfma_busy_fraction"),
                    "lowrank_flops": lowrank[1], "dense_form_flops": flops,
                    "dense_equivalent_tflops": flops / ((reduce_ms + apply_ms) * 1e-3) / 1e12,
                    "dense_equivalent_note": "the rate a dense u @ P over the same layers (SURVEY 8d's 118.3 GFLOP) would need to match the projection's own "
                                             "launches (reduce + apply; T rides in the update launch); the dense fp16-split kernel itself is timed under "
                                             "hot_path.mfma_paths / hot_path.roofline_dense_f16x2"})
        return out
    alg_tf = flops / (gemm_ms * 1e-3) / 1e12
    mult = {"f16x2": 3}.get(split, 1)
    peak = PEAK_16BIT_MATRIX_TFLOPS if split else PEAK_FP32_MATRIX_TFLOPS
    kernel = {"f16x2": "nsgp_project_v2_kernel<SGD> (256x128 tiles, LDS-DMA, two-term fp16 split with per-row / per-column scales)",
              }.get(split, "nsgp_project_kernel<SGD,fast> (fp32 MFMA)")
    out = {"bound": "mfma", "kernel": kernel, "achieved": alg_tf, "peak": peak, "unit": "TFLOP/s", "frac": alg_tf / peak,
           "executed_flops_per_algorithmic_flop": mult, "executed_mfma_utilisation": mult * alg_tf / peak,
           "vs_fp32_matrix_peak": alg_tf / PEAK_FP32_MATRIX_TFLOPS,
           "kernel_ms": gemm_ms, "profiled_steps": n_prof, "algorithmic_flops": flops, "tiles": ntiles, "layers": nproj,
           "algorithmic_bytes": abytes_kernel, "hbm_frac_of_peak": abytes_kernel / (gemm_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
           "algorithmic_bytes_note": "this launch: projectors (4 B/element, read once) + the update's split copy (4 B/element) + p read and written",
           "elementwise_kernel_ms": update_ms, "elementwise_kernel_hbm_gbs": (5 * 4 * numel) / (update_ms * 1e-3) / 1e9,
           "elementwise_note": "nsgp_update_kernel: g r, buf r+w, p r+w = 20 B/element algorithmic (+4 B/element of split copy for projected tensors on the fp16 path)",
           "traffic": None}
    if tr and (split == "f16x2" or "kernel" not in tr):   # HBM bytes per launch + MFMA-busy from the rocprofv3 --pmc passes (tools/profile.sh)
        out["traffic"] = tr.get("nsgp_project_kernel_hbm_bytes_per_launch")
        out["traffic_source"] = tr.get("source")
        if "mfma_busy_fraction" in tr:
            out["mfma_busy_fraction_pmc"] = tr["mfma_busy_fraction"]
    return out


def end_to_end_training(N, dev, world, local_rank, basis_cache, steps, warmup, amp, batch_size=1, split=(0, 15, 20), K=150):
    """SURVEY 8(d) "end-to-end img/s": the whole task-2 training step of cl_faster_rcnn_nsgp_repre_15_5_2.py on synthetic
    800x1344 batches -- teacher predict + pseudo-label filter, student forward (RPN + RoI losses + replay loss on the
    K=150 bank), backward (DDP bucketed RCCL all-reduce overlapped with it when world > 1) and the projected SGDNSCL
    step.  The detector is nsgp_repre_amd.detection (stock recipe in plain PyTorch-ROCm; mmdet is not in the image)."""
    import copy
    import tempfile
    import torch.distributed as dist
    from nsgp_repre_amd.detection import build_faster_rcnn, relocate_segment_final_weights, synthetic_batch
    torch.manual_seed(4321)
    split = list(split)
    model = build_faster_rcnn(depth=50, num_classes=split[-1], task_id=2, task_split=split).to(dev)
    relocate_segment_final_weights(model)      # guard against a stock MIOpen over-read (profiles/README.md, incident analysis)
    head = model.roi_head
    mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
    mix.task_id = 2
    # EWC on the BatchNorm parameters (runner:558-565, 946-1073): the importance of "task 1" from a seeded pass over two synthetic
    # old-class batches (calculate_save_importance, before the teacher exists), loaded back and hooked onto model.loss like the
    # reference's task-2 start does -- so the timed step carries `ewc_loss` (and its backward)
    tmp = tempfile.mkdtemp(prefix="nsgp_bench_")
    mix._task_work_dir, mix.previous_dir, mix.reg_params, mix.ewc_reg_terms = tmp, tmp, {}, {}
    old_batches = [synthetic_batch(1, (split[0], split[1]), dev, seed=900 + 10 * local_rank + i) for i in range(2)]

    def importance_loss(m, b):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            losses = m(b[0], copy.deepcopy(b[1]), mode="loss")
        return sum(v for k, v in losses.items() if "loss" in k)
    model.train()
    mix.calculate_save_importance(model, old_batches, importance_loss)
    model.train()
    head.replay = True
    head.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev))
    head.tmp_label = torch.randint(0, split[1], (K,), device=dev)
    mix.attach_teacher(model)                                  # runner:527-547
    mix.load_importance(model)                                 # runner:558
    mix.wrap_loss_with_ewc(model)                              # runner:559-565
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    opt = N.SGDNSCL(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
    N.runner.nullspace.wire_param_names(opt, model)             # runner:473-484
    ignore = N.runner.nullspace.full_ignore_keys(["rpn", "roi_head"])
    n_proj = 0
    for n, p in model.named_parameters():
        if p.requires_grad and p.dim() == 4 and not N.runner.nullspace.should_ignore(n, ignore):
            D = p[0].numel()
            if D not in basis_cache:
                basis_cache[D] = make_basis(D, dev, 2000 + D)
            opt.set_basis(n, basis_cache[D][0], basis_cache[D][1])
            n_proj += 1
    model.train()
    batches = [synthetic_batch(batch_size, (split[1], split[2]), dev, seed=100 * local_rank + i) for i in range(4)]
    net = model
    if world > 1:
        # find_unused_parameters=True is the reference's own setting (_base_/brnsrunetime.py:27); the frozen heads of future tasks
        # and the teacher never receive gradients
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                        gradient_as_bucket_view=os.environ.get("NSGP_BENCH_BUCKET_VIEW", "1") == "1",
                                                        find_unused_parameters=os.environ.get("NSGP_BENCH_FIND_UNUSED", "1") == "1")      # (rehearsal knobs; the driver sets neither)
    fwd_bwd, opt_ms = [], []

    verbose = bool(os.environ.get("NSGP_BENCH_DUMP_AFTER"))

    def one_step(i):
        if verbose:
            print(f"[rank {os.environ.get('RANK', 0)}] step {i} begins", file=sys.stderr, flush=True)
        x, samples = batches[i % len(batches)]
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            losses = net(x, copy.deepcopy(samples), mode="loss")
        loss = sum(v for k, v in losses.items() if "loss" in k)
        loss.backward()
        e1.record()
        opt.step()
        opt.zero_grad()
        e2.record()
        return losses, (e0, e1, e2)

    for i in range(warmup):
        losses, _ = one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    opt.profile_begin(steps)
    t0 = time.perf_counter()
    evs = []
    for i in range(steps):
        losses, ev = one_step(i)
        evs.append(ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    n_prof, update_ms, gemm_ms = opt.profile_end()
    detail, lowrank = opt.profile_detail(), opt.lowrank_stats()
    flops, _abytes, ntiles, nproj = opt.plan_stats()
    split = opt.uses_split_mfma()
    proj_numel = sum(p.numel() for g_ in opt.param_groups for n_, p in zip(g_["names"], g_["params"]) if n_ in opt.transforms)
    proj_bytes = sum(P.numel() * 4 for P in opt.transforms.values())
    all_numel = sum(p.numel() for g_ in opt.param_groups for p in g_["params"])
    for e0, e1, e2 in evs:
        fwd_bwd.append(e0.elapsed_time(e1))
        opt_ms.append(e1.elapsed_time(e2))
    finite = all(bool(torch.isfinite(v)) for v in losses.values())
    out = {"img_s": world * batch_size * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
           "batch_per_gpu": batch_size,
           "image": "3x800x1344 (1333x800 padded to /32)", "n_gpus": world,
           "teacher_student_fwd_bwd_ms": sum(fwd_bwd) / len(fwd_bwd), "optimizer_step_ms": sum(opt_ms) / len(opt_ms),
           "nsgp_kernels_ms": update_ms + gemm_ms, "projected_layers": n_proj,
           "_roofline": dict(split=split, flops=flops, abytes_kernel=proj_bytes + 3 * 4 * proj_numel, gemm_ms=gemm_ms, update_ms=update_ms,
                             n_prof=n_prof, numel=all_numel, ntiles=ntiles, nproj=nproj, detail=detail, lowrank=lowrank,
                             proj_numel=proj_numel, proj_bytes=proj_bytes, shape=opt.launch_shape()),
           "nsgp_launch_ms": dict(zip(("update", "update_lr_fused_t", "dense_gemm", "lowrank_reduce", "lowrank_apply"), detail)),
           "layers_on_low_rank_form": lowrank[0], "prototype_bank_rows": K, "task_split": split,
           "host_cores_per_rank": host_cores(),
           "trainable_tensors": sum(len(g["params"]) for g in opt.param_groups),
           "losses_finite": finite, "loss_keys": sorted(losses.keys()),
           "detector_dtype": "bf16 autocast (replay-bank pass, losses, NSGP step fp32)" if amp else "f32",
           "parallelism": f"DDP x{world}: bucketed RCCL all-reduce of the gradients overlapped with backward" if world > 1 else "single",
           "detector": "nsgp_repre_amd.detection (R-50-FPN Faster R-CNN, stock recipe in plain PyTorch-ROCm: MIOpen convolutions, "
                       "hipBLASLt GEMMs; teacher predict + pseudo-label filter every step, as det:65-109)"}
    opt.close()
    del net, model, opt
    torch.cuda.empty_cache()
    return out


def hot_path_only(N, dev, args, cache):
    """The fork's additions alone (no detector): SGDNSCL.step over the full 162-tensor table + the RePRE replay loss on the K=150
    bank, on synthetic gradients living in one flat bucket (what DDP's gradient_as_bucket_view gives).  Reported under `hot_path`."""
    table = r50_fpn_voc_parameter_table()
    gen = torch.Generator(device=dev).manual_seed(1234)
    params, names = [], []
    for n, shape, _ in table:
        params.append(torch.nn.Parameter(torch.randn(shape, device=dev, generator=gen) * 0.02))
        names.append(n)
    bbox_head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=20,
                                                  task_split=[0, 15, 20], task_id=2).to(dev)
    for n, p in bbox_head.named_parameters():
        params.append(p)
        names.append("roi_head.bbox_head." + n)
        table.append(("roi_head.bbox_head." + n, tuple(p.shape), False))
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)  # schedule_1x_sgdnscl.py:21
    opt.param_groups[0]["names"] = names
    for n, shape, proj in table:
        if proj:
            D = shape[1] * shape[2] * shape[3]
            if D not in cache:
                cache[D] = make_basis(D, dev, 2000 + D)
            opt.set_basis(n, cache[D][0], cache[D][1])      # one [D x D] projector per layer, as in the reference
    K = 150  # <= 10 prototypes x 15 old classes (VOC 15+5)

    class Replay(N.roi_heads.PrototypeReplay):   # the product's replay_loss (head:468-501) on a synthetic bank
        pass
    replay = Replay()
    replay.bbox_head, replay.task_split, replay.task_id, replay.replay = bbox_head, [0, 15, 20], 2, True
    replay.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev, generator=gen))
    replay.tmp_label = torch.randint(0, 15, (K,), device=dev, generator=gen)
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    flat_numel_real = sum(p.numel() for p in params)
    flat_grads = torch.zeros(total, device=dev)
    synth_flat = torch.randn(total, device=dev, generator=gen) * 1e-3
    for p, o in zip(params, offs):
        p.grad = flat_grads[o:o + p.numel()].view_as(p)
    host_step = []

    def one_step():
        flat_grads.copy_(synth_flat)           # the detector's backward() writes the grads (synthetic)
        loss = replay.add_replay_loss({})["replay_loss_cls"]   # RePRE replay pass, forward: the fused HIP path (csrc/replay_head.hip)
        loss.backward()                        # + backward: accumulates into the head's grad views
        h0 = time.perf_counter()
        opt.step()                             # NSGP projected step: 4 HIP launches on the default path
        host_step.append(time.perf_counter() - h0)

    def timed(steps):
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        opt.profile_begin(steps)
        t1 = time.perf_counter()
        for _ in range(steps):
            one_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / steps * 1e3
        _, u_ms, g_ms = opt.profile_end()
        return ms, u_ms, g_ms

    flops = None
    out = {"workload": "SGDNSCL.step over 50 projected layers + 112 plain tensors (41.2M params) + replay loss fwd/bwd on K=150 prototypes; synthetic gradients; "
                       "projectors from SURVEY 8d's seeded covariances (eigh -> elbow -> set_basis)",
           "mfma_paths": {}}
    proj_numel = sum(int(torch.tensor(shape).prod()) for _, shape, proj in table if proj)
    # ---- the DEFAULT path: head-form projectors applied in the low-rank form (north_star: g - U (U^T g))
    assert opt.low_rank is True
    ms, u_ms, g_ms = timed(args.steps)
    out["host_ms_in_optimizer_step"] = 1e3 * sum(host_step[-args.steps:]) / max(1, len(host_step[-args.steps:]))
    n_lr, lr_flops, lt1, lt2 = opt.lowrank_stats()
    plain_ms, fused_ms, dense_ms, t_ms, a_ms = opt.profile_detail()
    flops = opt.plan_stats()[0]
    out["low_rank_form"] = {
        "default": True, "ms_per_step": ms, "nsgp_step_ms": u_ms + g_ms, "update_kernel_ms": plain_ms, "update_lr_fused_t_kernel_ms": fused_ms,
        "lowrank_reduce_ms": t_ms, "lowrank_apply_ms": a_ms, "projection_launches_ms": g_ms, "layers": n_lr, "removed_directions_per_width": {str(D): int(v[1]) for D, v in sorted(cache.items())},
        "lowrank_flops": lr_flops, "dense_form_flops": flops, "dense_equivalent_tflops": flops / (g_ms * 1e-3) / 1e12,
        "launch_shape_workgroups": dict(zip(("update", "fused_lowrank_units", "fused_plain_chunks", "reduce", "apply"), opt.launch_shape())),
        "update_hbm_gbs": (24 * (flat_numel_real - proj_numel) / (plain_ms * 1e-3) / 1e9 if plain_ms else None) if opt.launch_shape()[0] else None,
        "update_lr_fused_t_hbm_gbs": (20 * proj_numel + (24 * (flat_numel_real - proj_numel) if opt.launch_shape()[2] else 0)) / (fused_ms * 1e-3) / 1e9 if fused_ms else None,
        "bytes_note": "update: g r+w, buf r+w, p r+w = 24 B/element; fused update + T: g r+w, buf r+w, p r = 20 B/element; apply: u r, p r+w = 12 B/element",
        "lowrank_apply_hbm_gbs": 12 * proj_numel / (a_ms * 1e-3) / 1e9 if a_ms else None,
        "workgroups": {"update_lr_fused_t": lt1, "lowrank_apply": lt2},
        "note": "parity: tests/test_gpu_parity.py::test_full_table_low_rank_default_vs_oracle_per_row, ::test_low_rank_form_matches_dense_form, ::test_g1b_default_pipeline_from_covariance_per_row"}
    # the same step without mirroring the reference's in-place mutation of p.grad (`grad.add_(wd, p)`, SGD_NSCL.py:400): zero_grad()
    # discards it right after the step; `optimizer.mutate_grad = False` saves the write-back (4 B per element)
    if not args.hot_path_only:      # (not under the profiler: the per-kernel counter means of tools/profile.sh describe the default setting only)
        opt.mutate_grad = False
        ms2, u2, g2 = timed(args.steps)
        out["low_rank_form"]["without_grad_mirror"] = {"nsgp_step_ms": u2 + g2, "launch_ms": dict(zip(("update", "update_lr_fused_t", "dense_gemm", "lowrank_reduce",
                                                                                                       "lowrank_apply"), opt.profile_detail())),
                                                       "note": "optimizer.mutate_grad = False: p.grad is left as backward() wrote it"}
        opt.mutate_grad = True
    # MIXED steps: the three 4608-wide layers with (a) 160 removed directions -- the wide rank class (129 .. 256: its own fused + apply launches
    # beside the common ones, all 50 layers on the low-rank form) -- and (b) 300 -- beyond the low-rank form: those three fall back to the dense
    # fp16-split GEMM (logged by set_basis), 47 stay on the low-rank launches
    if not args.hot_path_only:
        wide = [n for n, shape, proj in table if proj and shape[1] * shape[2] * shape[3] == 4608]
        for r_wide, key in ((160, "mixed_step_3_layers_at_160_directions_wide_rank_class"), (300, "mixed_step_3_layers_at_300_directions_dense_fallback")):
            for n in wide:
                opt.set_basis(n, cache[4608][0], r_wide)
            _ms3, u3, g3 = timed(args.steps)
            out["low_rank_form"][key] = {
                "nsgp_step_ms": u3 + g3, "layers_on_low_rank": f"{opt.lowrank_stats()[0]}/{len([1 for _, _, pr in table if pr])}", "dense_tiles_f16x2": opt.tile_counts()[2],
                "launch_ms": dict(zip(("update", "update_lr_fused_t", "dense_gemm", "lowrank_reduce", "lowrank_apply"), opt.profile_detail()))}
        for n in wide:
            opt.set_basis(n, cache[4608][0], cache[4608][1])
    # ---- the dense GEMM on every MFMA path, with the SAME projectors (what externally assigned projectors run on)
    opt.low_rank = False
    for path in ("f16x2", False):
        opt.split_mfma = path
        ms, u_ms, g_ms = timed(args.steps)
        tf = flops / (g_ms * 1e-3) / 1e12
        if path == "f16x2":
            v2_tiles = opt.tile_counts()[2]
        out["mfma_paths"][path or "f32"] = {
            "uses_split_mfma": opt.uses_split_mfma(), "ms_per_step": ms, "nsgp_step_ms": u_ms + g_ms, "elementwise_kernel_ms": u_ms,
            "projection_kernel_ms": g_ms, "fp32_equivalent_tflops": tf,
            "frac_of_unit_peak": tf / (PEAK_16BIT_MATRIX_TFLOPS if path else PEAK_FP32_MATRIX_TFLOPS),
            "frac_of_fp32_matrix_peak": tf / PEAK_FP32_MATRIX_TFLOPS}
    f32, f16 = out["mfma_paths"]["f32"], out["mfma_paths"]["f16x2"]
    out["roofline_fp32_mfma"] = {"bound": "mfma", "kernel": "nsgp_project_kernel<SGD,fast> (v_mfma_f32_32x32x2_f32, exact fp32)",
                                 "achieved": f32["fp32_equivalent_tflops"], "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                 "frac": f32["frac_of_fp32_matrix_peak"], "kernel_ms": f32["projection_kernel_ms"]}
    pbytes = sum(P.numel() * 4 for P in opt.transforms.values())
    out["roofline_dense_f16x2"] = roofline_block("f16x2", flops, pbytes + 3 * 4 * proj_numel, f16["projection_kernel_ms"], f16["elementwise_kernel_ms"],
                                                 args.steps, flat_numel_real, v2_tiles, n_lr)
    opt.split_mfma = "f16x2"
    opt.low_rank = True
    # the AdamW flavour of the same step (schedule_1x_adamwnscl.py:21) -- row a3 of SURVEY 8; parity: the G1 / G1b adamw goldens
    adamw = N.AdamWNSCL(params, lr=1e-4, weight_decay=0.1, svd=True)
    adamw.param_groups[0]["names"] = names
    for n, shape, proj in table:
        if proj:
            D = shape[1] * shape[2] * shape[3]
            adamw.set_basis(n, cache[D][0], cache[D][1])
    flat_grads.copy_(synth_flat)
    for _ in range(3):
        adamw.step()
    torch.cuda.synchronize()
    adamw.profile_begin(args.steps)
    for _ in range(args.steps):
        adamw.step()
    torch.cuda.synchronize()
    _, aw_u, aw_g = adamw.profile_end()
    out["adamw_nscl"] = {"nsgp_step_ms": aw_u + aw_g, "elementwise_kernel_ms": aw_u, "projection_launches_ms": aw_g,
                         "layers_on_low_rank_form": adamw.lowrank_stats()[0],
                         "elementwise_hbm_gbs": 8 * 4 * flat_numel_real / (aw_u * 1e-3) / 1e9,
                         "note": "AdamWNSCL.step, same table, default (low-rank) path; elementwise bytes: g r, m r+w, v r+w, p r (+ u w for projected tensors)"}
    adamw.close()
    opt.close()
    return out, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip hot_path / once_per_task / cpu_baseline")
    ap.add_argument("--f32-detector", action="store_true", help="run the detector in fp32 instead of bf16 autocast (the NSGP step is fp32 either way)")
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--config", choices=("v15", "v10"), default="v15",
                    help="v15 (default) = BASELINE configs[1]: VOC 15+5 task 2, 1 image per GPU, K = 150; v10 = configs[2]'s per-GPU step: VOC 10+10 "
                         "task 2 (split [0,10,20]), 16 images per GPU (voc_10_10_task2_2007.py:40), K = 100")
    ap.add_argument("--hot-path-only", action="store_true", help="profiling aid (tools/profile.sh): only the `hot_path` block -- the projected "
                    "step on every MFMA path + replay loss on synthetic gradients -- no detector, no once-per-task units")
    args = ap.parse_args()

    if os.environ.get("NSGP_BENCH_DUMP_AFTER"):      # rehearsal aid: every thread's Python stack to stderr after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NSGP_BENCH_DUMP_AFTER"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    # rehearsal knobs (single-GPU box): NSGP_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # NSGP_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); never set by the driver
    if os.environ.get("NSGP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("NSGP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import nsgp_repre_amd as N

    cache = {}
    wl = {"v15": dict(split=(0, 15, 20), K=150, batch=1, name="R-50-FPN VOC 15+5 task 2 (configs[1])"),
          "v10": dict(split=(0, 10, 20), K=100, batch=16, name="R-50-FPN VOC 10+10 task 2 (configs[2], per-GPU step)")}[args.config]
    if args.batch_per_gpu is None:
        args.batch_per_gpu = wl["batch"]
    if args.hot_path_only:
        hp, _ = hot_path_only(N, dev, args, cache)
        print(json.dumps({"hot_path": hp, "repre": {"v15": repre_step(N, dev, **REPRE_CONFIGS["v15"])}}))
        return
    # ---- the headline: K full training iterations, DDP gradient all-reduce inside for N > 1
    e2e = end_to_end_training(N, dev, world, local_rank, cache, args.steps, args.warmup, not args.f32_detector, batch_size=args.batch_per_gpu,
                              split=wl["split"], K=wl["K"])
    if rank == 0:
        rf = e2e.pop("_roofline")
        out = {
            "metric": "NSGP-RePRE training img/s (Faster R-CNN R-50-FPN, VOC 15+5 task 2: teacher + student fwd/bwd + replay loss + projected SGDNSCL step) + NSGP-projection step ms",
            "value": e2e["img_s"], "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": e2e["ms_per_step"],
            "nsgp_step_ms": rf["update_ms"] + rf["gemm_ms"],
            "repre_step_ms": None, "covariance_forward_ms": None,      # filled from hot_path.repre / once_per_task below (N = 1)
            # SURVEY 8(d)'s bytes for the whole step -- g r(+w), buf r+w, p r+w = 24 B per element of every tensor the step touches --
            # over the step's launches, against 8 TB/s (the per-launch fractions are in `roofline`)
            "step_frac_vs_8d_bytes": 24.0 * rf["numel"] / ((rf["update_ms"] + rf["gemm_ms"]) * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "layers_on_low_rank": f"{e2e['layers_on_low_rank_form']}/{e2e['projected_layers']}",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (NSGP step: parameters, gradients, state and every product -- the low-rank form runs on the exact fp32 MFMA)"
                     + ("" if args.f32_detector else "; detector forward/backward under bf16 autocast"),
            "data": "synthetic",
            "config": {"workload": f"{wl['name']}, {args.batch_per_gpu} synthetic 3x800x1344 image(s) per GPU per step, 50 projected layers, "
                                   f"K={wl['K']} prototype bank, EWC on the BatchNorm parameters",
                       "global_batch": world * args.batch_per_gpu, "parallelism": f"ddp{world}" if world > 1 else "single"},
            "roofline": roofline_block(**rf),
            "training_step": e2e,
        }
        if world == 1 and not args.no_extras:
            def guarded(name, fn):      # the headline above must reach the JSON line whatever happens in the wider sections
                try:
                    out[name] = fn()
                except Exception as exc:    # reported, never hidden
                    import traceback
                    traceback.print_exc()
                    out[name] = {"error": f"{type(exc).__name__}: {exc}"}
            table_box = {}

            def _hot():
                hp, table_box["table"] = hot_path_only(N, dev, args, cache)
                return hp
            guarded("hot_path", _hot)
            if not args.f32_detector:    # the same training step with an fp32 detector, beside the bf16 number
                def _f32():
                    r = end_to_end_training(N, dev, world, local_rank, cache, max(4, args.steps // 2), 3, False, batch_size=args.batch_per_gpu,
                                            split=wl["split"], K=wl["K"])
                    return {k: r[k] for k in ("img_s", "ms_per_step", "teacher_student_fwd_bwd_ms", "optimizer_step_ms", "losses_finite")}
                guarded("training_step_f32_detector", _f32)
            guarded("repre", lambda: {name: repre_step(N, dev, c["K"], c["split"]) for name, c in REPRE_CONFIGS.items()})
            if isinstance(out.get("repre"), dict) and args.config in out["repre"]:
                out["repre_step_ms"] = out["repre"][args.config].get("repre_step_ms")
            guarded("once_per_task", lambda: once_per_task_units(N, dev))
            try:
                out["covariance_forward_ms"] = out["once_per_task"]["covariance_forward_r50"]["ms"]
            except Exception:
                pass
            if not args.no_cpu_baseline:
                def _cpu():
                    cb = cpu_baseline(table_box.get("table") or r50_fpn_voc_parameter_table())
                    cb["gpu_nsgp_step_speedup"] = cb["step_ms"] / out["nsgp_step_ms"]
                    if out.get("repre_step_ms"):      # the north star's ratio: NSGP projection + RePRE step, GPU vs the CPU path on this host
                        cb["gpu_repre_step_speedup"] = cb["repre_step_ms"] / out["repre_step_ms"]
                        cb["gpu_nsgp_plus_repre_step_ms"] = out["nsgp_step_ms"] + out["repre_step_ms"]
                        cb["gpu_nsgp_plus_repre_speedup"] = cb["nsgp_plus_repre_step_ms"] / cb["gpu_nsgp_plus_repre_step_ms"]
                    return cb
                guarded("cpu_baseline", _cpu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
