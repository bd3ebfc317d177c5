"""bench.py -- NSGP-RePRE hot path on MI355X (see DESIGN.md section 'Measurement').

One "step" = one training iteration's worth of the fork's hot path for the workload
BASELINE.json configs[1] names (Faster R-CNN R-50-FPN, VOC 15+5 task 2, batch 1 image per GPU):
  (1) SGDNSCL.step over the full parameter table -- 50 projected conv layers
      (118.3 GFLOP of projection, 0.584 GB of projectors) + the un-projected tensors
      (BN, biases, RPN, RoI head: ~14.7 M elements) -- two HIP launches;
  (2) the RePRE replay loss on the K=150 prototype bank through a Shared2FCBBoxHeadTask-shaped
      head (12544->1024->1024->21), forward + backward (PyTorch-ROCm GEMMs).
The detector's own forward/backward (stock PyTorch-ROCm, SURVEY section 2.1 "out of scope")
is NOT inside the timed region and the metric name says so.

Multi-GPU: one process per GPU (torchrun), replicas of the same step exactly as DDP runs the
optimizer (identical grads after all-reduce); the per-step gradient all-reduce of the 41.5 M
fp32 parameters over RCCL IS inside the timed region for N>1.  value = images/s over all ranks.
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


def r50_fpn_hooked_convs(H=800, W=1344):
    """(name, cin, k, stride, pad, Hin, Win) of the 61 convs cal_fea_in hooks on R-50-FPN for one
    800x1344 padded image (ignore_keys drop rpn/roi_head): sum 2*L*D^2 = 1.86 TFLOP."""
    out = [("backbone.conv1", 3, 7, 2, 3, H, W)]
    h, w, inpl = H // 4, W // 4, 64
    for li, (nb, planes, stride) in enumerate(((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2)), start=1):
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


def once_per_task_units(dev):
    """The other two units of work of SURVEY 8d, timed on their own (never part of `value`):
    one hooked covariance forward (61 convs, 1.86 TFLOP of reference FLOPs) and the VOC-15+5-sized
    prototype-bank build (15 old classes x 300 RoIs x 12544)."""
    from nsgp_repre_amd import ops
    from nsgp_repre_amd.roi_heads.prototype_bank import build_prototype_bank
    g = torch.Generator(device=dev).manual_seed(5)
    layers = r50_fpn_hooked_convs()
    acts, covs, ws_bytes, ref_flops = {}, {}, 0, 0.0
    for n, cin, k, s, p, h, w in layers:
        if (cin, h, w) not in acts:
            acts[(cin, h, w)] = torch.randn(1, cin, h, w, device=dev, generator=g).abs()
        ws_bytes = max(ws_bytes, ops.cov_workspace_bytes(cin, h, w, (k, k), (s, s), (p, p)))
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        ref_flops += 2.0 * ho * wo * (cin * k * k) ** 2
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def forward():
        for n, cin, k, s, p, h, w in layers:
            covs[n] = ops.cov_accumulate_conv2d(acts[(cin, h, w)], (k, k), (s, s), (p, p), covs.get(n), ws)
    forward()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); forward(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    cov_ms = sorted(ts)[1]
    del acts, covs, ws
    # per class: 4 ReLU'd cluster centres + 0.6 * noise, ReLU'd (SURVEY 8d "Synthetic inputs -- RePRE")
    centres = torch.relu(torch.randn(15, 4, 12544, device=dev, generator=g))
    which = torch.randint(0, 4, (15, 300), device=dev, generator=g)
    feats = torch.relu(torch.gather(centres, 1, which[..., None].expand(-1, -1, 12544))
                       + 0.6 * torch.randn(15, 300, 12544, device=dev, generator=g)).reshape(15 * 300, 12544).contiguous()
    cls = torch.arange(15, device=dev).repeat_interleave(300)
    build_prototype_bank(feats, cls, [0, 15, 20], 2, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bank, labels, _, _ = build_prototype_bank(feats, cls, [0, 15, 20], 2, 10)
    torch.cuda.synchronize()
    bank_ms = (time.perf_counter() - t0) * 1e3
    return {"covariance_forward_ms": cov_ms, "covariance_reference_flops": ref_flops,
            "covariance_tflops_by_reference_flops": ref_flops / (cov_ms * 1e-3) / 1e12,
            "covariance_note": "61 hooked convs of R-50-FPN at 800x1344; only the upper triangle is computed (half the reference FLOPs), X never materialised",
            "prototype_bank_build_ms": bank_ms, "prototype_bank_rows": int(bank.shape[0]),
            "prototype_bank_note": "15 old classes x 300 RoIs x 12544 (VOC 15+5 sized), wall time incl. the host-side greedy cover"}


def make_basis(D, dev, seed):
    """A random orthonormal eigenbasis V [D x D]; the synthetic rank of the feature space is r = D//16,
    so the projector is V[:, r:] V[:, r:]^T (built by the HIP SYRK kernel through set_basis)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    V = torch.eye(D, device=dev)
    for _ in range(4):   # a product of 4 Householder reflections: dense, orthonormal to a few ulp, 8 launches
        u = torch.randn(D, 1, device=dev, generator=g)
        u = u / u.norm()
        V = V - 2.0 * (V @ u) @ u.t()
    return V.contiguous(), max(1, D // 16)


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
    return dict(value=1e3 / step_ms, unit="img/s", cores=torch.get_num_threads(), kind="port", step_ms=step_ms,
                sample=f"oracle SGDNSCL.step over the full R-50-FPN table (162 tensors, 50 projected, 118.3 GFLOP): "
                       f"1 warm-up + median of {len(timed)} steps in ~{seconds_budget:.0f} s; the replay loss is not "
                       "included on the CPU side (that favours the CPU number)")


PEAK_BF16_MATRIX_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def roofline_block(split, flops, abytes, gemm_ms, update_ms, nsgp_ms, n_prof, numel, ntiles, nproj):
    """`roofline` for the dominant kernel.  The algorithm is an fp32 contraction of 118.3 GFLOP.  On the split paths
    every fp32 product is evaluated as three fp16 (or six bf16) MFMA products (fp32-accurate), so the
    matrix cores EXECUTE 3 x (6 x) the algorithmic FLOPs: `achieved`/`peak`/`frac` are executed bf16 FLOP/s against the dense bf16 peak (a
    true utilisation, <= 1), and the fp32-equivalent rate against the fp32 matrix peak is given beside it."""
    alg_tf = flops / (gemm_ms * 1e-3) / 1e12
    common = {"kernel_ms": gemm_ms, "elementwise_kernel_ms": update_ms, "profiled_steps": n_prof, "algorithmic_flops": flops,
              "algorithmic_bytes": abytes, "step_hbm_frac_of_peak": abytes / (nsgp_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
              "elementwise_kernel_hbm_gbs": 5 * 4 * numel / (update_ms * 1e-3) / 1e9, "tiles": ntiles, "layers": nproj,
              "traffic": None, "algorithmic_fp32_equivalent_tflops": alg_tf,
              "vs_fp32_matrix_peak": alg_tf / PEAK_FP32_MATRIX_TFLOPS}
    if split:
        mult = {"bf16x3": 6, "f16x2": 3}[split]
        return {"bound": "mfma", "kernel": f"nsgp_project_kernel<SGD,fast,{split}>", "achieved": mult * alg_tf, "peak": PEAK_BF16_MATRIX_TFLOPS,
                "unit": "TFLOP/s", "frac": mult * alg_tf / PEAK_BF16_MATRIX_TFLOPS, "executed_flops_per_algorithmic_flop": mult,
                "note": ("three v_mfma_f32_32x32x16_f16 per fp32-equivalent product (2-term fp16 split of both operands, one power-of-two "
                         "scale per operand matrix, fp32 accumulation)" if split == "f16x2" else
                         "six v_mfma_f32_32x32x16_bf16 per fp32-equivalent product (3-term bf16 split of both operands, fp32 accumulation)")
                        + "; peak = dense 16-bit MFMA peak",
                **common}
    return {"bound": "mfma", "kernel": "nsgp_project_kernel<SGD,fast>", "achieved": alg_tf, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": alg_tf / PEAK_FP32_MATRIX_TFLOPS, "executed_flops_per_algorithmic_flop": 1, **common}


def end_to_end_training(N, dev, world, local_rank, basis_cache, steps, warmup, amp, batch_size=1, channels_last=False, graphs=False):
    """SURVEY 8(d) "end-to-end img/s": the whole task-2 training step of cl_faster_rcnn_nsgp_repre_15_5_2.py on synthetic
    800x1344 batches -- teacher predict + pseudo-label filter, student forward (RPN + RoI losses + replay loss on the
    K=150 bank), backward (DDP bucketed RCCL all-reduce overlapped with it when world > 1) and the projected SGDNSCL
    step.  The detector is nsgp_repre_amd.detection (stock recipe in plain PyTorch-ROCm; mmdet is not in the image)."""
    import copy
    import torch.distributed as dist
    from nsgp_repre_amd.detection import build_faster_rcnn, synthetic_batch
    torch.manual_seed(4321)
    model = build_faster_rcnn(depth=50, num_classes=20, task_id=2, task_split=[0, 15, 20]).to(dev)
    head = model.roi_head
    head.replay, K = True, 150
    head.bbox_featss = torch.relu(torch.randn(K, 12544, device=dev))
    head.tmp_label = torch.randint(0, 15, (K,), device=dev)
    mix = N.runner.br_nullspace_runner.NullSpaceTaskMixin()
    mix.task_id = 2
    mix.attach_teacher(model)                                  # runner:527-547
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
    # channels_last: only the ACTIVATIONS (the input image decides the layout of every convolution's output);
    # parameters stay contiguous -- the optimizer's [Cout x D] view of a conv weight is the reference's layout
    batches = [synthetic_batch(batch_size, (15, 20), dev, seed=100 * local_rank + i) for i in range(4)]
    if channels_last:
        batches = [(x.contiguous(memory_format=torch.channels_last), s) for x, s in batches]
    if graphs:      # hipGraphs for the static-shape convolutional trunk (student fwd+bwd, teacher fwd); see detection/graphs.py
        model.enable_graphs(batches[0][0], torch.bfloat16 if amp else None)
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                        gradient_as_bucket_view=True)
    fwd_bwd, opt_ms = [], []

    def one_step(i):
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
    for e0, e1, e2 in evs:
        fwd_bwd.append(e0.elapsed_time(e1))
        opt_ms.append(e1.elapsed_time(e2))
    finite = all(bool(torch.isfinite(v)) for v in losses.values())
    out = {"img_s": world * batch_size * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
           "batch_per_gpu": batch_size,
           "hip_graphs": "backbone + FPN + RPN convolutions: student forward/backward and teacher forward replayed from captured graphs" if graphs else "off",
           "memory_format": "channels_last" if channels_last else "contiguous (NCHW)", "image": "3x800x1344 (1333x800 padded to /32)", "n_gpus": world,
           "teacher_student_fwd_bwd_ms": sum(fwd_bwd) / len(fwd_bwd), "optimizer_step_ms": sum(opt_ms) / len(opt_ms),
           "nsgp_kernels_ms": update_ms + gemm_ms, "projected_layers": n_proj,
           "trainable_tensors": sum(len(g["params"]) for g in opt.param_groups),
           "losses_finite": finite, "loss_keys": sorted(losses.keys()),
           "detector_dtype": "bf16 autocast (replay-bank pass, losses, NSGP step fp32)" if amp else "f32",
           "parallelism": f"DDP x{world}: bucketed RCCL all-reduce of the gradients overlapped with backward" if world > 1 else "single",
           "detector": "nsgp_repre_amd.detection (R-50-FPN Faster R-CNN, stock recipe in plain PyTorch-ROCm: MIOpen convolutions, "
                       "hipBLASLt GEMMs; teacher predict + pseudo-label filter every step, as det:65-109)"}
    if graphs:
        model.disable_graphs()
    del net, model, opt
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end training img/s section")
    ap.add_argument("--e2e-steps", type=int, default=20)
    ap.add_argument("--e2e-f32", action="store_true", help="run the detector of the end-to-end section in fp32 instead of bf16 autocast")
    ap.add_argument("--e2e-graphs", action="store_true", help="end-to-end section with hipGraph capture of the convolutional trunk "
                    "(measured SLOWER on this stack: 47.9 vs 44.9 ms per step, so off by default)")
    ap.add_argument("--amp", action="store_true", help="run the replay head's GEMMs under bf16 autocast (measured 7x SLOWER "
                    "than fp32 on this image's hipBLASLt for the M=150 shapes: 12.4 vs 1.66 ms per step, so off by default)")
    args = ap.parse_args()

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

    table = r50_fpn_voc_parameter_table()
    gen = torch.Generator(device=dev).manual_seed(1234)
    params, names = [], []
    for n, shape, _ in table:
        params.append(torch.nn.Parameter(torch.randn(shape, device=dev, generator=gen) * 0.02))
        names.append(n)
    # the RoI bbox head is the product's own module (Shared2FCBBoxHeadTask, VOC 15+5 task 2); its 14
    # tensors join the optimizer table un-projected (ignore_keys=['rpn','roi_head'])
    bbox_head = N.roi_heads.Shared2FCBBoxHeadTask(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=20,
                                                  task_split=[0, 15, 20], task_id=2).to(dev)
    for n, p in bbox_head.named_parameters():
        params.append(p)
        names.append("roi_head.bbox_head." + n)
        table.append(("roi_head.bbox_head." + n, tuple(p.shape), False))
    opt = N.SGDNSCL(params, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)  # schedule_1x_sgdnscl.py:21
    opt.param_groups[0]["names"] = names
    cache = {}
    for i, (n, shape, proj) in enumerate(table):
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
    # Gradients live in ONE flat bucket with 16-byte-aligned slices (what DDP's
    # gradient_as_bucket_view gives): p.grad are views, filled by "backward" each step.
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
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.amp):   # configs[1] trains under bf16 autocast (--amp)
            loss = replay.replay_loss(replay.bbox_featss)["replay_loss"]["replay_loss_cls"]   # RePRE replay loss: forward
        loss.backward()                        # + backward: accumulates into the head's grad views
        h0 = time.perf_counter()
        opt.step()                             # NSGP projected step: 2 HIP launches
        host_step.append(time.perf_counter() - h0)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    opt.profile_begin(args.steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    allreduce_ms = None
    if world > 1:   # DDP's gradient all-reduce for this parameter set, measured on its own (not in `value`)
        for _ in range(3):
            dist.all_reduce(flat_grads)
        torch.cuda.synchronize()
        dist.barrier()
        t_ar = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat_grads)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t_ar) / 10 * 1e3
    e2e = None
    if not args.no_end_to_end:      # every rank takes part (DDP); the hot-path numbers above are already in the bag
        try:    # the hot-path measurement above must reach the JSON line whatever happens in this wider section
            e2e = end_to_end_training(N, dev, world, local_rank, cache, args.e2e_steps, 3, not args.e2e_f32, graphs=args.e2e_graphs)
            if not args.e2e_f32 and world == 1:    # the same step with an fp32 detector, for reference beside the bf16 number
                f32 = end_to_end_training(N, dev, world, local_rank, cache, max(4, args.e2e_steps // 2), 3, False)
                e2e["f32_detector"] = {k: f32[k] for k in ("img_s", "ms_per_step", "teacher_student_fwd_bwd_ms", "optimizer_step_ms", "losses_finite")}
        except Exception as exc:    # reported, never hidden
            import traceback
            traceback.print_exc()
            e2e = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        flops, abytes, ntiles, nproj = opt.plan_stats()
        # dominant kernel = the grouped projection GEMM: HIP events recorded by the library around
        # that launch, on the stream it is launched on, for every one of the K timed steps
        n_prof, update_ms, gemm_ms = opt.profile_end()
        nsgp_ms = update_ms + gemm_ms          # both launches of SGDNSCL.step, HIP-event timed
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "NSGP-RePRE hot-path img/s (SGDNSCL projected step + RePRE replay loss; detector fwd/bwd not in the timed region) + NSGP-projection step ms",
            "value": world * 1.0 / (ms_per_step / 1e3),
            "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "nsgp_step_ms": nsgp_ms,
            "host_ms_in_optimizer_step": 1e3 * sum(host_step[-args.steps:]) / max(1, len(host_step[-args.steps:])),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x2": "f32 (parameters, gradients, state, accumulation; the projection's products as 2-term fp16 splits on the matrix cores)",
                      "bf16x3": "f32 (parameters, gradients, state, accumulation; the projection's products as 3-term bf16 splits on the matrix cores)"
                      }.get(opt.uses_split_mfma(), "f32"), "data": "synthetic",
            "config": {"workload": "R-50-FPN VOC 15+5 task 2 (configs[1]): SGDNSCL step over 50 projected layers "
                                   "+ 112 plain tensors (41.2M params), replay loss on K=150 prototypes, 1 img/GPU/step",
                       "autocast": "bf16 for the replay head's GEMMs (as tools/train.py --amp does for the detector); parameters, gradients, optimizer state and the whole NSGP step fp32" if args.amp else "off",
                       "global_batch": world, "parallelism": f"replicas x{world} (no exchange step on this path)" if world > 1 else "single"},
            "ddp_grad_allreduce_ms": allreduce_ms,
            "roofline": roofline_block(opt.uses_split_mfma(), flops, abytes, gemm_ms, update_ms, nsgp_ms, n_prof, flat_numel_real, ntiles, nproj),
        }
        # opt-in low-rank form of the same projectors (north_star: g - U(U^T g)); timed separately so that
        # the headline numbers above stay those of the dense parity path
        if world == 1:      # the other MFMA paths of the same dense step, timed separately
            default_path = opt.split_mfma
            out["other_mfma_paths"] = {}
            for other in ("f16x2", "bf16x3", False):
                if other == opt.uses_split_mfma():
                    continue
                opt.split_mfma = other
                for _ in range(3):
                    one_step()
                torch.cuda.synchronize()
                opt.profile_begin(args.steps)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    one_step()
                torch.cuda.synchronize()
                o_elapsed = (time.perf_counter() - t1) / args.steps * 1e3
                _, o_update_ms, o_gemm_ms = opt.profile_end()
                out["other_mfma_paths"][other or "f32"] = {
                    "uses_split_mfma": opt.uses_split_mfma(), "ms_per_step": o_elapsed, "nsgp_step_ms": o_update_ms + o_gemm_ms,
                    "projection_kernel_ms": o_gemm_ms, "fp32_equivalent_tflops": flops / (o_gemm_ms * 1e-3) / 1e12,
                    "frac_of_fp32_matrix_peak": flops / (o_gemm_ms * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS}
            opt.split_mfma = default_path
        if world == 1:
            opt.low_rank = True
            for _ in range(3):
                one_step()
            torch.cuda.synchronize()
            opt.profile_begin(args.steps)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            lr_elapsed = (time.perf_counter() - t1) / args.steps * 1e3
            n_lr_prof, lr_update_ms, lr_gemm_ms = opt.profile_end()
            n_lr, lr_flops, lt1, lt2 = opt.lowrank_stats()
            out["lowrank_form"] = {"ms_per_step": lr_elapsed, "nsgp_step_ms": lr_update_ms + lr_gemm_ms,
                                   "projection_launches_ms": lr_gemm_ms, "layers": n_lr, "algorithmic_flops": lr_flops,
                                   "achieved_tflops": lr_flops / (lr_gemm_ms * 1e-3) / 1e12, "tiles_phase1": lt1, "tiles_phase2": lt2,
                                   "synthetic_rank": "r = D/16", "note": "opt-in (optimizer.low_rank=True); parity vs the dense form: tests/test_gpu_parity.py::test_low_rank_form_matches_dense_form"}
            opt.low_rank = False
            # the AdamW flavour of the same step (schedule_1x_adamwnscl.py:21: lr 1e-4, weight_decay 0.1) over the same
            # tensors, projectors and synthetic gradients -- row a3 of SURVEY 8; parity: the G1 adamw goldens
            adamw = N.AdamWNSCL(params, lr=1e-4, weight_decay=0.1, svd=True)
            adamw.param_groups[0]["names"] = names
            for n, shape, proj in table:
                if proj:
                    adamw.transforms[n] = opt.transforms[n]
            flat_grads.copy_(synth_flat)
            for _ in range(3):
                adamw.step()
            torch.cuda.synchronize()
            adamw.profile_begin(args.steps)
            for _ in range(args.steps):
                adamw.step()
            torch.cuda.synchronize()
            _, aw_update_ms, aw_gemm_ms = adamw.profile_end()
            out["adamw_nscl"] = {"nsgp_step_ms": aw_update_ms + aw_gemm_ms, "elementwise_kernel_ms": aw_update_ms,
                                 "projection_kernel_ms": aw_gemm_ms, "achieved_tflops": flops / (aw_gemm_ms * 1e-3) / 1e12,
                                 "elementwise_hbm_gbs": 7 * 4 * flat_numel_real / (aw_update_ms * 1e-3) / 1e9,
                                 "note": "AdamWNSCL.step, same table; elementwise bytes: g r, m r+w, v r+w, p r+w"}
            del adamw
        if e2e is not None:
            out["end_to_end"] = e2e
        if world == 1:
            out["once_per_task"] = once_per_task_units(dev)
        traffic_file = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(traffic_file):   # HBM bytes per launch from rocprofv3 --pmc passes of this same command
            tr = json.load(open(traffic_file))
            out["roofline"]["traffic"] = tr.get("nsgp_project_kernel_hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = tr.get("source")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(table)
            out["cpu_baseline"]["gpu_nsgp_step_speedup"] = out["cpu_baseline"]["step_ms"] / nsgp_ms
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
