"""Full-size timings of the once-per-task kernels (GPU box only): covariance SYRK, projector build,
prototype similarity / means, and torch.linalg.eigh for reference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import nsgp_repre_amd as N
from nsgp_repre_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


print("== covariance (one hooked conv, 800x1344 input strides)", flush=True)
for name, cin, k, s, p, hw in (("neck.fpn_convs.0 3x3 D=2304 L=67200", 256, (3, 3), (1, 1), (1, 1), (200, 336)),
                               ("layer2 conv2 3x3 D=1152 L=16800", 128, (3, 3), (1, 1), (1, 1), (100, 168)),
                               ("layer4 conv2 3x3 D=4608 L=1050", 512, (3, 3), (1, 1), (1, 1), (25, 42)),
                               ("layer1 conv1 1x1 D=64 L=67200", 64, (1, 1), (1, 1), (0, 0), (200, 336)),
                               ("layer3 conv1 1x1 D=1024 L=4200", 1024, (1, 1), (1, 1), (0, 0), (50, 84)),
                               ("stem 7x7 s2 D=147 L=268800", 3, (7, 7), (2, 2), (3, 3), (800, 1344))):
    x = torch.randn(1, cin, *hw, device=dev).abs()
    D = cin * k[0] * k[1]
    Ho, Wo = (hw[0] + 2 * p[0] - k[0]) // s[0] + 1, (hw[1] + 2 * p[1] - k[1]) // s[1] + 1
    L = Ho * Wo
    cov = ops.cov_accumulate_conv2d(x, k, s, p)
    ws = torch.empty(ops.cov_workspace_bytes(cin, hw[0], hw[1], k, s, p), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: ops.cov_accumulate_conv2d(x, k, s, p, cov, ws))
    ref_fl = 2.0 * L * D * D
    print(f"{name:42s} {ms:9.3f} ms   {ref_fl / ms / 1e9:7.1f} TF (reference FLOPs 2LD^2)   ws {ws.numel() / 2**20:7.1f} MiB", flush=True)
    if L * D * 4 < 2**31:
        X = torch.nn.functional.unfold(x, k, padding=p, stride=s)[0].t()
        msr = timeit(lambda: X.t() @ X, reps=3, warm=1)
        print(f"{'   torch unfold^T@unfold (rocBLAS) same GPU':42s} {msr:9.3f} ms   rel err {((cov / 8 if False else ops.cov_accumulate_conv2d(x, k, s, p)) - X.t() @ X).abs().max().item() / (X.t() @ X).abs().max().item():.2e}", flush=True)

print("== projector build", flush=True)
for D in (1024, 2304, 4608):
    Q, _ = torch.linalg.qr(torch.randn(D, D, device=dev))
    Q = Q.contiguous()
    r = D // 16
    ms = timeit(lambda: ops.build_projector(Q, r, True))
    print(f"D={D} r={r}: {ms:8.3f} ms   {2.0 * D * D * (D - r) / ms / 1e9:7.1f} TF (reference FLOPs)", flush=True)

print("== eigh (torch/rocSOLVER)", flush=True)
for D in (1024, 2304, 4608):
    X = torch.randn(2 * D, D, device=dev) * torch.logspace(0, -3, D, device=dev)
    C = (X.t() @ X).contiguous()
    t0 = time.perf_counter(); lam, Qe = torch.linalg.eigh(C); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"D={D}: eigh {t1 - t0:8.3f} s", flush=True)

print("== prototype similarity / means (D=12544)", flush=True)
for n in (300, 1500, 6000):
    F = torch.relu(torch.randn(n, 12544, device=dev))
    ms = timeit(lambda: ops.sim_counts(F, 0.6), reps=3)
    mm = timeit(lambda: ops.masked_mean(F), reps=3)
    print(f"N={n}: sim_counts {ms:9.3f} ms  {2.0 * n * n * 12544 / ms / 1e9:7.1f} TF (reference FLOPs 2N^2D)   all-row mean {mm:7.3f} ms  {4.0 * n * 12544 / mm / 1e6:7.1f} GB/s", flush=True)

print("== EWC regulariser (R-50: 106 BN tensors, T=1), fused vs the reference's per-parameter loop", flush=True)
import sys as _s
_s.path.insert(0, os.path.join(ROOT, "oracle"))
import nsgp_oracle as O
sizes = [64] * 8 + [256] * 6 + [128] * 10 + [512] * 10 + [256] * 16 + [1024] * 14 + [512] * 8 + [2048] * 6
sizes = (sizes + sizes)[:106]
params = {f"backbone.l{i}.bn.weight": torch.nn.Parameter(torch.randn(n, device=dev)) for i, n in enumerate(sizes)}
terms = {"importance": {n: [torch.rand(1, p.numel(), device=dev)] for n, p in params.items()},
         "task_param": {n: [torch.randn(1, p.numel(), device=dev)] for n, p in params.items()}}
reg = N.runner.ewc.EWCRegulariser(params, terms)
def fused():
    for p in params.values(): p.grad = None
    reg().backward()
def loop():
    for p in params.values(): p.grad = None
    O.ewc_loss(params, terms["importance"], terms["task_param"]).backward()
tf, tl = timeit(fused, reps=10), timeit(loop, reps=10)
t0 = time.perf_counter(); [fused() for _ in range(20)]; torch.cuda.synchronize(); wf = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter(); [loop() for _ in range(20)]; torch.cuda.synchronize(); wl = (time.perf_counter() - t0) / 20 * 1e3
print(f"fused: {tf:.3f} ms GPU / {wf:.3f} ms wall    per-parameter torch loop (the reference's route on this GPU): {tl:.3f} ms GPU / {wl:.3f} ms wall", flush=True)
