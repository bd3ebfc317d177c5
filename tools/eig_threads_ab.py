"""get_eigens over the 50 R-50-FPN layers (SURVEY 8d covariances): solver calls issued from 1 / 2 / 4 / 8 host threads x eigh_batch 16 / 4.
Usage (GPU box): python tools/eig_threads_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
import nsgp_oracle as O  # noqa: E402
import nsgp_repre_amd as N  # noqa: E402

dev = torch.device("cuda:0")
layers = O.resnet_fpn_projected_layers(50)
params, names, fea = [], [], {}
for idx, (n, cout, D) in enumerate(layers):
    params.append(torch.nn.Parameter(torch.empty(cout, D, device=dev)))
    names.append(n)
    gen = torch.Generator(device=dev).manual_seed(2000 + idx)
    X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
    fea[n] = (X.t() @ X).contiguous()
    del X
ref = None
for threads, batch in ((1, 16), (2, 16), (4, 16), (8, 16), (4, 4), (8, 4), (8, 2), (1, 16), (4, 16)):
    opt = N.SGDNSCL(params, lr=0.02, svd=True)
    opt.param_groups[0]["names"] = names
    opt.eigh_batch, opt.eigh_threads = batch, threads
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.get_eigens(fea)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    vals = {n: opt.eigens[n]["eigen_value"].clone() for n in names}
    if ref is None:
        ref = vals
    same = all(torch.equal(vals[n], ref[n]) for n in names)
    print(f"threads {threads} eigh_batch {batch}: {t * 1e3:.1f} ms   spectra bitwise equal to the first run: {same}", flush=True)
