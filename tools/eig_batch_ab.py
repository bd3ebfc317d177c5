import sys, time, os
sys.path[:0] = ["/root/repo", "/root/repo/oracle"]
import torch, nsgp_repre_amd as N, nsgp_oracle as O
dev = torch.device("cuda:0")
layers = O.resnet_fpn_projected_layers(50)
params, names, fea = [], [], {}
for idx, (n, cout, D) in enumerate(layers):
    params.append(torch.nn.Parameter(torch.empty(cout, D, device=dev))); names.append(n)
    gen = torch.Generator(device=dev).manual_seed(2000 + idx)
    X = torch.randn(4 * D, D, device=dev, generator=gen) * torch.logspace(0, -3, D, device=dev)[None, :]
    fea[n] = (X.t() @ X).contiguous(); del X
res = {}
for batch in (1, 8, 1, 8, 16):
    opt = N.SGDNSCL(params, lr=0.02, svd=True); opt.param_groups[0]["names"] = names; opt.eigh_batch = batch
    torch.cuda.synchronize(); t0 = time.perf_counter(); opt.get_eigens(fea); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("eigh_batch", batch, f"{t*1e3:.1f} ms", flush=True)
    res[batch] = {n: (opt.eigens[n]["eigen_value"].clone(), opt.eigens[n]["eigen_vector"].clone()) for n in names}
same_v = all(torch.equal(res[1][n][0], res[8][n][0]) for n in names)
dv = max((res[1][n][0] - res[8][n][0]).abs().max().item() / res[1][n][0].max().item() for n in names)
print("spectra bitwise equal:", same_v, "max rel diff", dv)
