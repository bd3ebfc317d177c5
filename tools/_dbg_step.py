import sys, os, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch, nsgp_repre_amd as N
dev = torch.device("cuda:0")
ps = [torch.nn.Parameter(torch.randn(64, 256, device=dev)), torch.nn.Parameter(torch.randn(7, device=dev))] + [torch.nn.Parameter(torch.randn(1000, device=dev)) for _ in range(160)]
opt = N.SGDNSCL(ps, lr=0.02, momentum=0.9, weight_decay=1e-4, svd=True)
opt.param_groups[0]["names"] = ["neck.a.weight", "x.bias"] + [f"p{i}" for i in range(160)]
Q, _ = torch.linalg.qr(torch.randn(256, 256))
opt.set_basis("neck.a.weight", Q.contiguous().to(dev), 20)
for p in ps: p.grad = torch.randn_like(p)
for i in range(5):
    opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ok = opt._structure_unchanged(); t1 = time.perf_counter()
    print(i, "unchanged:", ok, f"{(t1 - t0) * 1e6:.1f} us")
    if not ok:
        f = opt._fast
        from operator import is_
        print("  ptrs", [p.data_ptr() for p in f["plist"]] == f["ptrs"], "states", all(map(is_, map(opt.state.get, f["plist"]), f["states"])), "ntr", len(opt.transforms), f["n_transforms"])
        for n, rP, rver, rptr in f["proj"]:
            P = opt.transforms.get(n); print("  proj", n, P is rP, P._version, rver, P.data_ptr() == rptr)
ts = []
for i in range(50):
    t0 = time.perf_counter(); opt.step(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host per step us:", sorted(ts)[25] * 1e6)
