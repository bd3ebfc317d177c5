"""torch.linalg.eigh per width under the two linalg backends of this PyTorch build (default = rocSOLVER via hipSOLVER; 'magma' if built in)."""
import time
import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
print("has magma:", torch.cuda.has_magma, flush=True)
for backend in ("default", "magma"):
    try:
        torch.backends.cuda.preferred_linalg_library(backend)
    except Exception as exc:
        print(backend, "not selectable:", exc)
        continue
    for D in (4608, 2304, 1024, 256):
        X = torch.randn(2 * D, D, device=dev, generator=g) * torch.logspace(0, -3, D, device=dev)[None, :]
        C = (X.t() @ X).contiguous()
        del X
        try:
            lam, Q = torch.linalg.eigh(C)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lam, Q = torch.linalg.eigh(C)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res = float(((C @ Q - Q * lam[None, :]).norm() / C.norm()).item())
            print(f"{backend:8s} D = {D:5d}: {1e3 * dt:8.1f} ms   residual {res:.2e}", flush=True)
        except Exception as exc:
            print(backend, D, "failed:", str(exc)[:200], flush=True)
