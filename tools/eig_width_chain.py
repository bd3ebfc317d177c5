"""How long ONE torch.linalg.eigh takes per width on the GPU (fp32, rocSOLVER syevd), alone and as a batch of the multiplicity the width has in
R-50-FPN's table: the sequential column chain of the widest matrix is the floor of get_eigens whatever the number of issuing threads."""
import time
import torch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for D, mult in ((4608, 3), (2304, 11), (2048, 3), (1152, 4), (1024, 7), (576, 3), (512, 7), (256, 8)):
    X = torch.randn(2 * D, D, device=dev, generator=g) * torch.logspace(0, -3, D, device=dev)[None, :]
    C = (X.t() @ X).contiguous()
    del X
    for batch in (1, mult):
        A = C[None].repeat(batch, 1, 1).contiguous() if batch > 1 else C
        torch.linalg.eigh(A)
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            torch.linalg.eigh(A)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"D = {D:5d}  batch {batch:2d}: {1e3 * min(ts):8.1f} ms", flush=True)
        del A
