import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
import nsgp_repre_amd as N
from nsgp_repre_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (C, H, W) in ((64, 6, 6), (64, 40, 56), (192, 9, 13)):
    x = torch.randn(1, C, H, W).abs()
    X = F.unfold(x.double(), 3, padding=1)[0].t()
    ref = (X.t() @ X)
    prev = ops.cov_set_corr_mode(2)
    plan = ops.CovGroupPlan([(1, C, H, W, (3, 3), (1, 1), (1, 1))], dev)
    ops.cov_set_corr_mode(prev)
    cov = plan.run([x.to(dev)], [None])[0].double().cpu()
    plan.close()
    err = (cov - ref).abs().view(C, 9, C, 9).amax(dim=(0, 2))
    mag = ref.abs().view(C, 9, C, 9).amax(dim=(0, 2))
    torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
    print(C, H, W, "n_corr", plan.n_correlation_form, "max ref", float(ref.abs().max()))
    print((err / mag))
    # signed mean difference per tap pair relative
    print(((cov - ref).view(C, 9, C, 9).mean(dim=(0, 2)) / ref.view(C, 9, C, 9).mean(dim=(0, 2))))
