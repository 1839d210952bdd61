import sys, traceback
sys.path.insert(0, "/root/repo")
import torch, colpali_amd as amd
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
Q = torch.nn.functional.normalize(torch.randn(16, 32, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
D = torch.nn.functional.normalize(torch.randn(16, 200, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
N = torch.nn.functional.normalize(torch.randn(16, 2, 50, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
for cls, kw, neg in (("ColbertSigmoidLoss", {}, False), ("ColbertPairwiseCELoss", {}, False), ("ColbertLoss", dict(pos_aware_negative_filtering=True), False),
                ("ColbertNegativeCELoss", {}, True), ("ColbertPairwiseNegativeCELoss", dict(use_smooth_max=True), True)):
    fn = getattr(amd, cls)(**kw)
    q, d, n = Q.clone().requires_grad_(True), D.clone().requires_grad_(True), N.clone().requires_grad_(True)
    args = (q, d, n) if neg else (q, d)
    fn(*args).backward()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        fn(*args).backward()
        print(cls, kw, "no synchronising call")
    except Exception:
        print(cls, kw, "SYNC:")
        traceback.print_exc(limit=8)
    torch.cuda.set_sync_debug_mode("default")
