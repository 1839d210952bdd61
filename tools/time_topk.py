import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch, colpali_amd as amd
s = torch.randn(1000, 125000, device="cuda") * 0.1 + 9.3
for k in (10, 100):
    amd.topk(s, k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): amd.topk(s, k)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"topk k={k} 1000x125000: {ms:.3f} ms = {s.numel()*4/ms/1e6:.0f} GB/s of score reads")
