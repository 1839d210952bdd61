import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import colpali_amd as amd
from colpali_amd import loss as L
z = np.load("/root/repo/tests/golden/loss_small.npz")
Q = torch.from_numpy(z["Q"]).to(torch.bfloat16).cuda(); D = torch.from_numpy(z["D"]).to(torch.bfloat16).cuda()
B, C = Q.shape[0], D.shape[0]
raw = amd.loss.maxsim(Q, D).float()
lib = amd._lib.lib()
G = torch.zeros((B, C), dtype=torch.float32, device="cuda")
ws = torch.zeros((lib.msim_loss_epilogue_workspace_bytes(B),), dtype=torch.uint8, device="cuda")
out = torch.empty(3, dtype=torch.float32, device="cuda")
rc = lib.msim_loss_epilogue(1, raw.data_ptr(), C, B, C, Q.data_ptr(), 0, Q.shape[1], 128, 0, 0.02, 0, 0, 0.95, 0.5, G.data_ptr(), None, None, None, ws.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
l64 = raw.double().cpu() / 0.02
p = torch.softmax(l64, dim=1)
g64 = (p - torch.eye(B, C, dtype=torch.float64)) / 0.02 / B
print("rc", rc, "loss", float(out[0]), "truth", float(torch.nn.functional.cross_entropy(l64, torch.arange(B))))
print("max abs err per row", (G.cpu().double() - g64).abs().max(dim=1).values.tolist())
print("G row0", G[0].tolist())
print("g64 row0", g64[0].tolist())
s32 = raw.clone().requires_grad_(True)
torch.nn.functional.cross_entropy(s32 / 0.02, torch.arange(B, device="cuda")).backward()
print("torch fp32 err per row", (s32.grad.cpu().double() - g64).abs().max(dim=1).values.tolist())
