import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd
L = colpali_amd._lib.lib()
L.msim_debug_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.msim_debug_probe.restype = ctypes.c_int
dev = torch.device("cuda:0")
names = ["128B x32rows d4 (K3 today)", "256B x32rows d2", "512B x16rows d2", "256B x16rows d4", "1KiB x16rows d1", "512B x32rows d2, 4 waves", "128B x16rows d8"]
for H in (2048,):
    M = 515000 // 256 * 256
    X = torch.randn((M, H), dtype=torch.float32, device=dev).to(torch.bfloat16) if H == 2048 else torch.ones((M, H), dtype=torch.bfloat16, device=dev)
    sink = torch.zeros(4, device=dev)
    for v, name in enumerate(names):
        ms = []
        for r in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); rc = L.msim_debug_probe(v, X.data_ptr(), M, H, sink.data_ptr(), torch.cuda.current_stream().cuda_stream); b.record()
            torch.cuda.synchronize(); assert rc == 0, rc
            ms.append(a.elapsed_time(b))
        t = sorted(ms)[1]
        print(f"H={H} {name:28s} {t:7.3f} ms  {M*H*2/t/1e6:6.0f} GB/s", flush=True)
    del X
