#!/usr/bin/env python
"""msim_probe_stream(MSIM_PROBE_PIECES128B): K3's hidden-state access pattern BY ITSELF (128-byte pieces of 32 rows per wave, 4-deep
ring, nt, no arithmetic, no output) over matrices of ~2 GiB with different row widths: does the 4 KiB row stride of hidden 2048 hurt
the bare pattern, and do neighbouring strides (2048 + 64 elements = 4224 bytes) escape it?  Also the 256- and 512-byte-piece patterns."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
from tools import probe
L = probe.lib()          # tools/probe/libmaxsim_probe.so (include/maxsim_probe.h): make -C tools/probe
sink = torch.zeros(4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream()
for H in [int(x) for x in os.environ.get("PS_WIDTHS", "1536,1984,2048,2112,2560,3072,3584,4096").split(",")]:
    rows = (2 << 30) // (2 * H) // 256 * 256
    x = torch.empty((rows, H), dtype=torch.bfloat16, device=dev).normal_()
    line = f"row width {H:5d} ({2 * H:5d} B stride), {rows * 2 * H / 2**30:.2f} GiB:"
    for variant, name in ((1, "128-B x 32 rows"), (0, "256-B x 32 rows"), (2, "512-B x 16 rows"), (11, "128-B, rows 16 KiB apart"),
                          (12, "128-B, chunk skewed per row"), (13, "128-B, chunk skewed per instruction"),
                          (21, "512-B x 8 rows x 3 slots (96 KiB)"), (22, "512-B x 16 rows x 2 slots (128 KiB)"),
                          (23, "256-B x 16 rows x 3 slots (96 KiB)")):
        piece = {1: 128, 0: 256, 2: 512, 11: 128, 12: 128, 13: 128, 21: 512, 22: 512, 23: 256}[variant]
        if (2 * H) % piece:
            line += f"   {name}: n/a"
            continue
        ms = []
        for i in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            rc = L.msim_probe_stream(variant, x.data_ptr(), rows, H, sink.data_ptr(), st.cuda_stream)
            b.record(st)
            torch.cuda.synchronize()
            assert rc == 0, L.msim_probe_last_error()
            if i >= 2:
                ms.append(a.elapsed_time(b))
        t = sorted(ms)[len(ms) // 2]
        line += f"   {name}: {rows * 2 * H / t / 1e6:5.0f} GB/s"
    print(line, flush=True)
    del x
