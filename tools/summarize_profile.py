#!/usr/bin/env python
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/prof.sh on the GPU box) into the small,
committed evidence under profiles/: the rocprofv3 --kernel-trace --stats rows of our kernels and the
per-launch PMC counters, with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950 -> doubled; unit is KiB... see below).

usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>
"""
import csv
import json
import os
import sys
from collections import defaultdict


def main(src, dst_prefix):
    out = {"source": src}
    stats_path = os.path.join(src, "trace", "trace_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats_path)))
    ours = [r for r in rows if "msim::" in r["Name"]]
    out["kernel_stats"] = [{k: (r[k] if k == "Name" else float(r[k])) for k in r} for r in ours]
    with open(dst_prefix + "_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:12]:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    counters = defaultdict(lambda: defaultdict(list))
    meta = {}
    for sub in sorted(os.listdir(src)):
        p = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            if "msim::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0]
            counters[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[name] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    pmc = {}
    for name, cs in counters.items():
        d = {"launch_config": meta[name]}
        for c, vals in cs.items():
            vals = vals[3:] if len(vals) > 6 else vals       # skip warm-up launches
            d[c] = sum(vals) / len(vals)
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB... the values match bytes/1024 of the known stream;
        # gfx950 correction: FETCH_SIZE x2 for 16 B/lane coalesced streams (guide, HBM section)
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes_per_launch_corrected"] = d["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes_per_launch_uncalibrated"] = d["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
            d["note_mfma"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs*4 SIMDs) ~ MFMA pipe utilisation"
        pmc[name] = d
    out["pmc"] = pmc
    for b in ("bench_trace.json", "bench_pmc_fetch.json"):
        p = os.path.join(src, b)
        if os.path.exists(p) and os.path.getsize(p):
            try:
                out[b] = json.loads(open(p).read().strip().splitlines()[-1])
            except Exception as e:  # noqa
                out[b] = f"unparsed: {e}"
    with open(dst_prefix + "_summary.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernel_stats"], indent=1))
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
