#!/bin/bash
# SQ counters of the dense backward's kernels (real build): where the wave cycles go.  One --pmc pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -f csv -d $R/gpurun_out/pmc_dense_t_$i -o p -- python $R/tools/ab_dense_t.py > $R/gpurun_out/pmc_dense_t_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_dense_t_$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("dense_t_bwd_long", "dense_t_bwd_short_kernel", "batch_t_kernel"):
        if k in n:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  find $R/gpurun_out/pmc_dense_t_$i -name "*.csv" -size +1M -delete
done
