#!/bin/bash
# Instruction counts per tile of k_canon_decode, cumulative per phase limit (diagnostic build: 1 tables, 2 + synchronisation pass,
# 3 + value pass; 0 everything, i.e. + the predictor inverse).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lim in 1 2 3 0; do
rm -rf gpurun_out/pmcp
GVRS_HIP_DIAG=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcp -- python3 tools/run_kernels.py dec 0 $lim 2 etopo1 canon > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcp/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_canon_decode" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("canon dec limit=$lim", {c.replace("SQ_", ""): "%.5g" % (v[-1] / 12960) for c, v in acc.items()})
PY
done
rm -rf gpurun_out/pmcp
