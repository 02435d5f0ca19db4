#!/bin/bash
# instruction counts of the decode kernel per phase limit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dl in 1 2 3 0; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcc_$dl -- python3 tools/run_kernels.py dec 0 $dl 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcc_$dl/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "decode" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("declimit $dl", {c: "%.4g" % (v[-1]/12960) for c, v in acc.items()})
PY
done
