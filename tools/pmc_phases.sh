#!/bin/bash
# VALU / SALU / LDS / VMEM instruction counts per tile of the encode and decode kernels, cumulative per phase limit
# (phase limit N = the kernel stops after phase N; 0 = complete).  usage: tools/pmc_phases.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "enc 1 0" "enc 2 0" "enc 0 0" "dec 0 1" "dec 0 2" "dec 0 3" "dec 0 0"; do
set -- $spec
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcp_$1_$2_$3 -- python3 tools/run_kernels.py $1 $2 $3 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcp_$1_$2_$3/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if ("encode" if "$1" == "enc" else "decode") in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("$1 limit enc=$2 dec=$3", {c.replace("SQ_", ""): "%.5g" % (v[-1] / 12960) for c, v in acc.items()})
PY
done
