#!/bin/bash
# usage: tools/pmc2.sh <tag> <which> <encLimit> <decLimit>
tag=$1; which=$2; el=$3; dl=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmcb_${tag}_1 -- python3 tools/run_kernels.py $which $el $dl 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcb_${tag}_2 -- python3 tools/run_kernels.py $which $el $dl 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/pmcb_${tag}_3 -- python3 tools/run_kernels.py $which $el $dl 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for pas in (1,2,3):
    for f in glob.glob("gpurun_out/pmcb_${tag}_%d/**/*counter_collection.csv" % pas, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "huffman" not in k: continue
            k = "enc" if "encode" in k else "dec"
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in acc.items():
            # last dispatch of each kernel = steady state with the requested phase limit
            print("${tag}", k, {c: "%.4g" % v[-1] for c, v in d.items()})
PY
