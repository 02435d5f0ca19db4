#!/bin/bash
# usage: tools/pmc.sh <tag> <which> <encLimit> <decLimit>   -- two PMC passes (8 SQ counters each), CSV summaries
tag=$1; which=$2; el=$3; dl=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_${tag}_1 -- python3 tools/run_kernels.py $which $el $dl 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC --output-format csv -d gpurun_out/pmc_${tag}_2 -- python3 tools/run_kernels.py $which $el $dl 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for pas in (1,2):
    for f in glob.glob("gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv" % pas, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-24:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, d in acc.items():
            if "huffman" not in k: continue
            print("${tag}", k, {c: "%.3g" % v for c, v in d.items()})
PY
