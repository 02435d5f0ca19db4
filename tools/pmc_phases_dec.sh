#!/bin/bash
# Instruction counts per tile of the decode kernel, cumulative per phase limit (diagnostic build; limit N = stop after phase N:
# 1 header, 6 lookup tables, 7 count table + text staged, 8 synchronisation pass, 2 Huffman -> M32 (write pass), 4 value starts marked, 5 border prologue; 262144: everything but the rows finished from the ring).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "dec 0 1" "dec 0 6" "dec 0 7" "dec 0 8" "dec 0 2" "dec 0 4" "dec 0 5" "dec 0 262144" "dec 0 0"; do
set -- $spec
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
rm -rf gpurun_out/pmcp
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcp -- python3 tools/run_kernels.py $1 $2 $3 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcp/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "k_huffman_decode<2>" in row["Kernel_Name"].replace("(anonymous namespace)::", ""): acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("dec limit=$3", {c.replace("SQ_", ""): "%.5g" % (v[-1] / 12960) for c, v in acc.items()})
PY
done
done
rm -rf gpurun_out/pmcp
