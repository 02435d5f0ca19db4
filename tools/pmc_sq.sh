#!/bin/bash
# SQ counters of the decode (or encode) kernel per tile, optionally phase-limited (diagnostic library).
# usage: tools/pmc_sq.sh <enc|dec> <encLimit> <decLimit> "<counters pass 1>" ["<counters pass 2>" ...]
WHICH=$1; EL=$2; DL=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "$@"; do
i=$((i+1))
rm -rf gpurun_out/pmcsq_$i
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcsq_$i -- python3 tools/run_kernels.py $WHICH $EL $DL 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcsq_$i/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if ("encode" in k or "pack" in k) if "$WHICH" == "enc" else ("decode" in k):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c.replace("SQ_", ""): "%.5g" % (v[-1] / 12960) for c, v in d.items()})
PY
rm -rf gpurun_out/pmcsq_$i
done
