#!/bin/bash
# Wave-instruction counts per tile of the kernels of a codec's encode and decode (rocprofv3 --pmc, shipping library): what the
# bench line's roofline_issue replays.  usage: GF_COMMIT=<hash> tools/pmc_issue.sh <out.json> [workload] ["huffman canon lsop"]
OUT=${1:-gpurun_out/issue_counts.json}; WL=${2:-etopo1}; CODECS=${3:-huffman}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcissue
for k in $CODECS; do
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmcissue/$k/$i -- python3 tools/run_kernels.py both 0 0 2 $WL $k > /dev/null 2>&1
  done
done
python3 - <<PY
import sys; sys.path.insert(0, '.')
import csv, glob, collections, json, os, re
nt = {"etopo1": 12960, "etopo1_rough": 12960, "dem1024": 1024}["$WL"]
res = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmcissue/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).split("(")[0].replace("void ", "").strip()
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res[k][c.replace("SQ_", "").lower()] = round(v[-1] / nt, 2)     # last (warm) call, per tile
out = {"workload": "$WL", "tiles": nt, "commit": os.environ.get("GF_COMMIT", ""), "csrc_digest": __import__("gridfour_amd.build", fromlist=["x"]).csrc_digest(), "unit": "wave-instructions (cycle counters: quad-cycles) per tile, summed over the tile's waves",
       "kernels": {k: d for k, d in sorted(res.items()) if d.get("insts_valu", 0) > 1}}
json.dump(out, open("$OUT", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmcissue
