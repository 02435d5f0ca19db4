#!/bin/bash
# HBM traffic of the encode / decode kernels from the L2 memory-side counters (separate passes for
# FETCH_SIZE and WRITE_SIZE as MI355X_MICROARCH.md prescribes; KB units, FETCH_SIZE doubled on gfx950).
# usage: tools/pmc_hbm.sh <outdir> [workload] ["huffman canon lsop"]   (kernels of all listed codecs go into one JSON)
OUT=${1:-gpurun_out/hbm}; WL=${2:-etopo1}; CODECS=${3:-huffman}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
for k in $CODECS; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c/$k -- python3 tools/run_kernels.py both 0 0 3 $WL $k > $OUT/$c.$k.log 2>&1
done
done
python3 - <<PY
import sys; sys.path.insert(0, '.')
import csv, glob, collections, json, re
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                acc[re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"]).split("(")[0]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res[k][c] = v[-1]          # the last (warm) call: every kernel runs once per call (round 5: the fast decode kernel's
                                       # second run is an instantiation of its own)
import os
out = {"workload": "$WL", "commit": os.environ.get("GF_COMMIT", ""), "csrc_digest": __import__("gridfour_amd.build", fromlist=["x"]).csrc_digest(), "unit": "bytes per launch", "correction": "FETCH_SIZE KB x1024 x2 (gfx950 half-count), WRITE_SIZE KB x1024", "kernels": {}}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out["kernels"][k] = {"fetch_size_kb_raw": d["FETCH_SIZE"], "write_size_kb_raw": d["WRITE_SIZE"],
                             "read_bytes": d["FETCH_SIZE"] * 1024 * 2, "write_bytes": d["WRITE_SIZE"] * 1024,
                             "traffic": d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024}
json.dump(out, open("$OUT/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
