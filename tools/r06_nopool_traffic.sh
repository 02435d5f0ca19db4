cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nopool; mkdir -p $O
for v in "" nopool; do
  echo "== ${v:-shipping}"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p; GVRS_HIP_VARIANT=$v rocprofv3 --pmc $c --output-format csv -d $O/p -- python3 tools/run_kernels.py dec 0 0 3 etopo1 huffman > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "$c" and "decode" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items(): print("$c", k, "%.3f GB" % (v[-1] * 1024 * (2 if "$c" == "FETCH_SIZE" else 1) / 1e9))
PY
  done
  GVRS_HIP_VARIANT=$v python3 tools/shape_time.py 120 150 12960 | tail -1
done
