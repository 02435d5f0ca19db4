# LSOP12 decode: per-kernel times of the shipping library against experiment builds (given as "name:-DFLAG,-DFLAG ..."), and the HBM counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06ab}; mkdir -p $O
AB_LINES=8 bash tools/ab_kernels.sh "$2" $O 120 150 12960 lsop 2>&1 | grep -v "predict16\|pack2\|copyBuffer"
if [ -n "$3" ]; then bash tools/pmc_hbm.sh $O/hbm etopo1 lsop > $O/pmc_hbm.txt 2>&1; python3 - <<PY
import json; d=json.load(open("$O/hbm/hbm_traffic.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["traffic"])[:10]: print("%-40s read %.3f GB write %.3f GB" % (k[:40], v["read_bytes"]/1e9, v["write_bytes"]/1e9))
PY
fi
