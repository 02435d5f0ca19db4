#!/bin/bash
# Regenerates the judged artifacts of a round on the GPU box: full GPU test suite, smoke, bench lines of every codec,
# rocprofv3 kernel statistics of the default bench command, HBM traffic (PMC passes).
# usage: GF_COMMIT=<hash> tools/round_artifacts.sh <tag>        (writes gpurun_out/<tag>/; the hash goes into the replayed PMC files)
TAG=${1:-r01_vX}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout 1200 python3 -m pytest tests -m gpu -q -rs > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
# HBM traffic first (shipping library, all three integer codecs): the bench lines below replay it
GF_COMMIT=$GF_COMMIT bash tools/pmc_hbm.sh $OUT/hbm etopo1 "huffman canon lsop" > $OUT/pmc_hbm.txt 2>&1
cp $OUT/hbm/hbm_traffic.json $OUT/hbm_traffic.json 2>/dev/null && cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
rm -rf $OUT/hbm/FETCH_SIZE $OUT/hbm/WRITE_SIZE
# instruction counts of the shipping kernels (bench.py's roofline_issue replays them), and why the fraction is what it is
GF_COMMIT=$GF_COMMIT bash tools/pmc_issue.sh $OUT/issue_counts.json etopo1 "huffman canon lsop" > $OUT/pmc_issue.txt 2>&1
cp $OUT/issue_counts.json profiles/issue_counts.json 2>/dev/null
bash tools/pmc_why.sh $OUT both > $OUT/pmc_why.log 2>&1
bash tools/pmc_phases_dec.sh > $OUT/pmc_phases_dec.txt 2>&1
bash tools/pmc_phases_canon.sh > $OUT/pmc_phases_canon.txt 2>&1
[ -x tools/bin/issue_rate ] && tools/bin/issue_rate > $OUT/issue_rate.json 2>&1
[ -x tools/bin/single_tile_latency ] && tools/bin/single_tile_latency > $OUT/single_tile_latency.json 2>&1
GF_COMMIT=$GF_COMMIT bash tools/pmc_issue.sh $OUT/issue_counts_rough.json etopo1_rough "huffman" > $OUT/pmc_issue_rough.txt 2>&1
timeout 600 python3 bench.py 2>/dev/null | tail -1 > $OUT/bench.json
timeout 600 python3 bench.py --workload etopo1_nulls --cpu-sample-tiles 0 2>/dev/null | tail -1 > $OUT/bench_etopo1_nulls.json
timeout 600 python3 bench.py --workload etopo1_rough --cpu-sample-tiles 0 2>/dev/null | tail -1 > $OUT/bench_etopo1_rough.json
rm -rf $OUT/profr; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profr -- python3 bench.py --workload etopo1_rough --cpu-sample-tiles 0 > /dev/null 2>> $OUT/rocprof.log
f=$(find $OUT/profr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_etopo1_rough.csv; rm -rf $OUT/profr
timeout 900 python3 tools/codec_master_rate.py 2>/dev/null | tail -1 > $OUT/codec_master_rate.json
timeout 900 python3 tools/host_path_rate.py etopo1 2>/dev/null | tail -1 > $OUT/host_path_etopo1.json
timeout 900 python3 tools/host_path_rate.py etopo1 2 2>/dev/null | tail -1 > $OUT/host_path_etopo1_multi2.json
timeout 900 python3 tools/host_path_rate.py gebco_full 2>/dev/null | tail -1 > $OUT/host_path_gebco_full.json
timeout 900 python3 tools/float_host_rate.py 4096 6 2>/dev/null | tail -1 > $OUT/float_host_path.json
GF_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --cpu-sample-tiles 0 2>/dev/null | tail -1 > $OUT/bench_single_process_2shards_one_gpu.json
timeout 600 python3 bench.py --codec canon 2>/dev/null | tail -1 > $OUT/bench_canon.json
timeout 600 python3 bench.py --codec lsop 2>/dev/null | tail -1 > $OUT/bench_lsop.json
timeout 600 python3 bench.py --workload dem1024 2>/dev/null | tail -1 > $OUT/bench_dem1024.json
timeout 600 python3 bench.py --workload gebco_shard --cpu-sample-tiles 0 2>/dev/null | tail -1 > $OUT/bench_gebco_shard.json
timeout 600 python3 bench.py --codec float --workload float256 2>/dev/null | tail -1 > $OUT/bench_float256.json
timeout 900 python3 bench.py --codec lsop --workload float256_lsop 2>/dev/null | tail -1 > $OUT/bench_float256_lsop.json
timeout 300 python3 tools/lsop_recon_time.py 2>/dev/null | tail -1 > $OUT/lsop_kernels_etopo1.json
timeout 300 python3 tools/lsop_recon_time.py 256 256 4096 2>/dev/null | tail -1 > $OUT/lsop_kernels_256x256.json
rm -rf $OUT/profn; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profn -- python3 bench.py --workload etopo1_nulls --cpu-sample-tiles 0 > /dev/null 2>> $OUT/rocprof.log
f=$(find $OUT/profn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_etopo1_nulls.csv; rm -rf $OUT/profn
for c in "" canon lsop; do
  rm -rf $OUT/prof$c
  if [ -z "$c" ]; then
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --cpu-sample-tiles 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.log
  else
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$c -- python3 bench.py --codec $c --cpu-sample-tiles 0 > /dev/null 2>> $OUT/rocprof.log
  fi
  f=$(find $OUT/prof$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats${c:+_$c}.csv
  rm -rf $OUT/prof$c
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proff -- python3 bench.py --codec float --workload float256 --cpu-sample-tiles 0 > /dev/null 2>> $OUT/rocprof.log
f=$(find $OUT/proff -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_float.csv; rm -rf $OUT/proff
# the counters' calibration: 1 GiB read / written once with 4-, 8- and 16-byte accesses
if [ -x tools/bin/fetch_calib ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/calib; rocprofv3 --pmc $c --output-format csv -d $OUT/calib -- tools/bin/fetch_calib > /dev/null 2>&1
    python3 - <<PY >> $OUT/counter_calibration.txt
import csv, glob
for f in glob.glob("$OUT/calib/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "$c" and "fillBuffer" not in row["Kernel_Name"]:
            print("$c", row["Kernel_Name"].split("(")[0], "counter x 1024 / bytes moved =", round(float(row["Counter_Value"]) * 1024 / 2**30, 4))
PY
  done
  rm -rf $OUT/calib
fi
# round 5: the rough batch's phase stamps, and whether the fast decode kernel's two runs overlap (start / end per dispatch)
GF_DEM_STYLE=1 timeout 300 python3 tools/phase_cycles_dec.py > $OUT/phase_cycles_dec_rough.txt 2>&1
rm -rf $OUT/trace; GF_DEM_STYLE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/shape_time.py 120 150 12960 > /dev/null 2>> $OUT/rocprof.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/overlap_trace.py $f > $OUT/decode_overlap_rough.txt 2>&1; rm -rf $OUT/trace
# ... HBM traffic of the rough batch (bench.py's rough.roofline.traffic replays it) and the device-side timeline of a one-tile call
GF_COMMIT=$GF_COMMIT bash tools/pmc_hbm.sh $OUT/hbm_rough etopo1_rough huffman > $OUT/pmc_hbm_rough.txt 2>&1
cp $OUT/hbm_rough/hbm_traffic.json $OUT/hbm_traffic_rough.json 2>/dev/null && cp $OUT/hbm_traffic_rough.json profiles/hbm_traffic_rough.json; rm -rf $OUT/hbm_rough
if [ -x tools/bin/single_tile_latency ]; then
  rm -rf $OUT/tr; rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- tools/bin/single_tile_latency > /dev/null 2>> $OUT/rocprof.log
  f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/single_tile_timeline.py $f > $OUT/single_tile_timeline.txt 2>&1; rm -rf $OUT/tr
fi
head -c 600 $OUT/bench.json; echo; head -5 $OUT/kernel_stats.csv
