# state of the round on one box: GPU tests, the default bench line (with its sub-records), per-kernel times of the headline, LSOP12 and rough runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06s}; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
( time timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
for c in "" lsop canon; do
  rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py ${c:+--codec $c} --cpu-sample-tiles 0 > $O/bench_under_rocprof${c:+_$c}.json 2>> $O/rocprof.log
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats${c:+_$c}.csv; rm -rf $O/prof
  echo "== ${c:-huffman}"; python3 tools/kernel_times.py $O/kernel_stats${c:+_$c}.csv | head -12
done
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload etopo1_rough --cpu-sample-tiles 0 > /dev/null 2>> $O/rocprof.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_etopo1_rough.csv; rm -rf $O/prof
echo "== rough"; python3 tools/kernel_times.py $O/kernel_stats_etopo1_rough.csv | head -8
[ -x tools/bin/single_tile_latency ] && tools/bin/single_tile_latency > $O/single_tile_latency.json 2>&1; cat $O/single_tile_latency.json
