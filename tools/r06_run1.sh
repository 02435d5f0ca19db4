cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06a}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_random_shapes.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/ab.sh noplane > $O/ab.txt 2>&1; cat $O/ab.txt
GF_DEM_STYLE=1 bash tools/ab.sh noplane > $O/ab_rough.txt 2>&1; cat $O/ab_rough.txt
bash tools/ab.sh noplane 200 200 1024 > $O/ab_dem1024.txt 2>&1; cat $O/ab_dem1024.txt
bash tools/ab.sh noplane 200 200 11664 > $O/ab_gebco.txt 2>&1; cat $O/ab_gebco.txt
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --cpu-sample-tiles 0 > $O/bench_under_rocprof.json 2> $O/rocprof.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv; rm -rf $O/prof
python3 tools/kernel_times.py $O/kernel_stats.csv | head -12
( time timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
