cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06l}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_lsop.py tests/test_gpu_lsop_head.py tests/test_gpu_lsop_plane.py -m gpu -x -q > $O/pytest_lsop.txt 2>&1; tail -15 $O/pytest_lsop.txt
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --codec lsop --cpu-sample-tiles 0 > $O/bench_lsop.json 2>> $O/rocprof.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_lsop.csv; rm -rf $O/prof
python3 tools/kernel_times.py $O/kernel_stats_lsop.csv | head -12
python3 -c "
import json; d=json.loads(open('$O/bench_lsop.json').read().strip().splitlines()[-1]); print('lsop enc', d['encode_ms'], 'dec', d['decode_ms'], 'value', d['value'], 'bit_exact', d['bit_exact'])"
