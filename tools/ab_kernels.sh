#!/bin/bash
# A/B on ONE box with per-kernel times: rocprofv3 kernel statistics of tools/shape_time.py for the shipping library and experiment builds.
# usage: tools/ab_kernels.sh "<variant> ..." <outdir> [nRows nCols nTiles [codec]]     ("" = the shipping library only)
V=$1; O=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
for v in "" $V; do
  n=${v:-shipping}
  rm -rf $O/p_$n
  GVRS_HIP_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$n -- python3 tools/shape_time.py ${1:-120} ${2:-150} ${3:-12960} ${4:-huffman} > $O/run_$n.txt 2>&1
  f=$(find $O/p_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n: $(tail -1 $O/run_$n.txt)"
  [ -n "$f" ] && python3 tools/kernel_times.py $f | grep -v "rocclr\|synth" | head -${AB_LINES:-8}
  [ -n "$f" ] && cp $f $O/kernel_stats_$n.csv
  rm -rf $O/p_$n
done
