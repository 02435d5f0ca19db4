#!/bin/bash
# A/B on ONE box (devices differ by several per cent): decode / encode time of the shipping library against experiment builds.
# usage: tools/ab.sh "<variant> ..." [nRows nCols nTiles [codec]]     ("" = the shipping library)
V=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" $V; do
  echo -n "[${v:-shipping}] "; GVRS_HIP_VARIANT=$v python3 tools/shape_time.py ${1:-120} ${2:-150} ${3:-12960} ${4:-huffman} 2>&1 | tail -1
done
done
