#!/bin/bash
# Occupancy sensitivity of k_huffman_decode<fast>: one tile shape whose LDS footprint allows six workgroups per CU, then unused
# dynamic LDS takes them away one at a time (needs: python -m gridfour_amd.build --variant ldspad -DGF_DEC_LDS_PAD_ENV).
# usage: tools/occupancy_sweep.sh [nRows nCols nTiles]
NR=${1:-70}; NC=${2:-100}; NT=${3:-33000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GVRS_HIP_VARIANT=ldspad
for pad in 0 1280 5120 6400 12800 14080; do
  echo -n "pad $pad: "; GF_DEC_LDS_PAD=$pad python3 tools/shape_time.py $NR $NC $NT 2>&1 | tail -1
done
