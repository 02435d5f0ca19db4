#!/bin/bash
# Round 6's own artifacts beside tools/round_artifacts.sh (same <tag>): the LSOP12 byte planes (share of plane tiles, phase stamps of
# k_lsop_unpack2, A/B against a build without planes on one box, HBM counters of its kernels are in round_artifacts' pmc_hbm), the review's
# histogram aggregation A/B with its two LDS counters, one process enqueueing 2 / 8 shards, a soak of every codec.
# Experiment builds it expects (made on the build box, they travel with the snapshot):
#   python -m gridfour_amd.build --variant histagg -DGF_ENC_HIST_AGG ; python -m gridfour_amd.build --variant noplanes -DGF_LSOP_NO_PLANES
TAG=${1:-r06_vX}
O=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
for wl in etopo1 float256_lsop; do echo "== $wl"; timeout 600 python3 tools/lsop_plane_share.py $wl 2>&1 | grep "tiles\|roundtrip"; done > $O/lsop_plane_share.txt
timeout 600 python3 tools/phase_cycles_lsop.py > $O/phase_cycles_lsop.txt 2>&1
timeout 600 python3 tools/phase_cycles_lsop.py 256 256 4096 > $O/phase_cycles_lsop_256.txt 2>&1
if [ -f gridfour_amd/lib/libgvrs_hip_noplanes.so ]; then
  for v in "" noplanes; do
    echo "== ${v:-shipping}"
    for shape in "120 150 12960" "256 256 4096" "200 200 1024"; do GVRS_HIP_VARIANT=$v python3 tools/shape_time.py $shape lsop 2>&1 | tail -1; done
    GVRS_HIP_VARIANT=$v GF_DEM_STYLE=1 python3 tools/shape_time.py 120 150 12960 lsop 2>&1 | tail -1 | sed 's/^/rough: /'
  done > $O/lsop_planes_ab.txt 2>&1
  AB_LINES=8 bash tools/ab_kernels.sh "noplanes" $O 120 150 12960 lsop > $O/lsop_planes_ab_kernels.txt 2>&1
fi
if [ -f gridfour_amd/lib/libgvrs_hip_histagg.so ]; then
  P3="SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_INSTS_FLAT"
  P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
  rm -f $O/hist_agg_ab.txt
  for v in "" histagg; do
    echo "== ${v:-shipping}" >> $O/hist_agg_ab.txt
    GVRS_HIP_VARIANT=$v bash tools/pmc_sq.sh enc 0 0 "$P3" "$P1" >> $O/hist_agg_ab.txt 2>&1
  done
  AB_LINES=4 bash tools/ab_kernels.sh "histagg" $O >> $O/hist_agg_ab.txt 2>&1
  AB_LINES=4 bash tools/ab_kernels.sh "histagg" $O >> $O/hist_agg_ab.txt 2>&1
fi
for g in 2 8; do GF_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus $g --cpu-sample-tiles 0 2>/dev/null | tail -1 > $O/bench_single_process_${g}shards_one_gpu.json; done
timeout 900 python3 tools/soak.py 300 60061001 > $O/soak_60061001.txt 2>&1; tail -3 $O/soak_60061001.txt
rm -f $O/run_*.txt
