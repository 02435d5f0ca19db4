cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
P3="SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_INSTS_FLAT"
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
for v in "" histagg; do
  echo "== ${v:-shipping}" >> $O/hist_agg_ab.txt
  GVRS_HIP_VARIANT=$v bash tools/pmc_sq.sh enc 0 0 "$P3" "$P1" >> $O/hist_agg_ab.txt 2>&1
done
AB_LINES=4 bash tools/ab_kernels.sh "histagg" $O >> $O/hist_agg_ab.txt 2>&1
AB_LINES=4 bash tools/ab_kernels.sh "histagg" $O >> $O/hist_agg_ab.txt 2>&1
cat $O/hist_agg_ab.txt
for g in 2 8; do GF_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus $g --cpu-sample-tiles 0 2>/dev/null | tail -1 > $O/bench_single_process_${g}shards_one_gpu.json; python3 -c "
import json; d=json.load(open('$O/bench_single_process_${g}shards_one_gpu.json')); print($g, 'shards: value', d['value'], 'ms_per_step', d['ms_per_step'], 'host_enqueue_ms_per_step', d['host_enqueue_ms_per_step'], 'bit_exact', d['bit_exact'])"; done
