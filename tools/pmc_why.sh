#!/bin/bash
# The "why is the fraction low" evidence of a round, for the kernels that SHIP: SQ instruction counts and wait / busy cycles per
# tile (rocprofv3 --pmc, separate passes, shipping library), and the s_memtime phase stamps of the diagnostic build.
# usage: tools/pmc_why.sh <outdir> [dec|enc|both]      (writes <outdir>/pmc_sq_{dec,enc}.txt, phase_cycles_{dec,enc,canon}.txt)
OUT=${1:-gpurun_out/why}; WHICH=${2:-both}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
P2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES"
P3="SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_INSTS_FLAT"
P4="GRBM_GUI_ACTIVE"
for w in dec enc; do
  if [ "$WHICH" = both ] || [ "$WHICH" = $w ]; then
    bash tools/pmc_sq.sh $w 0 0 "$P1" "$P2" "$P3" "$P4" > $OUT/pmc_sq_$w.txt 2>&1
    cat $OUT/pmc_sq_$w.txt
  fi
done
timeout 300 python3 tools/phase_cycles_dec.py > $OUT/phase_cycles_dec.txt 2>&1; cat $OUT/phase_cycles_dec.txt
timeout 300 python3 tools/phase_cycles.py > $OUT/phase_cycles_enc.txt 2>&1; tail -20 $OUT/phase_cycles_enc.txt
timeout 300 python3 tools/phase_cycles_canon.py > $OUT/phase_cycles_canon.txt 2>&1; tail -20 $OUT/phase_cycles_canon.txt
