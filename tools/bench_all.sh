#!/bin/bash
# One line per bench configuration (value, ms per step, encode ms, decode ms, bit-exact, roofline fraction): the quick look
# after a kernel change.  usage (on the GPU box): [GVRS_HIP_VARIANT=name] bash tools/bench_all.sh [codecs...]
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('encode_ms'), d.get('decode_ms'), d['bit_exact'], d['roofline']['frac'])"; }
WHAT=${*:-huffman canon lsop dem1024 gebco_shard float256 float256_lsop kernels}
for w in $WHAT; do
  case $w in
    huffman) python3 bench.py --cpu-sample-tiles 0 2>/dev/null | tail -1 | p huffman ;;
    canon|lsop) python3 bench.py --codec $w --cpu-sample-tiles 0 2>/dev/null | tail -1 | p $w ;;
    dem1024|gebco_shard) python3 bench.py --workload $w --cpu-sample-tiles 0 2>/dev/null | tail -1 | p $w ;;
    float256) python3 bench.py --codec float --workload float256 --cpu-sample-tiles 0 2>/dev/null | tail -1 | p float256 ;;
    float256_lsop) python3 bench.py --codec lsop --workload float256_lsop --cpu-sample-tiles 0 2>/dev/null | tail -1 | p float256_lsop ;;
    kernels) python3 tools/lsop_recon_time.py 2>/dev/null | tail -1 ;;
  esac
done
