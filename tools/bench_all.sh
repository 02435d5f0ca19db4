p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('encode_ms'), d.get('decode_ms'), d['bit_exact'], d['roofline']['frac'])"; }
python bench.py --cpu-sample-tiles 0 2>/dev/null | tail -1 | p huffman
python bench.py --codec canon --cpu-sample-tiles 0 2>/dev/null | tail -1 | p canon
python bench.py --codec lsop --cpu-sample-tiles 0 2>/dev/null | tail -1 | p lsop
python bench.py --workload dem1024 --cpu-sample-tiles 0 2>/dev/null | tail -1 | p dem1024
python bench.py --workload gebco_shard --cpu-sample-tiles 0 2>/dev/null | tail -1 | p gebco_shard
python bench.py --codec float --workload float256 --cpu-sample-tiles 0 2>/dev/null | tail -1 | p float256
python bench.py --codec lsop --workload float256_lsop --cpu-sample-tiles 0 2>/dev/null | tail -1 | p float256_lsop
python tools/lsop_recon_time.py 2>/dev/null | tail -1
