"""BASELINE config 1: one 120x150 int32 tile per call through the plug-in style entry points (host memory in, packing out),
the way Gridfour's CodecMaster calls a codec today -- latency of the GPU path against the CPU port on the same tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridfour_amd
import oracle
n_rows, n_cols = 120, 150
tile = oracle.dem_tiles(oracle.DEM_SEED + 2, n_rows, n_cols, 144, 0, 1)[0]
N = 300
for name, codec, enc, dec in (("CodecHuffman", gridfour_amd.CodecHuffmanHip(), oracle.codec_huffman_encode, oracle.codec_huffman_decode),
                              ("CodecCanonHuffman", gridfour_amd.CodecCanonHuffmanHip(), oracle.codec_canon_encode, oracle.codec_canon_decode)):
    pk = codec.encode(0, n_rows, n_cols, tile)
    assert pk == enc(0, n_rows, n_cols, tile)[0]
    for _ in range(20):
        codec.encode(0, n_rows, n_cols, tile); codec.decode(n_rows, n_cols, pk)
    t0 = time.perf_counter()
    for _ in range(N):
        codec.encode(0, n_rows, n_cols, tile)
    t1 = time.perf_counter()
    for _ in range(N):
        out = codec.decode(n_rows, n_cols, pk)
    t2 = time.perf_counter()
    assert np.array_equal(out, tile)
    c0 = time.perf_counter()
    for _ in range(N):
        enc(0, n_rows, n_cols, tile)
    c1 = time.perf_counter()
    for _ in range(N):
        dec(n_rows, n_cols, pk)
    c2 = time.perf_counter()
    print("%s one tile per call: GPU encode %.0f us, decode %.0f us (%.1f MB/s round trip); CPU port encode %.0f us, decode %.0f us (%.1f MB/s)" % (
        name, (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, tile.nbytes * N / (t2 - t0) / 1e6,
        (c1 - c0) / N * 1e6, (c2 - c1) / N * 1e6, tile.nbytes * N / (c2 - c0) / 1e6))
