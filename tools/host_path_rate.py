"""Throughput of the host-memory entry points (pageable host buffers in, packings out and back): the PCIe-inclusive
figure DESIGN.md quotes beside the device-resident one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gridfour_amd
import oracle
n_rows, n_cols, nt = 120, 150, 12960
vals = oracle.dem_tiles(oracle.DEM_SEED + 2, n_rows, n_cols, 144, 0, nt)
for name, cls in (("CodecHuffman", gridfour_amd.CodecHuffmanHip), ("CodecCanonHuffman", gridfour_amd.CodecCanonHuffmanHip)):
    codec = cls()
    codec.encode_batch(0, n_rows, n_cols, vals[:64])
    t0 = time.perf_counter()
    packs, preds, st = codec.encode_batch(0, n_rows, n_cols, vals)
    t1 = time.perf_counter()
    out, st2 = codec.decode_batch(n_rows, n_cols, packs)
    t2 = time.perf_counter()
    assert (st == 0).all() and (st2 == 0).all() and np.array_equal(out, vals)
    mb = vals.nbytes / 1e6
    print("%s host path (incl. H2D/D2H and Python list handling): encode %.0f MB/s, decode %.0f MB/s, round trip %.0f MB/s" % (
        name, mb / (t1 - t0), mb / (t2 - t1), mb / (t2 - t0)))
