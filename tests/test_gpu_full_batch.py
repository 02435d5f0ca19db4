"""The headline batch sizes of BASELINE.json through the C ABI on the GPU (not only bench.py): the whole ETOPO1-shaped batch
(configs[2]: 12,960 tiles of 120x150) and a GEBCO-shaped shard (configs[3]: 200x200 tiles) -- size-independent properties
(every tile survives the round trip, every status OK, sum of packing lengths = length of the compacted blob, compacted blob
decodes to the same cells) plus sampled byte parity with the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,nt,tpr,seed", [((120, 150), 12960, 144, 2), ((200, 200), 4096, 432, 3), ((200, 200), 11664, 432, 3)],
                         ids=["etopo1_12960x120x150", "gebco_4096x200x200", "gebco_shard_11664x200x200"])
@pytest.mark.parametrize("codec", ["huffman", "canon"])
def test_full_batch_properties(shape, nt, tpr, seed, codec):
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib
    from gridfour_amd._lib import check
    n_rows, n_cols = shape
    cells = n_rows * n_cols
    ctx = gridfour_amd.GvrsHipContext(0)
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(2 * cells + 1024 + 15) // 16 * 16, codec=codec)
    b.synth_dem(oracle.DEM_SEED + seed, tpr)
    b.encode()
    b.decode()
    ctx.synchronize()
    assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)                       # every one of the tiles
    lengths = b.get_lengths().astype(np.int64)
    preds = b.get_predictors()
    assert ((preds >= 1) & (preds <= 3)).all() and (lengths > 10).all() and (lengths < 4 * cells).all()
    # sampled byte parity (the generator itself is pinned by tests/test_gpu_parity.py::test_synth_dem*)
    enc = oracle.codec_canon_encode if codec == "canon" else oracle.codec_huffman_encode
    for t in list(range(0, nt, nt // 24))[:24] + [nt - 1]:
        ref, used = enc(0, n_rows, n_cols, vals[t])
        assert used == preds[t] and b.get_packing(t, int(lengths[t])) == ref, t
    # compaction: exclusive scan of the lengths, then the compact blob decodes to the same cells
    total = int(lengths.sum())
    d_off, d_blob = DeviceBuffer(ctx, (nt + 1) * 8), DeviceBuffer(ctx, total + 64)
    check(lib().gf_compact_dev(ctx.handle, None, nt, b.slots.ptr, b.stride, b.lengths.ptr, d_off.ptr, d_blob.ptr, total + 64), "compact")
    ctx.synchronize()
    off = d_off.download(np.uint64, nt + 1)
    assert int(off[nt]) == total and np.array_equal(np.diff(off.astype(np.int64)), lengths)
    b.decoded.fill(0)
    fn = lib().gf_canon_decode_batch_i32_dev if codec == "canon" else lib().gf_huffman_decode_batch_i32_dev
    check(fn(ctx.handle, None, n_rows, n_cols, nt, d_blob.ptr, total + 64, d_off.ptr, 0, b.lengths.ptr, b.decoded.ptr, b.dec_status.ptr),
          "decode of the compact blob")
    ctx.synchronize()
    assert (b.get_dec_status() == 0).all() and np.array_equal(b.get_decoded(), vals)


@pytest.mark.parametrize("codec", ["huffman", "canon", "lsop"])
def test_more_than_2_20_tiles(codec):
    """More tiles than one grid row holds (gf_tile_grid: 2^20 workgroups in x, the rest in y): every tile survives the round
    trip, the tiles either side of the x/y seam and the last one match the oracle byte for byte."""
    import gridfour_amd
    from gridfour_amd import DeviceTileBatch
    n_rows, n_cols = (8, 9) if codec == "lsop" else (4, 5)             # LSOP12 declines tiles below 6 x 6
    nt = (1 << 20) + 4099
    cells = n_rows * n_cols
    ctx = gridfour_amd.GvrsHipContext(0)
    b = DeviceTileBatch(ctx, n_rows, n_cols, nt, slot_stride=(8 * cells + 1024 + 15) // 16 * 16, codec=codec)
    b.synth_dem(oracle.DEM_SEED + 11, 1 << 10)
    b.encode()
    b.decode()
    ctx.synchronize()
    es, ds = b.get_enc_status(), b.get_dec_status()
    ok = es == 0
    assert ((es == 0) | (es == 1)).all() and ok.sum() > nt // 2            # tiny tiles: some packings are declined (status 1)
    assert (ds[ok] == 0).all()
    vals, dec = b.get_values(), b.get_decoded()
    assert np.array_equal(dec[ok], vals[ok])
    lengths = b.get_lengths()
    enc = {"huffman": lambda v: oracle.codec_huffman_encode(0, n_rows, n_cols, v)[0],
           "canon": lambda v: oracle.codec_canon_encode(0, n_rows, n_cols, v)[0],
           "lsop": lambda v: oracle.lsop12_encode(0, n_rows, n_cols, v, False)[0]}[codec]
    for t in [0, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, nt - 1]:
        ref = enc(vals[t])
        if ref is None:
            assert es[t] == 1, t
        else:
            assert es[t] == 0 and b.get_packing(t, int(lengths[t])) == ref, t


def test_more_than_2_20_tiles_float_planes():
    """The CodecFloat plane stage on more tiles than one grid row holds: planes of the tiles either side of the seam and of the
    last tile equal the oracle's, every tile comes back bit for bit."""
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    from gridfour_amd._lib import check
    n_rows, n_cols, nt = 3, 5, (1 << 20) + 777
    cells = n_rows * n_cols
    ctx = gridfour_amd.GvrsHipContext(0)
    rng = np.random.default_rng(5)
    vals = (rng.standard_normal((nt, cells)) * 1000).astype(np.float32)
    vals[rng.random((nt, cells)) < 0.01] = np.nan
    pstride = int(lib().gf_float_planes_bytes(n_rows, n_cols))
    d_in, d_planes, d_out = DeviceBuffer(ctx, nt * cells * 4), DeviceBuffer(ctx, nt * pstride), DeviceBuffer(ctx, nt * cells * 4)
    d_in.upload(vals)
    d_out.fill(0)
    check(lib().gf_float_planes_encode_dev(ctx.handle, None, n_rows, n_cols, nt, d_in.ptr, d_planes.ptr, pstride), "enc")
    check(lib().gf_float_planes_decode_dev(ctx.handle, None, n_rows, n_cols, nt, d_planes.ptr, pstride, d_out.ptr), "dec")
    ctx.synchronize()
    back = d_out.download(np.uint32, nt * cells).reshape(nt, cells)
    assert np.array_equal(back, vals.view(np.uint32))
    planes = d_planes.download(np.uint8, nt * pstride).reshape(nt, pstride)
    for t in [0, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, nt - 1]:
        assert planes[t].tobytes() == bytes(oracle.float_planes_encode(n_rows, n_cols, vals[t].view(np.uint32))), t


def test_whole_gebco_grid_through_one_context():
    """BASELINE configs[3] at its full size on ONE GPU: the 93,312 tiles of the GEBCO_2023-shaped grid (14.9 GB of cells),
    generated on the device in pieces, through gf_huffman_{encode,decode}_batch_i32 on host memory -- chunked, pipelined
    staging inside the library, device memory bounded by the chunk.  Every status OK, every tile back bit for bit, the
    offsets consistent, sampled packings equal to the oracle's (what tools/host_path_rate.py gebco_full times)."""
    import gridfour_amd
    from gridfour_amd import DeviceTileBatch, lib
    from gridfour_amd.sharding import _ptr
    n_rows, n_cols, nt, tpr = 200, 200, 93312, 432
    cells = n_rows * n_cols
    # 34 GB of host arrays (cells in, cells back, packings): skipped where the box does not have them to give
    try:
        avail = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo")}.get("MemAvailable", 0) * 1024
    except OSError:
        avail = 0
    if avail and avail < 40 * 2 ** 30:
        pytest.skip("needs about 40 GB of host memory (%.0f GB available)" % (avail / 2 ** 30))
    ctx = gridfour_amd.GvrsHipContext(0)
    vals = np.empty((nt, cells), np.int32)
    piece = 7776
    for t0 in range(0, nt, piece):
        n = min(piece, nt - t0)
        b = DeviceTileBatch(ctx, n_rows, n_cols, n)
        b.synth_dem(oracle.DEM_SEED + 3, tpr, tile0=t0)
        ctx.synchronize()
        vals[t0:t0 + n] = b.get_values()
        b.free()
    cap = nt * cells                                                      # a byte per cell: DEM packings are ~0.55
    blob = np.empty(cap, np.uint8)
    off = np.zeros(nt + 1, np.uint64)
    pred = np.zeros(nt, np.uint8)
    st = np.zeros(nt, np.int32)
    assert lib().gf_huffman_encode_batch_i32(ctx.handle, 0, n_rows, n_cols, nt, _ptr(vals), _ptr(blob), cap, _ptr(off), _ptr(pred), _ptr(st)) == 0
    assert (st == 0).all() and ((pred >= 1) & (pred <= 3)).all()
    lengths = np.diff(off.astype(np.int64))
    assert off[0] == 0 and (lengths > 10).all() and (lengths < 4 * cells).all()
    for t in [0, 1, nt // 3, nt // 2, nt - 2, nt - 1]:
        ref, used = oracle.codec_huffman_encode(0, n_rows, n_cols, vals[t])
        assert used == pred[t] and blob[int(off[t]):int(off[t + 1])].tobytes() == ref, t
    out = np.empty_like(vals)
    assert lib().gf_huffman_decode_batch_i32(ctx.handle, n_rows, n_cols, nt, _ptr(blob), _ptr(off), _ptr(out), _ptr(st)) == 0
    assert (st == 0).all()
    for t0 in range(0, nt, piece):                                        # compared in pieces: no third 15 GB array
        assert np.array_equal(out[t0:t0 + piece], vals[t0:t0 + piece]), t0
