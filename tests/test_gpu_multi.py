"""Several contexts in one process (gf_multi_*) and the pipelined host-memory path, through the C ABI on the GPU.

The box has one GPU, so the shards of a gf_multi are contexts on device 0 (a device may be listed more than once): the
partition, the per-context host threads and the concatenation by exclusive scan are exactly what runs on a node with eight.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from tilegen import NULL, make_tile

pytestmark = pytest.mark.gpu


def _batch(n_rows, n_cols, nt, seed=5):
    tiles = oracle.dem_tiles(oracle.DEM_SEED + seed, n_rows, n_cols, 16, 0, nt).copy()
    return tiles


def _single(codec, n_rows, n_cols, tiles):
    import gridfour_amd
    from gridfour_amd import lib
    from gridfour_amd.sharding import _ptr
    nt = tiles.shape[0]
    cap = nt * int(lib().gf_huffman_default_stride(n_rows, n_cols))
    blob = np.empty(cap, np.uint8)
    off = np.zeros(nt + 1, np.uint64)
    pred = np.zeros(nt, np.uint8)
    st = np.zeros(nt, np.int32)
    ctx = gridfour_amd.GvrsHipContext(0)
    rc = getattr(lib(), "gf_%s_encode_batch_i32" % codec)(ctx.handle, 1, n_rows, n_cols, nt, _ptr(tiles), _ptr(blob), cap, _ptr(off),
                                                          _ptr(pred), _ptr(st))
    assert rc == 0, rc
    return blob[:int(off[nt])].copy(), off, pred, st


@pytest.mark.parametrize("codec", ["huffman", "canon"])
@pytest.mark.parametrize("shards,nt", [(2, 64), (3, 7), (4, 3), (2, 1)])
def test_multi_equals_single_context(codec, shards, nt):
    import gridfour_amd
    n_rows, n_cols = 40, 50
    tiles = _batch(n_rows, n_cols, nt)
    if nt > 4:
        tiles[2, 5:90] = NULL                       # a tile for the nulls predictor
        tiles[4, :] = NULL                          # a declined tile (Java null): no bytes in the blob
    blob1, off1, pred1, st1 = _single(codec, n_rows, n_cols, tiles)
    m = gridfour_amd.GvrsHipMulti([0] * shards)
    assert len(m) == shards
    blob, off, pred, st = m.encode_batch(1, n_rows, n_cols, tiles, codec=codec)
    assert np.array_equal(off, off1) and np.array_equal(pred, pred1) and np.array_equal(st, st1)
    assert blob.tobytes() == blob1.tobytes()
    vals, dst = m.decode_batch(n_rows, n_cols, blob, off, codec=codec)
    for t in range(nt):
        if st[t] == 0:
            assert dst[t] == 0 and np.array_equal(vals[t], tiles[t]), t
        else:
            assert st[t] == 1                      # declined: nothing to decode (zero-length packing)
    m.close()


def test_multi_dev_shards():
    """device-resident shards: one call enqueues every shard on its context's stream"""
    import gridfour_amd
    from gridfour_amd import DeviceTileBatch, lib
    n_rows, n_cols, per = 60, 70, 40
    m = gridfour_amd.GvrsHipMulti([0, 0])
    batches = []
    for g in range(2):
        ctx = gridfour_amd.GvrsHipContext(0)
        b = DeviceTileBatch(ctx, n_rows, n_cols, per)
        b.synth_dem(oracle.DEM_SEED + 9, 8, tile0=g * per)
        ctx.synchronize()
        batches.append(b)
    G = 2
    arr = lambda ptrs: (C.c_void_p * G)(*ptrs)
    nT = (C.c_size_t * G)(per, per)
    rc = lib().gf_huffman_encode_batch_i32_multi_dev(m.handle, 0, n_rows, n_cols, nT, arr([b.values.ptr for b in batches]),
                                                     arr([b.slots.ptr for b in batches]), batches[0].stride,
                                                     arr([b.lengths.ptr for b in batches]), arr([b.predictors.ptr for b in batches]),
                                                     arr([b.enc_status.ptr for b in batches]), 0xF)
    assert rc == 0
    m.synchronize()
    blobBytes = (C.c_size_t * G)(*[per * b.stride for b in batches])
    rc = lib().gf_huffman_decode_batch_i32_multi_dev(m.handle, n_rows, n_cols, nT, arr([b.slots.ptr for b in batches]), blobBytes, None,
                                                     batches[0].stride, arr([b.lengths.ptr for b in batches]),
                                                     arr([b.decoded.ptr for b in batches]), arr([b.dec_status.ptr for b in batches]))
    assert rc == 0
    m.synchronize()
    for g, b in enumerate(batches):
        want = oracle.dem_tiles(oracle.DEM_SEED + 9, n_rows, n_cols, 8, g * per, per)
        assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
        assert np.array_equal(b.get_decoded(), want)
        ref, _ = oracle.codec_huffman_encode(0, n_rows, n_cols, want[3])
        assert b.get_packing(3) == ref


def test_host_pipeline_many_chunks_overflow_and_pinned():
    """a batch of several staging chunks, with a declined tile, a nulls tile and an incompressible tile (longer than the
    default slot: the overflow path) in different chunks; then the same through page-locked memory"""
    import gridfour_amd
    n_rows, n_cols = 120, 150
    nt = 2100                                       # chunk = 64 MB / 72 KB = 932 tiles -> 3 chunks
    tiles = _batch(n_rows, n_cols, nt, seed=6)
    rng = np.random.default_rng(3)
    tiles[1000] = rng.integers(-2 ** 31, 2 ** 31 - 1, n_rows * n_cols, dtype=np.int64).astype(np.int32)   # incompressible
    tiles[40, 100:300] = NULL
    tiles[1999, :] = NULL
    codec = gridfour_amd.CodecHuffmanHip()
    packs, preds, st = codec.encode_batch(0, n_rows, n_cols, tiles)
    assert st[1999] == 1 and packs[1999] is None
    assert st[1000] == 0 and len(packs[1000]) > 4 * n_rows * n_cols
    for t in (0, 40, 931, 932, 1000, 1001, 1863, 1864, 2099):
        ref, used = oracle.codec_huffman_encode(0, n_rows, n_cols, tiles[t])
        assert packs[t] == ref and preds[t] == used, t
    good = [t for t in range(nt) if packs[t] is not None]
    vals, dst = codec.decode_batch(n_rows, n_cols, [packs[t] for t in good])
    assert (dst == 0).all() and np.array_equal(vals, tiles[good])
    # page-locked input and output: same bytes
    from gridfour_amd import lib, PinnedArray
    from gridfour_amd.sharding import _ptr
    pin = PinnedArray((nt, n_rows * n_cols), np.int32)
    pin.array[:] = tiles
    cap = sum(len(p) for p in packs if p) + 64
    pblob = PinnedArray(cap, np.uint8)
    off = np.zeros(nt + 1, np.uint64)
    st2 = np.zeros(nt, np.int32)
    rc = lib().gf_huffman_encode_batch_i32(codec.ctx.handle, 0, n_rows, n_cols, nt, _ptr(pin.array), _ptr(pblob.array), cap, _ptr(off),
                                           None, _ptr(st2))
    assert rc == 0 and np.array_equal(st2, st)
    assert pblob.array[:int(off[nt])].tobytes() == b"".join(p for p in packs if p)
    out = PinnedArray((nt, n_rows * n_cols), np.int32)
    out.array[:] = 0
    rc = lib().gf_huffman_decode_batch_i32(codec.ctx.handle, n_rows, n_cols, nt, _ptr(pblob.array), _ptr(off), _ptr(out.array), _ptr(st2))
    assert rc == 0
    ok = st != 1
    assert (st2[ok] == 0).all() and np.array_equal(out.array[ok], tiles[ok])
    # too small a blob: GF_ERR_CAPACITY, offsets still complete
    off3 = np.zeros(nt + 1, np.uint64)
    small = np.empty(1000, np.uint8)
    rc = lib().gf_huffman_encode_batch_i32(codec.ctx.handle, 0, n_rows, n_cols, nt, _ptr(tiles), _ptr(small), 1000, _ptr(off3), None, None)
    assert rc == -3 and np.array_equal(off3, off)


@pytest.mark.parametrize("shards,nt", [(2, 9), (3, 5)])
def test_multi_deflate_lsop_float_equal_single_context(shards, nt):
    """gf_{deflate,lsop12}_*_batch_i32_multi and gf_float_*_batch_f32_multi (BASELINE config 5 sharded inside the library):
    the bytes of a batch cut over several contexts are those of one context, and decode back to the cells."""
    import gridfour_amd
    from gridfour_amd import lib
    from gridfour_amd.sharding import _ptr
    n_rows, n_cols = 24, 40
    tiles = _batch(n_rows, n_cols, nt, seed=7)
    m = gridfour_amd.GvrsHipMulti([0] * shards)
    ctx = gridfour_amd.GvrsHipContext(0)
    cells = n_rows * n_cols

    # CodecDeflate
    blob1, off1, pred1, st1 = _single("deflate", n_rows, n_cols, tiles)
    blob, off, pred, st = m.encode_batch(1, n_rows, n_cols, tiles, codec="deflate")
    assert np.array_equal(off, off1) and np.array_equal(pred, pred1) and np.array_equal(st, st1) and blob.tobytes() == blob1.tobytes()
    vals, dst = m.decode_batch(n_rows, n_cols, blob, off, codec="deflate")
    assert (dst == 0).all() and np.array_equal(vals, tiles)

    # LSOP12, the reference's default (Deflate alternative on) and without it
    for deflate in (True, False):
        cap = nt * (int(lib().gf_lsop12_max_packing(n_rows, n_cols)) + 64)
        b1, o1 = np.empty(cap, np.uint8), np.zeros(nt + 1, np.uint64)
        ty1, s1 = np.zeros(nt, np.uint8), np.zeros(nt, np.int32)
        rc = lib().gf_lsop12_encode_batch_i32(ctx.handle, 2, n_rows, n_cols, nt, _ptr(tiles), 1 if deflate else 0, _ptr(b1), cap, _ptr(o1),
                                              _ptr(ty1), _ptr(s1))
        assert rc == 0, rc
        blob, off, ty, st = m.encode_batch(2, n_rows, n_cols, tiles, codec="lsop12", deflate_enabled=deflate)
        assert np.array_equal(off, o1) and np.array_equal(ty, ty1) and np.array_equal(st, s1)
        assert blob.tobytes() == b1[:int(o1[nt])].tobytes()
        vals, dst = m.decode_batch(n_rows, n_cols, blob, off, codec="lsop12")
        assert (dst == 0).all() and np.array_equal(vals, tiles)

    # CodecFloat
    f = (tiles.astype(np.float32) * np.float32(0.1)).reshape(nt, cells)
    flt = gridfour_amd.CodecFloatHip(context=ctx, level=6)
    packs = flt.encode_floats_batch(3, n_rows, n_cols, f)
    blob, off, _, _ = m.encode_batch(3, n_rows, n_cols, f, codec="float", level=6)
    assert [bytes(blob[int(off[t]):int(off[t + 1])]) for t in range(nt)] == packs
    vals, dst = m.decode_batch(n_rows, n_cols, blob, off, codec="float")
    assert (dst == 0).all() and np.array_equal(vals.view(np.uint32), f.view(np.uint32))
    m.close()
