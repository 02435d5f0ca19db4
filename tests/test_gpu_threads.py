"""One codec instance under the call patterns of the reference's tile cache.

* The reference calls ONE decoder instance from two threads (gvrs/RasterTileCache.java:418-421 hands tiles to
  TileDecompressionAssistant.java:68-73, whose worker decodes with the same CodecMaster the caller's thread uses): the
  context's lock (include/gvrs_hip_codec.h, "One context per ...") makes such calls run one after the other.
* The one-tile calls replay recorded graphs that hold addresses of the context's buffers; a batch that makes the context grow
  those buffers between two replays must not leave the graphs pointing at freed memory (round-4 review).
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tiles(seed, n_rows, n_cols, n, rough=False):
    import oracle
    return oracle.dem_tiles(oracle.DEM_SEED + seed, n_rows, n_cols, 64, 0, n)


@pytest.mark.parametrize("cls", ["CodecHuffmanHip", "CodecCanonHuffmanHip"])
def test_one_codec_instance_decodes_from_two_threads(cls):
    import gridfour_amd
    codec = getattr(gridfour_amd, cls)(device=0)
    n_rows, n_cols, n = 120, 150, 48
    tiles = _tiles(11, n_rows, n_cols, n)
    packs, _, status = codec.encode_batch(0, n_rows, n_cols, tiles)
    assert (status == 0).all()
    other = _tiles(12, 60, 80, n)
    packs2, _, status2 = codec.encode_batch(0, 60, 80, other)
    assert (status2 == 0).all()
    errors = []
    start = threading.Barrier(3)

    def one_by_one(shape, pk, want, rounds):
        # the assistant's worker: ICompressionDecoder.decode, a tile per call
        try:
            start.wait()
            for r in range(rounds):
                for t in range(n):
                    got = codec.decode(shape[0], shape[1], pk[t])
                    if not np.array_equal(np.asarray(got).reshape(-1), want[t].reshape(-1)):
                        errors.append(("single", shape, r, t))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def batches(rounds):
        # the caller's own thread: batches of the same instance in between
        try:
            start.wait()
            for r in range(rounds):
                vals, st = codec.decode_batch(n_rows, n_cols, packs)
                if not ((st == 0).all() and np.array_equal(vals, tiles)):
                    errors.append(("batch", r))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=one_by_one, args=((n_rows, n_cols), packs, tiles, 3)),
          threading.Thread(target=one_by_one, args=((60, 80), packs2, other, 3)),
          threading.Thread(target=batches, args=(12,))]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th), "a thread hangs on the context's lock"
    assert not errors, errors[:3]


def test_one_codec_instance_encodes_and_decodes_from_two_threads():
    import gridfour_amd
    import oracle
    codec = gridfour_amd.CodecHuffmanHip(device=0)
    n_rows, n_cols, n = 90, 120, 24
    tiles = _tiles(21, n_rows, n_cols, n)
    ref = [oracle.codec_huffman_encode(0, n_rows, n_cols, tiles[t])[0] for t in range(n)]
    errors = []

    def enc():
        try:
            for r in range(3):
                for t in range(n):
                    if codec.encode(0, n_rows, n_cols, tiles[t]) != ref[t]:
                        errors.append(("encode", r, t))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def dec():
        try:
            for r in range(3):
                for t in range(n):
                    got = codec.decode(n_rows, n_cols, ref[t])
                    if not np.array_equal(np.asarray(got).reshape(-1), tiles[t].reshape(-1)):
                        errors.append(("decode", r, t))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=enc), threading.Thread(target=dec)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th)
    assert not errors, errors[:3]


@pytest.mark.parametrize("cls", ["CodecHuffmanHip", "CodecCanonHuffmanHip"])
def test_one_tile_graphs_survive_a_batch_that_grows_the_context(cls):
    """single x3 (the graph is recorded and replayed), a batch of some thousand tiles (the context's tree / selection records and
    workspace grow: hipFree + hipMalloc), single again -- and once more in the other order of directions."""
    import gridfour_amd
    import oracle
    codec = getattr(gridfour_amd, cls)(device=0)          # a fresh context: nothing reserved
    enc_ref = oracle.codec_huffman_encode if cls == "CodecHuffmanHip" else oracle.codec_canon_encode
    n_rows, n_cols = 120, 150
    few = _tiles(31, n_rows, n_cols, 4)
    ref = [enc_ref(0, n_rows, n_cols, few[t])[0] for t in range(4)]
    for _ in range(3):
        for t in range(4):
            assert codec.encode(0, n_rows, n_cols, few[t]) == ref[t]
            assert np.array_equal(np.asarray(codec.decode(n_rows, n_cols, ref[t])).reshape(-1), few[t].reshape(-1))
    for n_big in (700, 3000):
        big = _tiles(32, n_rows, n_cols, n_big)
        packs, _, status = codec.encode_batch(0, n_rows, n_cols, big)
        assert (status == 0).all()
        vals, st = codec.decode_batch(n_rows, n_cols, packs)
        assert (st == 0).all() and np.array_equal(vals, big)
        # a different allocation pattern in between, so that the freed ranges are handed out again
        scratch = [gridfour_amd.DeviceBuffer(codec.ctx, 3 << 20) for _ in range(8)]
        for s in scratch:
            s.fill(0xA5)
        for _ in range(2):
            for t in range(4):
                assert codec.encode(0, n_rows, n_cols, few[t]) == ref[t], (n_big, t)
                assert np.array_equal(np.asarray(codec.decode(n_rows, n_cols, ref[t])).reshape(-1), few[t].reshape(-1)), (n_big, t)
        for s in scratch:
            s.free()
        # ... and the batch still works after the graphs were recorded again
        vals, st = codec.decode_batch(n_rows, n_cols, packs[:64])
        assert (st == 0).all() and np.array_equal(vals, big[:64])
