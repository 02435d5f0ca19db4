"""The byte plane between k_lsop_unpack2 and k_lsop_reconstruct_plane (round 6): tiles whose residuals are all bytes leave the
entropy stage as a plane in the reconstruction's pipeline order (two tiles to a wave, a byte per cell of rows 2.., the rows' four
initialisers in the plane's holes); any other tile keeps the int32 residual array and the old kernels.  Against the CPU oracle
(lsop/LsDecoder12.java:107-383 restated): shapes on both sides of every condition of gf_lsop_plane_geom (columns, the plane's room in
the residual slot, the stage's capacity), batches that mix the two kinds of tile in one wave, odd tile counts, a wide value in each
of the places a row keeps one, damaged containers."""
import numpy as np
import pytest

import oracle
from tilegen import make_tile

pytestmark = pytest.mark.gpu


def _terrain(nr, nc, seed, amp):
    rng = np.random.default_rng(seed * 7919 + nr * 131 + nc)
    y, x = np.mgrid[0:nr, 0:nc]
    return (amp * np.sin(x / 3.0 + seed) * np.cos(y / 2.5) + 10 * np.sin(x * y / 30.0) + rng.integers(-3, 4, (nr, nc))).astype(np.int32).ravel()


def _smooth(nr, nc, seed):
    """Terrain whose LSOP12 residuals are bytes throughout (the bench's kind of tile)."""
    rng = np.random.default_rng(seed + 17 * nr + nc)
    y, x = np.mgrid[0:nr, 0:nc]
    return (400 * np.sin(x / 17.0 + seed) * np.cos(y / 13.0) + 150 * np.sin((x + 2 * y) / 29.0) + rng.integers(-2, 3, (nr, nc))).astype(np.int32).ravel()


@pytest.fixture(scope="module")
def codec():
    import gridfour_amd
    return gridfour_amd.LsCodecHip(deflate_enabled=False)


def _check(codec, nr, nc, tiles, codec_index=1):
    tiles = np.asarray(tiles, np.int32).reshape(len(tiles), -1)
    slots, ln = oracle.batch_lsop12_encode(codec_index, nr, nc, tiles, deflate_enabled=False)
    keep = np.nonzero(ln > 0)[0]
    assert keep.size >= max(1, len(tiles) * 2 // 3), "the oracle declined most of the tiles"
    packs = [bytes(slots[t, :ln[t]]) for t in keep]
    vals, st = codec.decode_batch(nr, nc, packs)
    n_same = 0
    for k in range(len(packs)):
        if st[k] != 0:
            with pytest.raises(IOError):                  # (what the reference does not decode either: see test_gpu_lsop_head.py)
                oracle.lsop12_decode(nr, nc, packs[k])
        else:
            assert np.array_equal(vals[k], oracle.lsop12_decode(nr, nc, packs[k])), (nr, nc, k)
            n_same += int(np.array_equal(vals[k], tiles[keep[k]]))
    assert n_same >= len(packs) * 9 // 10
    return packs


# nC < 32: never a plane; 9x40, 20x36: the plane does not fit the residual slot; 256x256: the stage does not hold the stream; the others
# take the plane path with 1 .. 5 rows per lane, periods of 112 (narrow tiles), nC rounded up, exactly nC (160, 208); 256x256 and
# 1000x40 in several windows
SHAPES = [(40, 31), (9, 40), (20, 36), (34, 32), (35, 33), (66, 47), (67, 112), (40, 113), (98, 160), (120, 150), (131, 64), (50, 208),
          (34, 250), (256, 256), (1000, 40), (1030, 40)]    # (the last two: 32 rows per lane and 3,992 initialisers in the table; more than 1,024 rows: no plane)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d" % s)
def test_shapes_on_both_sides_of_the_plane_conditions(codec, shape):
    nr, nc = shape
    n = 5 if nr * nc > 30000 else 9                       # (odd: the last wave of the reconstruction holds one tile)
    tiles = [_smooth(nr, nc, s) for s in range(n)]
    _check(codec, nr, nc, tiles)


def test_plane_tiles_and_int32_tiles_share_waves(codec):
    """Smooth terrain (plane) and tiles with wide residuals (int32 array) alternate, so that every pairing occurs in a wave of
    k_lsop_reconstruct_plane: plane | plane, plane | other, other | plane, other | other."""
    nr, nc = 60, 90
    kinds = []
    pattern = "ppoppooopopppoo"
    for i, ch in enumerate(pattern * 3):
        if ch == "p":
            kinds.append(_smooth(nr, nc, i))
        else:
            kinds.append([make_tile("noise16", nr, nc, seed=i), make_tile("sparse_big", nr, nc, seed=i), _terrain(nr, nc, i, 40000),
                          make_tile("steps", nr, nc, seed=i)][i % 4])
    _check(codec, nr, nc, kinds)
    for n in (1, 2, 3):
        _check(codec, nr, nc, kinds[:n])
        _check(codec, nr, nc, kinds[2:2 + n])


@pytest.mark.parametrize("where", ["column0", "column1", "tail0", "tail1", "interior_first", "interior_last", "row1", "seed"])
def test_one_wide_value_in_each_place_of_a_row(codec, where):
    """A tile of bytes throughout but for ONE cell: in column 0 / 1 or one of the two tail columns (the initialisers the plane carries in
    its holes), in the first / last interior cell of a row, in row 1, or a huge seed (rows 0 and 1 stay int32: still a plane tile)."""
    nr, nc = 70, 100
    tiles = []
    for s in range(6):
        v = _smooth(nr, nc, s).reshape(nr, nc).copy()
        r = 2 + 11 * s                                     # rows of the first, second and third period of their lane
        if where == "column0":
            v[r:, :] += 5000                               # the cell above + 5000: only the column's difference is wide
        elif where == "column1":
            v[r, 1:] += 700
        elif where == "tail0":
            v[r, nc - 2] += 900
        elif where == "tail1":
            v[r, nc - 1] -= 1300
        elif where == "interior_first":
            v[r, 2] += 800
        elif where == "interior_last":
            v[r, nc - 3] -= 800
        elif where == "row1":
            v[1, 5 + s] += 100000
        else:
            v += 1 << 29
        tiles.append(v.ravel())
    tiles.append(_smooth(nr, nc, 99))
    _check(codec, nr, nc, tiles)


def test_large_batch_behind_the_lane_per_tile_prepasses(codec):
    """More than 4,096 tiles: k_lsop_head walks the first stream, k_lsop_unpack2 starts at the second one and writes the planes."""
    nr, nc = 34, 40
    base = [_smooth(nr, nc, s) for s in range(24)] + [make_tile("noise8", nr, nc, seed=1), make_tile("noise16", nr, nc, seed=2), _terrain(nr, nc, 3, 3000)]
    rng = np.random.default_rng(11)
    tiles = np.empty((4501, nr * nc), np.int32)
    for t in range(len(tiles)):
        v = base[t % len(base)].copy()
        v[int(rng.integers(0, v.size))] += int(rng.integers(-4, 5))
        tiles[t] = v
    _check(codec, nr, nc, tiles)


def test_damaged_containers_of_a_plane_shape_match_the_oracle(codec):
    nr, nc = 40, 64
    tiles = [_smooth(nr, nc, s) for s in range(120)]
    slots, ln = oracle.batch_lsop12_encode(0, nr, nc, np.asarray(tiles), deflate_enabled=False)
    packs = [bytes(slots[t, :ln[t]]) for t in range(len(tiles)) if ln[t] > 0]
    rng = np.random.default_rng(5)
    damaged = set()
    for k in rng.choice(len(packs), len(packs) // 2, replace=False):
        b = bytearray(packs[k])
        how = int(rng.integers(0, 4))
        if how == 0:
            b = b[:int(rng.integers(3, len(b)))]
        else:
            lo = 0 if how == 1 else 55
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(lo, len(b)))] ^= 1 << int(rng.integers(0, 8))
        packs[k] = bytes(b)
        damaged.add(int(k))
    vals, st = codec.decode_batch(nr, nc, packs)
    n_ok = 0
    for k in range(len(packs)):
        try:
            ref = oracle.lsop12_decode(nr, nc, packs[k])
        except Exception:
            ref = None
        if ref is None:
            assert st[k] != 0, k
        else:
            assert st[k] == 0 and np.array_equal(vals[k], ref), (k, st[k], k in damaged)
            n_ok += 1
    assert n_ok >= len(packs) // 2
