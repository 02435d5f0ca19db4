#!/usr/bin/env python3
"""Randomised parity soak: random tile shapes, data kinds and seeds through every codec of the library against the oracle
(encode bytes, chosen predictor / container type, decode, error statuses) for a wall-clock budget.  Not part of the test
suite (tests/test_gpu_random_shapes.py is the fixed sample of it); prints the failing case's seed and stops.
    python tools/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gridfour_amd
import oracle
from tilegen import KINDS, NULL, add_nulls, make_tile


def expect(fn, *args):
    try:
        return fn(*args)[0], 0
    except ValueError as ex:
        return None, (-2 if "rc=-2" in str(ex) else -4)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    ctx = gridfour_amd.GvrsHipContext(0)
    fams = [("huffman", gridfour_amd.CodecHuffmanHip(context=ctx), oracle.codec_huffman_encode, oracle.codec_huffman_decode),
            ("canon", gridfour_amd.CodecCanonHuffmanHip(context=ctx), oracle.codec_canon_encode, oracle.codec_canon_decode),
            ("deflate", gridfour_amd.CodecDeflateHip(context=ctx), oracle.codec_deflate_encode, oracle.codec_deflate_decode)]
    lsop = gridfour_amd.LsCodecHip(context=ctx, deflate_enabled=True)
    fl = gridfour_amd.CodecFloatHip(context=ctx, level=6)
    t0 = time.time()
    n_cases = n_tiles = 0
    while time.time() - t0 < budget:
        mode = rng.integers(0, 4)
        if mode == 0:
            nr, nc = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        elif mode == 1:
            nr, nc = int(rng.integers(1, 200)), int(rng.integers(1, 260))
        elif mode == 2:
            nr, nc = (int(rng.integers(1, 4)), int(rng.integers(1, 3000))) if rng.integers(0, 2) else (int(rng.integers(1, 3000)), int(rng.integers(1, 4)))
        else:
            nr, nc = int(rng.integers(100, 330)), int(rng.integers(100, 330))
        case_seed = int(rng.integers(0, 2**31))
        tiles = []
        for k in range(int(rng.integers(2, 7))):
            kind = KINDS[int(rng.integers(0, len(KINDS)))]
            t = make_tile(kind, nr, nc, seed=case_seed + k).copy()
            r = rng.random()
            if r < 0.25 and nr * nc > 1:
                t = add_nulls(t, nr, nc, float(rng.choice([0.02, 0.3, 0.9])))
            elif r < 0.3:
                t[:] = NULL
            elif r < 0.4:
                t = (t.astype(np.int64) * int(rng.integers(2, 5000))).astype(np.int32)      # wide residuals
            tiles.append(t)
        tiles = np.stack(tiles)
        tag = "seed %d case %d shape %dx%d" % (seed, n_cases, nr, nc)
        for name, codec, enc, dec in fams:
            packs, _, status = codec.encode_batch(7, nr, nc, tiles)
            good, idx = [], []
            for t, v in enumerate(tiles):
                ref, err = expect(enc, 7, nr, nc, v)
                if err:
                    assert packs[t] is None and status[t] == err, (tag, name, t, status[t], err)
                elif ref is None:
                    assert packs[t] is None and status[t] == 1, (tag, name, t, status[t])
                else:
                    assert status[t] == 0 and packs[t] == ref, (tag, name, t, status[t])
                    good.append(ref)
                    idx.append(t)
            if good:
                vals, st = codec.decode_batch(nr, nc, good)
                for k, t in enumerate(idx):
                    assert st[k] == 0 and np.array_equal(vals[k], dec(nr, nc, good[k])), (tag, name, "decode", t, st[k])
                # a damaged copy must not hang or crash and, when accepted, must decode as the oracle decodes it
                bad = bytearray(good[0])
                bad[int(rng.integers(1, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                vals, st = codec.decode_batch(nr, nc, [bytes(bad)])
                # (1 = CodecDeflate's "the inflater gave nothing": decode returns null, CodecDeflate.java:143-154)
                assert st[0] in ((0, 1, -1, -2, -7) if name == "deflate" else (0, -1, -2, -7)), (tag, name, "damaged", st[0])
                if name == "deflate" and int.from_bytes(bytes(bad[6:10]), "little") <= 6 * nr * nc:    # (beyond: rejected up front)
                    try:
                        ref = dec(nr, nc, bytes(bad))
                        ok_ref = 0
                    except IOError as ex:
                        ref, ok_ref = None, (1 if "rc=1" in str(ex) else -1)
                    verdict_differs = (ok_ref == 1 and st[0] != 1) or (ok_ref == -1 and st[0] >= 0)
                    if verdict_differs or (ok_ref == 0 and not (st[0] == 0 and np.array_equal(vals[0], ref))):
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)       # the case, for a look on the CPU
                        np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), bad=np.frombuffer(bytes(bad), np.uint8),
                                 good=np.frombuffer(good[0], np.uint8), shape=np.array([nr, nc]), gpu=vals[0],
                                 ref=ref if ref is not None else np.zeros(0, np.int32), verdicts=np.array([ok_ref, int(st[0])]))
                    if ok_ref == 0:
                        assert st[0] == 0 and np.array_equal(vals[0], ref), (tag, name, "damaged accepted by the oracle", st[0])
                    elif ok_ref == 1:
                        assert st[0] == 1, (tag, name, "damaged: null by the oracle", st[0])
                    else:
                        assert st[0] < 0, (tag, name, "damaged: rejected by the oracle", st[0])
        if nr * nc <= 40000:
            packs, types, status = lsop.encode_batch(7, nr, nc, tiles)
            good, idx = [], []
            for t, v in enumerate(tiles):
                ref, typ = oracle.lsop12_encode(7, nr, nc, v, True)
                if ref is None:
                    assert packs[t] is None and status[t] == 1, (tag, "lsop", t, status[t])
                else:
                    assert status[t] == 0 and types[t] == typ and packs[t] == ref, (tag, "lsop", t, status[t], types[t], typ)
                    good.append(ref)
                    idx.append(t)
                    if nr >= 6 and nc >= 6 and rng.random() < 0.3:
                        try:
                            good.append(oracle.lsop12_encode_legacy_huffman(7, nr, nc, v))
                            idx.append(t)
                        except ValueError:
                            pass
            if good:
                vals, st = lsop.decode_batch(nr, nc, good)
                for k, t in enumerate(idx):
                    # the reference is the arbiter: with extreme values its canonical container does not round-trip
                    # (CanonicalHuffman.java:258 vs :395), and the library reproduces that
                    try:
                        ref = oracle.lsop12_decode(nr, nc, good[k])
                    except IOError:
                        # residuals in (-8388608, -8333608] are counted as one symbol and written as another: the
                        # reference cannot read such a stream back, and neither may the library
                        assert st[k] != 0, (tag, "lsop: undecodable stream accepted", t)
                        continue
                    if not (st[k] == 0 and np.array_equal(vals[k], ref)):
                        bad = np.flatnonzero(vals[k] != ref)
                        np.save("gpurun_out/soak_fail_tile.npy", tiles[t])
                        open("gpurun_out/soak_fail_packing.bin", "wb").write(good[k])
                        raise AssertionError((tag, "lsop decode", t, int(st[k]), "header", good[k][:3].hex(), "oracle ok",
                                              bool(np.array_equal(ref, tiles[t])), "bad cells", bad[:10].tolist(), len(bad),
                                              "got", vals[k][bad[:5]].tolist(), "want", ref[bad[:5]].tolist()))
        if nr * nc <= 70000:
            f = (tiles[:2].astype(np.float32) * np.float32(0.37)).reshape(-1, nr * nc)
            f[0, ::7] = np.nan
            pk = fl.encode_floats_batch(0, nr, nc, f)
            for t in range(f.shape[0]):
                assert pk[t] == oracle.codec_float_encode(0, nr, nc, f[t].view(np.uint32), 6), (tag, "float", t)
            back, st = fl.decode_floats_batch(nr, nc, pk)
            assert (st == 0).all() and np.array_equal(back.view(np.uint32), f.view(np.uint32)), (tag, "float decode")
        n_cases += 1
        n_tiles += len(tiles)
    print("soak ok: %d cases, %d tiles, %.0f s, seed %d" % (n_cases, n_tiles, time.time() - t0, seed))


if __name__ == "__main__":
    main()
