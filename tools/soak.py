#!/usr/bin/env python3
"""Randomised parity soak: random tile shapes, data kinds and seeds through every codec of the library against the oracle
(encode bytes, chosen predictor / container type, decode, error statuses) for a wall-clock budget.  Not part of the test
suite (tests/test_gpu_random_shapes.py is the fixed sample of it); prints the failing case's seed and stops.
    python tools/soak.py [seconds] [seed]
    python tools/soak.py replay <seed> <case> [out.npz]     the tiles of one case of a run, WITHOUT a GPU: the same loop with the
                                                            oracle standing in for the library (the random draws depend on
                                                            the oracle's packings only), saved for tests/ or a closer look"""
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


class _Dry:
    """the oracle behind the library's batch interface (replay mode)"""
    def __init__(self, enc=None, dec=None):
        self.enc, self.dec = enc, dec

    def encode_batch(self, ci, nr, nc, tiles):
        packs, status = [], []
        for v in tiles:
            if self.enc is None:                                   # LSOP12
                ref, typ = oracle.lsop12_encode(ci, nr, nc, v, True)
                err = 0
            else:
                ref, err = expect(self.enc, ci, nr, nc, v)
                typ = 0
            packs.append(ref)
            status.append(err if err else (1 if ref is None else 0))
        if self.enc is None:
            return packs, [oracle.lsop12_encode(ci, nr, nc, v, True)[1] for v in tiles], status
        return packs, None, status

    def decode_batch(self, nr, nc, packs):
        vals, st = [], []
        for p in packs:
            try:
                vals.append((self.dec or oracle.lsop12_decode)(nr, nc, p))
                st.append(0)
            except IOError as ex:
                vals.append(np.zeros(nr * nc, np.int32))
                st.append(1 if "rc=1" in str(ex) else -1)
        return vals, np.array(st)

    def encode_floats_batch(self, ci, nr, nc, f):
        return [oracle.codec_float_encode(ci, nr, nc, f[t].view(np.uint32), 6) for t in range(f.shape[0])]

    def decode_floats_batch(self, nr, nc, pk):
        return np.stack([oracle.codec_float_decode(nr, nc, p) for p in pk]).view(np.float32), np.zeros(len(pk), np.int32)


def main():
    replay = len(sys.argv) > 3 and sys.argv[1] == "replay"
    budget = 1e9 if replay else float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    stop_at = int(sys.argv[3]) if replay else -1
    rng = np.random.default_rng(seed)
    if replay:
        fams = [("huffman", _Dry(oracle.codec_huffman_encode, oracle.codec_huffman_decode), oracle.codec_huffman_encode, oracle.codec_huffman_decode),
                ("canon", _Dry(oracle.codec_canon_encode, oracle.codec_canon_decode), oracle.codec_canon_encode, oracle.codec_canon_decode),
                ("deflate", _Dry(oracle.codec_deflate_encode, oracle.codec_deflate_decode), oracle.codec_deflate_encode, oracle.codec_deflate_decode)]
        lsop = fl = _Dry()
    else:
        ctx = gridfour_amd.GvrsHipContext(0)
        fams = [("huffman", gridfour_amd.CodecHuffmanHip(context=ctx), oracle.codec_huffman_encode, oracle.codec_huffman_decode),
                ("canon", gridfour_amd.CodecCanonHuffmanHip(context=ctx), oracle.codec_canon_encode, oracle.codec_canon_decode),
                ("deflate", gridfour_amd.CodecDeflateHip(context=ctx), oracle.codec_deflate_encode, oracle.codec_deflate_decode)]
        lsop = gridfour_amd.LsCodecHip(context=ctx, deflate_enabled=True)
        lsop_nd = gridfour_amd.LsCodecHip(context=ctx, deflate_enabled=False)
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
        if n_cases == stop_at:
            out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "soak_case.npz")
            np.savez(out, tiles=tiles, shape=np.array([nr, nc]), seed=np.array([seed, n_cases]))
            print("saved", tag, "->", out)
            return
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
                    try:
                        want = dec(nr, nc, good[k])
                    except IOError:
                        # (canon: a residual inside the -8388608 .. -8333608 gap of CanonicalHuffman.java:258 / :395 -- the
                        # reference cannot read back what it wrote, and neither may the library; seed 30031004 case 9894)
                        assert st[k] != 0, (tag, name, "undecodable stream accepted", t)
                        continue
                    assert st[k] == 0 and np.array_equal(vals[k], want), (tag, name, "decode", t, st[k])
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
        if nr * nc <= 40000 and not replay:
            # round 5: the canonical container alone -- the device-resident form of the encoder (k_lsop_predict16: tile in LDS as digit
            # planes, int16 residuals; falls back per tile to the int32 kernels) -- with and without the value checksum
            cs = bool(n_cases & 1)
            lsop_nd.setValueChecksumEnabled(cs)
            packs, types, status = lsop_nd.encode_batch(5, nr, nc, tiles)
            good = []
            for t, v in enumerate(tiles):
                ref, typ = oracle.lsop12_encode(5, nr, nc, v, False, value_checksum=cs)
                if ref is None:
                    assert packs[t] is None and status[t] == 1, (tag, "lsop canonical", t, status[t])
                else:
                    assert status[t] == 0 and types[t] == typ and packs[t] == ref, (tag, "lsop canonical", t, status[t], types[t], typ, cs)
                    good.append((t, ref))
            if good:
                vals, st = lsop_nd.decode_batch(nr, nc, [g[1] for g in good])
                for k, (t, ref) in enumerate(good):
                    try:
                        want = oracle.lsop12_decode(nr, nc, ref)
                    except IOError:
                        assert st[k] != 0, (tag, "lsop canonical: undecodable stream accepted", t)
                        continue
                    assert st[k] == 0 and np.array_equal(vals[k], want), (tag, "lsop canonical decode", t, int(st[k]))
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
