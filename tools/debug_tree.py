"""Dumps the encode kernel's on-chip tables for a few tiles and diffs them against the host
build of the same headers + the oracle (diagnostic tool)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gridfour_amd  # noqa: E402
import oracle  # noqa: E402
from gridfour_amd import DeviceBuffer, DeviceTileBatch, lib  # noqa: E402
from tilegen import make_tile  # noqa: E402


def main():
    hh = C.CDLL(os.path.join(ROOT, "tests", "csrc", "libhost_harness.so"))
    for f in ("hh_sizeof_persist", "hh_sizeof_tree", "hh_off_persist", "hh_off_tree"):
        getattr(hh, f).restype = C.c_size_t
    szP, szT = hh.hh_sizeof_persist(), hh.hh_sizeof_tree()
    offP = [hh.hh_off_persist(i) for i in range(10)]
    offT = [hh.hh_off_tree(i) for i in range(7)]
    print("sizeof persist", szP, "tree", szT, "offP", offP, "offT", offT)
    L = lib()
    L.gf_internal_encode_debug_words.restype = C.c_size_t
    L.gf_internal_set_encode_debug.argtypes = [C.c_void_p]
    words = L.gf_internal_encode_debug_words()
    codec = gridfour_amd.CodecHuffmanHip()
    cases = [("smooth", 10, 10), ("noise8", 7, 9), ("steps", 33, 65)]
    for kind, n_rows, n_cols in cases:
        v = make_tile(kind, n_rows, n_cols)
        b = DeviceTileBatch(codec.ctx, n_rows, n_cols, 1)
        b.values.upload(v)
        dbg = DeviceBuffer(codec.ctx, words * 4).fill(0)
        L.gf_internal_set_encode_debug(dbg.ptr)
        b.encode(codec_index=3)
        codec.ctx.synchronize()
        L.gf_internal_set_encode_debug(None)
        raw = dbg.download(np.uint8, words * 4)
        P = raw[:szP]
        hist = P[offP[0]:offP[0] + 3 * 1024].view(np.uint32).reshape(3, 256)
        tab = P[offP[1]:offP[1] + 3 * 2048].view(np.uint64).reshape(3, 256)
        total = P[offP[3]:offP[3] + 24].view(np.uint64)
        treeEnd = P[offP[4]:offP[4] + 12].view(np.uint32)
        model = P[offP[8]:offP[8] + 12].view(np.int32)
        print("==", kind, n_rows, n_cols, "models", model, "totalBits", total, "treeEnd", treeEnd,
              "len", b.get_lengths(), "pred", b.get_predictors())
        for p in range(3):
            m = p + 1
            m32, seed = oracle.predictor_encode(m, n_rows, n_cols, v)
            ref_hist = np.bincount(np.frombuffer(m32, np.uint8), minlength=256).astype(np.uint32)
            ok_h = np.array_equal(ref_hist, hist[p])
            ref_pack, _ = oracle.codec_huffman_encode(3, n_rows, n_cols, v, predictor_mask=1 << p)
            _, endbits, cl, tb = oracle.huffman_encode(m32)
            print(" model", m, "hist", "ok" if ok_h else "MISMATCH", "ref total bits", 80 + endbits, "gpu", int(total[p]),
                  "ref bytes", len(ref_pack))
            if not ok_h:
                bad = np.nonzero(ref_hist != hist[p])[0]
                print("   hist diffs at", bad[:10], "gpu", hist[p][bad[:10]], "ref", ref_hist[bad[:10]])
            T = raw[szP + p * szT: szP + (p + 1) * szT]
            n = int(T[offT[6]:offT[6] + 4].view(np.int32)[0])
            cnt = T[offT[0]:offT[0] + 511 * 4].view(np.uint32)
            parent = T[offT[1]:offT[1] + 511 * 2].view(np.uint16)
            left = T[offT[2]:offT[2] + 255 * 2].view(np.uint16)
            nl = T[offT[3]:offT[3] + 511 * 2].view(np.uint16)
            sym = T[offT[5]:offT[5] + 256]
            refT = np.zeros(szT, np.uint8)
            rn = hh.hh_build_tree(ref_hist.ctypes.data_as(C.c_void_p), refT.ctypes.data_as(C.c_void_p))
            rcnt = refT[offT[0]:offT[0] + 511 * 4].view(np.uint32)
            rparent = refT[offT[1]:offT[1] + 511 * 2].view(np.uint16)
            rleft = refT[offT[2]:offT[2] + 255 * 2].view(np.uint16)
            rnl = refT[offT[3]:offT[3] + 511 * 2].view(np.uint16)
            rsym = refT[offT[5]:offT[5] + 256]
            print("   n gpu", n, "ref", rn,
                  "| leaves cnt", "ok" if np.array_equal(cnt[:rn], rcnt[:rn]) else "MISMATCH",
                  "| sym", "ok" if np.array_equal(sym[:rn], rsym[:rn]) else "MISMATCH",
                  "| branch cnt", "ok" if np.array_equal(cnt[rn:2 * rn - 1], rcnt[rn:2 * rn - 1]) else "MISMATCH",
                  "| parent", "ok" if np.array_equal(parent[:2 * rn - 1], rparent[:2 * rn - 1]) else "MISMATCH",
                  "| left", "ok" if np.array_equal(left[:rn - 1], rleft[:rn - 1]) else "MISMATCH",
                  "| nl", "ok" if np.array_equal(nl[:2 * rn - 1], rnl[:2 * rn - 1]) else "MISMATCH")
            if not np.array_equal(cnt[:rn], rcnt[:rn]) or not np.array_equal(sym[:rn], rsym[:rn]):
                print("     gpu cnt", cnt[:min(rn, 16)], "sym", sym[:min(rn, 16)])
                print("     ref cnt", rcnt[:min(rn, 16)], "sym", rsym[:min(rn, 16)])
            elif not np.array_equal(parent[:2 * rn - 1], rparent[:2 * rn - 1]):
                print("     gpu parent", parent[:min(2 * rn - 1, 24)])
                print("     ref parent", rparent[:min(2 * rn - 1, 24)])
            lens_gpu = (tab[p] >> np.uint64(56)).astype(np.int64)
            used = ref_hist > 0
            print("   code lens", "ok" if np.array_equal(lens_gpu[used], cl[used].astype(np.int64)) else
                  "MISMATCH gpu %s ref %s" % (lens_gpu[used][:12], cl[used][:12]))
        b.free()
        dbg.free()


if __name__ == "__main__":
    main()
