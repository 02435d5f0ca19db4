"""The GPU inflater (gvrs_inflate.hip, one wave per zlib stream) against the host's zlib: what an inflater produces is
defined by the stream, so every byte must agree -- on every block type (stored, fixed, dynamic), window sizes up to 32 KB,
long and overlapping matches, truncated room (Inflater.inflate(byte[]) semantics), truncated and corrupted input, and the
zlib streams inside the reference's own sample files."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inflate(streams, caps):
    """streams: list of bytes; caps: room per stream.  Returns (outputs, produced, status)."""
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, lib
    from gridfour_amd._lib import check
    from gridfour_amd.sharding import _ptr
    ctx = gridfour_amd.GvrsHipContext(0)
    n = len(streams)
    in_off = np.zeros(n, np.uint64)
    in_len = np.array([len(s) for s in streams], np.uint32)
    pos = 0
    for i, s in enumerate(streams):
        in_off[i] = pos
        pos += (len(s) + 7) // 4 * 4
    blob = np.zeros(pos + 16, np.uint8)
    for i, s in enumerate(streams):
        blob[int(in_off[i]):int(in_off[i]) + len(s)] = np.frombuffer(s, np.uint8)
    out_cap = np.array(caps, np.uint32)
    out_off = np.zeros(n, np.uint64)
    out_off[1:] = np.cumsum((out_cap[:-1].astype(np.uint64) + 15) // 16 * 16)
    total_out = int(out_off[-1]) + int(out_cap[-1]) + 16
    d_in, d_out = DeviceBuffer(ctx, blob.size), DeviceBuffer(ctx, total_out)
    d_prod, d_st = DeviceBuffer(ctx, n * 4 + 16), DeviceBuffer(ctx, n * 4 + 16)
    d_in.upload(blob)
    d_out.fill(0xEE)
    check(lib().gf_inflate_batch_dev(ctx.handle, None, n, d_in.ptr, _ptr(in_off), _ptr(in_len), d_out.ptr, _ptr(out_off), _ptr(out_cap),
                                     d_prod.ptr, d_st.ptr), "gf_inflate_batch_dev")
    ctx.synchronize()
    raw = d_out.download(np.uint8, total_out)
    prod = d_prod.download(np.uint32, n)
    st = d_st.download(np.int32, n)
    outs = [raw[int(out_off[i]):int(out_off[i]) + int(prod[i])].tobytes() for i in range(n)]
    # nothing may be written behind a stream's room
    for i in range(n):
        end = int(out_off[i]) + int(out_cap[i])
        nxt = int(out_off[i + 1]) if i + 1 < n else total_out
        assert (raw[end:nxt] == 0xEE).all(), i
    return outs, prod, st


def _host(stream, cap):
    """one zlib inflate call with all input and `cap` bytes of room"""
    d = zlib.decompressobj()
    try:
        out = d.decompress(stream, cap) if cap else b""
        return out, 0
    except zlib.error:
        return None, -1


def _payloads():
    rng = np.random.default_rng(11)
    p = []
    p.append(b"")                                                   # empty
    p.append(b"a")
    p.append(b"abcabcabcabc" * 50)
    p.append(bytes(70000))                                          # one long run: distance 1, maximal lengths
    p.append(rng.integers(0, 256, 50000, dtype=np.uint8).tobytes())   # incompressible: stored blocks at level 0..9
    p.append(rng.integers(-4, 5, 40000).astype(np.int8).tobytes())    # residual-like: literals, short codes
    p.append((rng.geometric(0.02, 30000) % 251).astype(np.uint8).tobytes())   # many symbols, long codes
    text = b"the quick brown fox jumps over the lazy dog. " * 3000
    p.append(text[:100000])                                         # matches at many distances (> one window)
    p.append(bytes(rng.integers(0, 2, 9000, dtype=np.uint8)) + text[:5000] + bytes(3000))
    big = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    p.append(big + big)                                             # distances beyond 32 KB are impossible: exactly at the window edge
    return p


def test_equals_host_zlib_all_block_types():
    streams, want = [], []
    for data in _payloads():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            c = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
            streams.append(c.compress(data) + c.flush())
            want.append(data)
    outs, prod, st = _inflate(streams, [len(w) for w in want])
    for i, w in enumerate(want):
        assert st[i] == 0 and prod[i] == len(w) and outs[i] == w, (i, st[i], prod[i], len(w))


def test_small_windows_and_multiple_blocks():
    rng = np.random.default_rng(5)
    data = (rng.integers(0, 40, 120000) % 37).astype(np.uint8).tobytes()
    streams = []
    for wbits in (9, 10, 12, 15):
        c = zlib.compressobj(6, zlib.DEFLATED, wbits)
        parts = []
        for i in range(0, len(data), 17000):                        # full flushes: many blocks, stored empties in between
            parts.append(c.compress(data[i:i + 17000]))
            parts.append(c.flush(zlib.Z_FULL_FLUSH if i % 2 else zlib.Z_SYNC_FLUSH))
        parts.append(c.flush())
        streams.append(b"".join(parts))
    outs, prod, st = _inflate(streams, [len(data)] * len(streams))
    for i in range(len(streams)):
        assert st[i] == 0 and outs[i] == data, i


def test_room_and_input_cut_short():
    rng = np.random.default_rng(8)
    data = (rng.integers(0, 20, 30000)).astype(np.uint8).tobytes() + b"xyz" * 4000
    full = zlib.compress(data, 6)
    streams, caps = [], []
    for cap in (0, 1, 100, 4095, 4096, 4097, 29999, len(data) - 1, len(data), len(data) + 1000):
        streams.append(full)
        caps.append(cap)
    for cut in (0, 1, 2, 3, 10, len(full) // 2, len(full) - 5, len(full) - 4, len(full) - 1):
        streams.append(full[:cut])
        caps.append(len(data))
    outs, prod, st = _inflate(streams, [max(c, 1) if False else c for c in caps])
    for i, (s, cap) in enumerate(zip(streams, caps)):
        want, err = _host(s, cap)
        assert st[i] == 0 and err == 0, (i, st[i])
        assert outs[i] == want, (i, cap, len(s), prod[i], len(want))


def test_corrupt_streams_are_reported_like_zlib():
    rng = np.random.default_rng(21)
    data = (rng.integers(0, 30, 20000)).astype(np.uint8).tobytes() + b"pattern" * 500
    good = bytearray(zlib.compress(data, 6))
    streams = []
    bad_adler = bytearray(good)
    bad_adler[-1] ^= 0x40
    streams.append(bytes(bad_adler))                                # checksum
    streams.append(bytes([0x79, 0x9c]) + bytes(good[2:]))           # method != 8
    streams.append(bytes([0x78, 0x9d]) + bytes(good[2:]))           # header check
    streams.append(bytes([0x78, 0xbb]) + bytes(good[2:]))           # preset dictionary
    streams.append(bytes([0x78, 0x9c, 0x07]) + bytes(20))           # block type 3
    streams.append(bytes([0x78, 0x9c, 0x01, 0x05, 0x00, 0x00, 0x00]) + b"hello")   # stored: LEN / NLEN mismatch
    for k in range(24):                                             # bit flips inside the deflate data
        b = bytearray(good)
        b[3 + (k * 97) % (len(good) - 8)] ^= 1 << (k % 8)
        streams.append(bytes(b))
    outs, prod, st = _inflate(streams, [len(data)] * len(streams))
    for i, s in enumerate(streams):
        want, err = _host(s, len(data))
        if i == 3:
            # Z_NEED_DICT is not an error to java.util.zip.Inflater: inflate() returns 0 with needsDictionary() set (Inflater.c),
            # so the kernel reports a stream that produced nothing (python's zlib raises instead)
            assert err and st[i] == 0 and prod[i] == 0, (i, st[i], prod[i])
        elif err:
            assert st[i] == -1, (i, st[i])
        else:
            assert st[i] == 0 and outs[i] == want, (i, st[i], prod[i])


def test_streams_of_the_reference_sample_files(golden_dir):
    """every zlib stream inside the tile packings of the reference's compressed sample files (CodecDeflate: one stream
    behind the 10-byte header; CodecFloat: five behind 4-byte lengths) inflates to what the host's zlib gives"""
    import os
    from gvrs_walk import tile_packings
    streams, want = [], []
    names = [n for n in sorted(os.listdir(os.path.join(golden_dir, "ref_samples"))) if "Comp" in n and "NoComp" not in n]
    for name in names:
        for _, elems in tile_packings(os.path.join(golden_dir, "ref_samples", name), 1).items():
            for blob in elems:
                off = 0
                while off + 2 <= len(blob):
                    if blob[off] == 0x78 and ((blob[off] << 8) | blob[off + 1]) % 31 == 0:
                        try:
                            d = zlib.decompressobj()
                            out = d.decompress(bytes(blob[off:]))
                            if d.eof and len(out) > 0:
                                used = len(blob) - off - len(d.unused_data)
                                streams.append(bytes(blob[off:off + used]))
                                want.append(out)
                                off += used
                                continue
                        except zlib.error:
                            pass
                    off += 1
    assert streams, "the sample files hold Deflate packings"
    outs, prod, st = _inflate(streams, [len(w) for w in want])
    for i, w in enumerate(want):
        assert st[i] == 0 and outs[i] == w, i


def _upload_packings(ctx, packs):
    from gridfour_amd import DeviceBuffer
    n = len(packs)
    off = np.zeros(n + 1, np.uint64)
    pos = 0
    for i, p in enumerate(packs):
        off[i] = pos
        pos += len(p)
    off[n] = pos
    blob = np.frombuffer(b"".join(packs) + bytes(32), np.uint8)
    ln = np.array([len(p) for p in packs], np.uint32)
    d_blob, d_off, d_len = DeviceBuffer(ctx, blob.size), DeviceBuffer(ctx, off.nbytes), DeviceBuffer(ctx, ln.nbytes + 16)
    d_blob.upload(blob)
    d_off.upload(off)
    d_len.upload(ln)
    return d_blob, d_off, d_len, pos


def test_deflate_packings_decoded_on_the_device():
    """gf_deflate_decode_batch_i32_dev: container walk, inflate and M32 decode without the host touching a byte; damaged
    packings get the statuses of the reference's own checks"""
    import gridfour_amd
    import oracle
    from gridfour_amd import DeviceBuffer, lib
    from gridfour_amd._lib import check
    n_rows, n_cols, nt = 60, 70, 24
    tiles = oracle.dem_tiles(oracle.DEM_SEED + 4, n_rows, n_cols, 8, 0, nt).copy()
    tiles[3, 100:200] = -(2 ** 31)                                   # nulls predictor
    codec = gridfour_amd.CodecDeflateHip()
    packs, _, st = codec.encode_batch(0, n_rows, n_cols, tiles)
    assert (st == 0).all()
    for t in (0, 3, 11):
        assert packs[t] == oracle.codec_deflate_encode(0, n_rows, n_cols, tiles[t])[0]
    packs = [bytes(p) for p in packs]
    bad = list(packs)
    b = bytearray(bad[5]); b[-1] ^= 0x10; bad[5] = bytes(b)          # Adler-32 -> DataFormatException -> IOException
    bad[6] = bad[6][:7]                                              # shorter than the header -> out of bounds
    b = bytearray(bad[7]); b[9] = 0x80; bad[7] = bytes(b)            # negative nM32 -> NegativeArraySizeException
    b = bytearray(bad[8]); b[12] ^= 0xFF; bad[8] = bytes(b)          # deflate data damaged
    ctx = codec.ctx
    d_blob, d_off, d_len, total = _upload_packings(ctx, bad)
    d_vals, d_st = DeviceBuffer(ctx, tiles.nbytes), DeviceBuffer(ctx, nt * 4 + 16)
    check(lib().gf_deflate_decode_batch_i32_dev(ctx.handle, None, n_rows, n_cols, nt, d_blob.ptr, total + 32, d_off.ptr, 0, d_len.ptr,
                                                d_vals.ptr, d_st.ptr), "gf_deflate_decode_batch_i32_dev")
    ctx.synchronize()
    got = d_vals.download(np.int32, tiles.size).reshape(tiles.shape)
    st = d_st.download(np.int32, nt)
    for t in range(nt):
        if t in (5, 8):
            assert st[t] == -1, (t, st[t])
        elif t in (6, 7):
            assert st[t] == -2, (t, st[t])
        else:
            assert st[t] == 0 and np.array_equal(got[t], tiles[t]), t


def test_float_packings_decoded_on_the_device():
    import gridfour_amd
    import oracle
    from gridfour_amd import DeviceBuffer, lib
    from gridfour_amd._lib import check
    n_rows, n_cols, nt = 48, 80, 12
    ints = oracle.dem_tiles(oracle.DEM_SEED + 6, n_rows, n_cols, 4, 0, nt)
    vals = (ints.astype(np.float32) * np.float32(0.1)).astype(np.float32)
    vals[2, 7] = np.nan
    vals[2, 8] = -0.0
    codec = gridfour_amd.CodecFloatHip(level=6)
    packs = [bytes(p) for p in codec.encode_floats_batch(1, n_rows, n_cols, vals)]
    assert packs[1] == oracle.codec_float_encode(1, n_rows, n_cols, vals[1].view(np.uint32), 6)
    bad = list(packs)
    b = bytearray(bad[4]); b[len(b) - 40] ^= 0x55; bad[4] = bytes(b)   # inside the last zlib stream
    bad[5] = bad[5][:len(bad[5]) // 2]                               # framing runs off the packing
    ctx = codec.ctx
    d_blob, d_off, d_len, total = _upload_packings(ctx, bad)
    d_vals, d_st = DeviceBuffer(ctx, vals.nbytes), DeviceBuffer(ctx, nt * 4 + 16)
    check(lib().gf_float_decode_batch_f32_dev(ctx.handle, None, n_rows, n_cols, nt, d_blob.ptr, total + 32, d_off.ptr, d_len.ptr,
                                              d_vals.ptr, d_st.ptr), "gf_float_decode_batch_f32_dev")
    ctx.synchronize()
    got = d_vals.download(np.uint32, vals.size).reshape(vals.shape)
    st = d_st.download(np.int32, nt)
    for t in range(nt):
        if t == 4:
            assert st[t] == -1
        elif t == 5:
            assert st[t] == -2
        else:
            assert st[t] == 0 and np.array_equal(got[t], vals[t].view(np.uint32)), t


def test_damaged_and_cut_short_at_once():
    """A bit flipped near the end of the input AND the input cut short behind it, in every combination over the tail of three
    small streams (literals only, i.e. an empty distance set; fixed codes; ordinary dynamic codes): where zlib merely runs out
    of input -- inside a code, inside extra bits, inside a header -- it reports what it has and no error, and an error only
    where a complete field is invalid.  (Found this way of looking by tools/soak.py: a stream that ended inside the extra bits
    of a code-length repeat code had been called invalid.)"""
    rng = np.random.default_rng(8)
    texts = [rng.integers(0, 256, 150).astype(np.uint8).tobytes(),                       # no matches: empty distance set
             b"abcabcabd" * 4,                                                           # short: fixed codes
             (rng.integers(0, 6, 400).astype(np.uint8).tobytes() + b"xyz" * 40)]         # dynamic codes, matches
    streams, caps = [], []
    for text in texts:
        good = zlib.compress(text, 9)
        tail = min(len(good) - 3, 28)
        for cut in range(len(good) - tail, len(good) + 1):
            for byte in range(max(2, cut - 12), cut):
                for bit in range(8):
                    b = bytearray(good[:cut])
                    b[byte] ^= 1 << bit
                    streams.append(bytes(b))
                    caps.append(len(text) + 64)
    outs, prod, st = _inflate(streams, caps)
    n_err = 0
    for i, s in enumerate(streams):
        want, err = _host(s, caps[i])
        if err:
            n_err += 1
            assert st[i] == -1, (i, st[i], prod[i])
        else:
            assert st[i] == 0 and outs[i] == want, (i, st[i], prod[i], len(want))
    assert 0 < n_err < len(streams)


def _bits_to_bytes(bits):
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i >> 3] |= b << (i & 7)
    return bytes(out)


def test_code_length_code_sets_as_zlib():
    """Dynamic-block headers whose code-length code is degenerate, whole and cut short at every byte: no code at all (zlib reads
    every length as 0, a bit each, and fails on the missing end-of-block code -- or just runs out of input), a single one-bit
    code (incomplete: rejected for this table even though a one-code distance set is allowed), and the complete two-code set
    for comparison.  (inftrees.c: `left > 0 && (type == CODES || max != 1)`.)"""
    def header(cl_lens, hlit=0, hdist=0):
        bits = [1, 0, 1]                                           # BFINAL, BTYPE = 2 (dynamic), LSB first
        bits += [(hlit >> i) & 1 for i in range(5)] + [(hdist >> i) & 1 for i in range(5)]
        hclen = len(cl_lens) - 4
        bits += [(hclen >> i) & 1 for i in range(4)]
        for v in cl_lens:
            bits += [(v >> i) & 1 for i in range(3)]
        return bits
    rng = np.random.default_rng(3)
    tails = [list(rng.integers(0, 2, 400)), [0] * 400, [1] * 400]
    heads = [header([0] * 19), header([0] * 4), header([1] + [0] * 18), header([0, 0, 0, 1] + [0] * 15),      # none; one 1-bit code
             header([1, 1] + [0] * 17), header([0] * 3 + [1] + [0] * 10 + [1] + [0] * 4, hlit=3, hdist=2)]     # complete sets
    streams, caps = [], []
    for h in heads:
        for t in tails:
            raw = _bits_to_bytes(h + t)
            z = b"\x78\x9c" + raw
            for cut in list(range(3, 16)) + [len(z) // 2, len(z)]:
                streams.append(z[:cut])
                caps.append(4096)
    outs, prod, st = _inflate(streams, caps)
    n_err = 0
    for i, s in enumerate(streams):
        want, err = _host(s, caps[i])
        if err:
            n_err += 1
            assert st[i] == -1, (i, st[i], prod[i], s[:8].hex(), len(s))
        else:
            assert st[i] == 0 and outs[i] == want, (i, st[i], prod[i], len(want), s[:8].hex(), len(s))
    assert 0 < n_err < len(streams)
