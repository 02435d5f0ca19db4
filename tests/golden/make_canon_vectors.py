#!/usr/bin/env python3
"""Generates tests/golden/canon_vectors.json from oracle/canon_ref.py -- the pure-Python, class-by-class restatement of the
reference's canonicalHuffman package written independently of the C oracle.  Deterministic (fixed seeds); run in the build
container:  python tests/golden/make_canon_vectors.py

The vectors are DATA: int inputs and the bytes the reference's algorithm produces for them according to that restatement.
tests/test_oracle_canon_vectors.py holds the C oracle to them, tests/test_gpu_canon_vectors.py the GPU.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import canon_ref as R  # noqa: E402

NULL = R.INT4_NULL_CODE


def stream_case(name, streams, note=""):
    streams = [[int(v) for v in s] for s in streams]
    data, ends = R.canon_encode_streams(streams)
    case = {"name": name, "streams": streams, "hex": data.hex(), "end_bits": ends, "note": note}
    # what the reference's own decoder makes of these bytes (the text arrays have the capacity the callers give them)
    try:
        outs, pos = R.canon_decode_streams(data, [len(s) for s in streams])
        case["decoded"] = outs
        case["decoded_end_bits"] = pos
        case["roundtrip"] = outs == streams and pos == ends
    except (IndexError, ValueError) as e:
        case["decoded"] = None
        case["roundtrip"] = False
        case["decode_error"] = type(e).__name__
    return case


def codec_case(name, n_rows, n_cols, values, codec_index=3):
    values = [int(v) for v in np.asarray(values).ravel()]
    case = {"name": name, "rows": n_rows, "cols": n_cols, "codec_index": codec_index, "values": values}
    try:
        packing, used = R.CodecCanonHuffman().encode(codec_index, n_rows, n_cols, values)
        case["hex"] = None if packing is None else packing.hex()
        case["predictor"] = used
    except ValueError:
        case["throws"] = True
    return case


def main():
    rng = np.random.default_rng(20240607)
    streams = []
    streams.append(stream_case("two_symbols", [[0, 0, 0]]))
    streams.append(stream_case("single_value", [[1]]))
    streams.append(stream_case("byte_range_ends", [[-128, 127]]))
    for nm, vals in (("esc2_a", [128]), ("esc2_b", [-129]), ("esc2_ends", [511, -512, 512, -513]),
                     ("esc4_ends", [2047, -2048, 2048, -2049]), ("esc6_ends", [8191, -8192, 8192, -8193]),
                     ("esc8_ends", [32767, -32768, 32768, -32769]),
                     ("esc16_24_ends", [8388607, -8333608, 8388608, 2 ** 31 - 1, -2 ** 31 + 1]),
                     ("null_code_mix", [NULL, 5, NULL, NULL, 0, -7, NULL])):
        streams.append(stream_case(nm, [vals]))
    # the -8333608 / -8388608 mismatch between encode (:258) and countSymbols (:395): counted as 2-byte escapes, written as
    # 3-byte ones; the high byte's symbol may have no code at all.  The bytes are what the encoder emits; whether its own
    # decoder reads them back is recorded.
    streams.append(stream_case("quirk_gap_low_end", [[-8388608, 3]], note="value inside the gap of CanonicalHuffman.java:258 vs :395"))
    streams.append(stream_case("quirk_gap_high_end", [[-8333609, -8333608, 5, 5]], note="last value inside / first outside the gap"))
    streams.append(stream_case("quirk_gap_with_coded_high_byte", [[-8388000, -1, -1, -1, -2 ** 31 + 5]],
                               note="a 3-byte value gives symbol 127 / 0 a code, so the gap value's high byte is written"))
    geo = lambda n, p: (rng.geometric(p, n) - 1) * rng.choice([-1, 1], n)
    streams.append(stream_case("geometric_small", [geo(300, 0.25)]))
    streams.append(stream_case("geometric_wide", [geo(400, 0.01)]))
    streams.append(stream_case("uniform_bytes_all_symbols", [rng.integers(-128, 128, 700)]))
    streams.append(stream_case("mixed_magnitudes", [np.concatenate([geo(200, 0.3), rng.integers(-40000, 40000, 30),
                                                                     rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, 10)])]))
    # counts like Fibonacci numbers: the unrestricted Huffman depth exceeds 15 -> PackageMerge
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    text = np.concatenate([np.full(c, i - 11, np.int64) for i, c in enumerate(fib)])
    rng.shuffle(text)
    streams.append(stream_case("package_merge_fibonacci", [text[:6000]], note="code lengths limited to 15"))
    fib2 = fib[:19]
    text2 = np.concatenate([np.full(c, (i * 7) % 200 - 100, np.int64) for i, c in enumerate(fib2)])
    rng.shuffle(text2)
    streams.append(stream_case("package_merge_scattered_symbols", [text2]))
    streams.append(stream_case("long_zero_runs_in_length_table", [[-128] * 5 + [127] * 3 + [0] * 9]))
    streams.append(stream_case("repeat_prev_runs", [list(range(-20, 21)) * 3]))
    # two streams in one bit store (LsEncoder12.java:148-151)
    streams.append(stream_case("two_streams_small", [[3, -1, 0, 700, 2], [0] * 40 + [-70000]]))
    streams.append(stream_case("two_streams_lsop_like", [geo(60, 0.05), geo(500, 0.3)]))
    streams.append(stream_case("two_streams_second_single_symbol", [geo(30, 0.2), [0] * 25]))

    codec = []

    def dem(nr, nc, amp, seed):
        r = np.random.default_rng(seed)
        y, x = np.mgrid[0:nr, 0:nc]
        return (1000 + amp * np.sin(x / 3.1) * np.cos(y / 4.3) + r.integers(-2, 3, (nr, nc))).astype(np.int64)

    codec.append(codec_case("smooth_9x11", 9, 11, dem(9, 11, 40, 1)))
    codec.append(codec_case("smooth_20x25", 20, 25, dem(20, 25, 300, 2), codec_index=0))
    codec.append(codec_case("ramp_linear_wins_8x16", 8, 16, np.add.outer(np.arange(8) * 1000, np.arange(16) * 37)))
    codec.append(codec_case("noisy_12x13", 12, 13, rng.integers(-30000, 30000, (12, 13))))
    codec.append(codec_case("wide_residuals_6x7", 6, 7, rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, (6, 7))))
    v = dem(10, 12, 25, 3)
    v[2, 3:7] = NULL
    v[5, 0] = NULL
    v[9, 11] = NULL
    codec.append(codec_case("nulls_10x12", 10, 12, v))
    v = dem(7, 9, 10, 4)
    v[:, 0] = NULL
    codec.append(codec_case("first_column_null_7x9", 7, 9, v))
    codec.append(codec_case("uniform_4x5", 4, 5, np.full((4, 5), -77), codec_index=9))
    codec.append(codec_case("all_null_4x5", 4, 5, np.full((4, 5), NULL)))
    codec.append(codec_case("two_by_two", 2, 2, [[5, 9], [-3, 100000]]))
    codec.append(codec_case("one_row_triangle_throws", 1, 9, np.arange(9) * 3))
    codec.append(codec_case("two_rows_37", 2, 37, dem(2, 37, 80, 5)))
    v = dem(6, 6, 5, 6).astype(np.int64)
    v[3, 3] += 8388608 - 300                  # a Differencing residual near the 2-byte / 3-byte escape boundary
    codec.append(codec_case("big_step_6x6", 6, 6, v))

    out = {"generator": "tests/golden/make_canon_vectors.py over oracle/canon_ref.py (pure-Python restatement of "
                        "compress/canonicalHuffman/*.java, written independently of gvrs_oracle_canon.c)",
           "streams": streams, "codec": codec}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "canon_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s: %d stream cases, %d codec cases, %d bytes" % (path, len(streams), len(codec), os.path.getsize(path)))
    for c in streams:
        if not c["roundtrip"]:
            print("  not read back by the reference's own decoder:", c["name"], c.get("decode_error", "(different values)"))


if __name__ == "__main__":
    main()
