#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the HIP GVRS tile codec on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload etopo1|dem1024|gebco_shard]

One "step" = one pass of the hot path over one batch of synthetic elevation tiles:
CodecHuffman.encode of every tile (all predictors tried, shortest kept) followed by
CodecHuffman.decode of every packing, inputs and outputs resident in HBM.

Workloads (BASELINE.json configs)
  etopo1      configs[2]: ETOPO1-shaped 10800x21600 int32 grid = 12,960 tiles of 120x150 per GPU
              (default: the configuration north_star's target is quoted on)
  dem1024     configs[1]: 1024 tiles of 200x200
  gebco_shard configs[3]: one eighth of the GEBCO-shaped grid = 11,664 tiles of 200x200 per GPU

Multi-GPU: tiles are independent, so every rank owns a contiguous tile range of the global
grid and there is no data-path collective; per-GPU work is fixed (weak scaling).  The only
collectives are the timing barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "encode+decode MB/s on int32 elevation tiles; bit-exact vs Java ref"
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    #  name         rows cols tiles/GPU tiles_per_row  description
    "etopo1": (120, 150, 12960, 144, "ETOPO1-shaped 10800x21600 int32 grid, 120x150 tiles, full encode+decode roundtrip"),
    "dem1024": (200, 200, 1024, 32, "1024-tile batch, 200x200 int32 synthetic DEM, all 3 predictors + Huffman"),
    "gebco_shard": (200, 200, 11664, 432, "1/8 shard of the GEBCO_2023-shaped 43200x86400 int32 grid, 200x200 tiles"),
    "float256": (256, 256, 4096, 64, "4096 tiles of 256x256 float32 (DEM x 0.1f), CodecFloat byte-plane stage"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="etopo1", choices=sorted(WORKLOADS))
    ap.add_argument("--codec", default="huffman", choices=["huffman", "canon", "lsop", "float"],
                    help="huffman = CodecHuffman (the north-star path, default); canon = CodecCanonHuffman; lsop = LSOP12, canonical container; "
                         "float = CodecFloat plane split/merge (use with --workload float256)")
    ap.add_argument("--cpu-sample-tiles", type=int, default=-1, help="tiles timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU port on every host core")
    ap.add_argument("--no-verify", action="store_true", help="skip the bit-exactness checks")
    return ap.parse_args()


def _pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_hbm.sh writes
    profiles/hbm_traffic.json: FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 corrections applied).
    PMC collection cannot run inside the timed process, so the per-launch figure measured for this same
    workload is read back; None when no measurement for the workload is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get("workload") != workload:
        return None
    total, found = 0, False
    for part in kernel.split("+"):                      # a launch may be a pre-pass kernel + the main kernel
        for name, d in rec.get("kernels", {}).items():
            if part in name:
                total += int(d["traffic"])
                found = True
                break
    return total if found else None


def run_float(args, ctx, rank, world, dist, torch):
    """BASELINE config 5(i): CodecFloat.  The GPU stage is the five byte planes (split + delta on encode, running sums +
    merge on decode, CodecFloat.java:328-458); the Deflate stage is the host's zlib and is timed separately, on a sample,
    through the host entry points.  One step = planes-encode then planes-decode of every tile, device-resident."""
    import gridfour_amd
    from gridfour_amd import DeviceBuffer, DeviceTileBatch, GpuTimer, lib
    from gridfour_amd._lib import check
    n_rows, n_cols, n_tiles, tiles_per_row, descr = WORKLOADS[args.workload]
    cells = n_rows * n_cols
    # floats = integer DEM x 0.1f: generated on the device as int32, converted on the host once
    gen = DeviceTileBatch(ctx, n_rows, n_cols, n_tiles, slot_stride=16)
    gen.synth_dem(0x9E3779B97F4A7C15 + 5, tiles_per_row, tile0=rank * n_tiles)
    ctx.synchronize()
    vals = (gen.get_values().astype(np.float32) * np.float32(0.1)).reshape(n_tiles, cells)
    del gen
    pstride = int(lib().gf_float_planes_bytes(n_rows, n_cols))
    pstride = (pstride + 15) // 16 * 16
    d_in, d_planes, d_out = (DeviceBuffer(ctx, vals.nbytes), DeviceBuffer(ctx, n_tiles * pstride),
                             DeviceBuffer(ctx, vals.nbytes))
    d_in.upload(vals)
    t_enc = [GpuTimer(ctx) for _ in range(args.steps)]
    t_dec = [GpuTimer(ctx) for _ in range(args.steps)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def step(i=None):
        if i is not None:
            t_enc[i].start()
        check(lib().gf_float_planes_encode_dev(ctx.handle, None, n_rows, n_cols, n_tiles, d_in.ptr, d_planes.ptr, pstride), "enc")
        if i is not None:
            t_enc[i].stop()
            t_dec[i].start()
        check(lib().gf_float_planes_decode_dev(ctx.handle, None, n_rows, n_cols, n_tiles, d_planes.ptr, pstride, d_out.ptr), "dec")
        if i is not None:
            t_dec[i].stop()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    enc_avg = float(np.mean([t.elapsed_ms() for t in t_enc]))
    dec_avg = float(np.mean([t.elapsed_ms() for t in t_dec]))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    bit_exact, cpu_baseline, host_path = None, None, None
    if not args.no_verify:
        import oracle
        back = d_out.download(np.uint32, n_tiles * cells).reshape(n_tiles, cells)
        roundtrip_ok = bool(np.array_equal(back, vals.view(np.uint32)))
        planes = d_planes.download(np.uint8, n_tiles * pstride).reshape(n_tiles, pstride)
        nb = int(lib().gf_float_planes_bytes(n_rows, n_cols))
        parity_ok = all(planes[t, :nb].tobytes() == bytes(oracle.float_planes_encode(n_rows, n_cols, vals[t].view(np.uint32)))
                        for t in range(0, n_tiles, max(1, n_tiles // 16)))
        bit_exact = bool(roundtrip_ok and parity_ok)
        ns = args.cpu_sample_tiles if args.cpu_sample_tiles >= 0 else min(n_tiles, 400)    # ~10 s: zlib level 9 dominates
        if ns > 0 and world == 1:
            sub = vals[:ns]
            mb = sub.nbytes / 1e6
            c0 = time.perf_counter()
            packs = [oracle.codec_float_encode(0, n_rows, n_cols, x.view(np.uint32), 9) for x in sub]
            c1 = time.perf_counter()
            dec = [oracle.codec_float_decode(n_rows, n_cols, pk) for pk in packs]
            c2 = time.perf_counter()
            assert all(np.array_equal(np.asarray(d, np.uint32), x.view(np.uint32)) for d, x in zip(dec, sub))
            cpu_baseline = {"value": round(mb / (c2 - c0), 2), "unit": "MB/s", "cores": 1, "kind": "port",
                            "sample": "first %d tiles (%.0f MB): oracle CodecFloat incl. zlib level 9, 1 thread; encode %.1f, "
                                      "decode %.1f MB/s" % (ns, mb, mb / (c1 - c0), mb / (c2 - c1))}
            codec = gridfour_amd.CodecFloatHip(ctx, level=9)
            h0 = time.perf_counter()
            hp = codec.encode_floats_batch(0, n_rows, n_cols, sub)
            h1 = time.perf_counter()
            hv, hst = codec.decode_floats_batch(n_rows, n_cols, hp)
            h2 = time.perf_counter()
            host_path = {"encode_MBps": round(mb / (h1 - h0), 1), "decode_MBps": round(mb / (h2 - h1), 1),
                         "equals_oracle_packings": bool(hp == packs), "compressed_bytes_per_cell": round(sum(map(len, hp)) / (ns * cells), 4),
                         "note": "host entry points on the same sample: PCIe + GPU planes + zlib level 9 on the host's threads"}
            assert np.array_equal(hv.view(np.uint32), sub.view(np.uint32)) and (hst == 0).all()
    steps = args.steps
    raw_mb = n_tiles * cells * 4 / 1e6
    plane_bytes = 4.125                              # sign bit + exponent + three mantissa bytes per cell
    alg_bytes = (4.0 + plane_bytes) * n_tiles * cells
    dom_name, dom_ms = ("k_float_planes_decode", dec_avg) if dec_avg >= enc_avg else ("k_float_planes_encode", enc_avg)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    out = {
        "metric": "CodecFloat plane stage encode+decode MB/s on float32 tiles (GPU stage; Deflate on the host's zlib)",
        "value": round(raw_mb * world * steps / elapsed, 1), "unit": "MB/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.workload, descr), "tile_rows": n_rows, "tile_cols": n_cols, "tiles_per_gpu": n_tiles,
                   "codec": "CodecFloat (sign / exponent / 3 delta-coded mantissa byte planes)", "sharding": "contiguous tile ranges, no collective"},
        "bit_exact": bit_exact, "encode_ms": round(enc_avg, 4), "decode_ms": round(dec_avg, 4),
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": _pmc_traffic(args.workload, dom_name),
                     "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(dom_ms, 4),
                     "roundtrip_frac": round((2 * alg_bytes) / ((enc_avg + dec_avg) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "cpu_baseline": cpu_baseline, "host_path": host_path,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the codec has no CPU path)")
    # GF_BENCH_BACKEND=gloo is a test hook: it lets several ranks share the one GPU of a test box to exercise the
    # multi-rank control flow (RCCL refuses two ranks on one device); the judged runs use the default, RCCL
    backend = os.environ.get("GF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import gridfour_amd
    from gridfour_amd import DeviceTileBatch, GpuTimer

    n_rows, n_cols, n_tiles, tiles_per_row, descr = WORKLOADS[args.workload]
    cells = n_rows * n_cols
    ctx = gridfour_amd.GvrsHipContext(local_rank)
    if (args.codec == "float") != (args.workload == "float256"):
        raise SystemExit("--codec float goes with --workload float256 (and only with it)")
    if args.codec == "float":
        return run_float(args, ctx, rank, world, dist, torch)
    stride = ((2 * cells + 1024) + 15) // 16 * 16           # DEM packings are far below 2 B/cell
    batch = DeviceTileBatch(ctx, n_rows, n_cols, n_tiles, slot_stride=stride, codec=args.codec)
    seed = 0x9E3779B97F4A7C15 + {"dem1024": 1, "etopo1": 2, "gebco_shard": 3}[args.workload]
    # rank r owns the contiguous tile range starting at r * n_tiles of the global grid
    batch.synth_dem(seed, tiles_per_row, tile0=rank * n_tiles)
    ctx.synchronize()

    # one HIP-event pair per timed step and kernel, recorded on the stream the kernels run on and
    # read only after the timed region (no host sync inside it)
    t_enc = [GpuTimer(ctx) for _ in range(args.steps)]
    t_dec = [GpuTimer(ctx) for _ in range(args.steps)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i=None):
        if i is not None:
            t_enc[i].start()
        batch.encode(codec_index=0)
        if i is not None:
            t_enc[i].stop()
            t_dec[i].start()
        batch.decode()
        if i is not None:
            t_dec[i].stop()

    for _ in range(args.warmup):
        step()
    barrier()

    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    enc_ms = [t.elapsed_ms() for t in t_enc]
    dec_ms = [t.elapsed_ms() for t in t_dec]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- verification (outside the timed region) ----------------
    lengths = batch.get_lengths()
    enc_status = batch.get_enc_status()
    dec_status = batch.get_dec_status()
    packed_bytes = int(lengths.astype(np.int64).sum())
    c_per_cell = packed_bytes / float(n_tiles * cells)
    ok_status = bool((enc_status == 0).all() and (dec_status == 0).all())
    bit_exact = None
    cpu_baseline = None
    if rank == 0 and not args.no_verify:
        import oracle
        vals = batch.get_values()
        roundtrip_ok = bool(np.array_equal(batch.get_decoded(), vals))
        preds = batch.get_predictors()
        sample = list(range(0, n_tiles, max(1, n_tiles // 64)))[:64]
        parity_ok = True
        for t in sample:
            if args.codec == "lsop":
                ref, _ = oracle.lsop12_encode(0, n_rows, n_cols, vals[t], False)
                used = preds[t]
            else:
                ref, used = (oracle.codec_canon_encode if args.codec == "canon" else oracle.codec_huffman_encode)(
                    0, n_rows, n_cols, vals[t])
            if ref != batch.get_packing(t, int(lengths[t])) or used != preds[t]:
                parity_ok = False
                break
        bit_exact = bool(roundtrip_ok and parity_ok and ok_status)

        # ---------------- CPU baseline: the oracle ("port"), 1 core, bounded sample ----------------
        ns = args.cpu_sample_tiles
        if ns < 0:
            ns = min(n_tiles, max(64, int(1200e6 / (4 * cells))))   # up to 1.2 GB of tiles: 9-20 s of CPU work on one core
        if ns > 0 and world == 1:
            sub = vals[:ns]
            c0 = time.perf_counter()
            if args.codec == "lsop":
                out, ln = oracle.batch_lsop12_encode(0, n_rows, n_cols, sub)
            elif args.codec == "canon":
                out, ln, _ = oracle.batch_canon_encode(0, n_rows, n_cols, sub)
            else:
                out, ln, _ = oracle.batch_huffman_encode(0, n_rows, n_cols, sub)
            c1 = time.perf_counter()
            if args.codec == "lsop":
                dec = oracle.batch_lsop12_decode(n_rows, n_cols, out, ln)
            elif args.codec == "canon":
                dec = oracle.batch_canon_decode(n_rows, n_cols, out, ln)
            else:
                dec = oracle.batch_huffman_decode(n_rows, n_cols, out, ln)
            c2 = time.perf_counter()
            assert np.array_equal(dec, sub)
            mb = sub.nbytes / 1e6
            cpu_baseline = {
                "value": round(mb / (c2 - c0), 2), "unit": "MB/s", "cores": 1, "kind": "port",
                "sample": "first %d tiles of the same workload (%.0f MB), oracle C restatement of the Java "
                          "algorithm, 1 thread; encode %.1f MB/s, decode %.1f MB/s" % (
                              ns, mb, mb / (c1 - c0), mb / (c2 - c1)),
            }
            if args.codec == "huffman" and args.cpu_all_cores:
                # the same sample on every host core (tiles are independent: one contiguous share per thread;
                # the oracle's C loops release the GIL inside ctypes)
                import concurrent.futures as cf
                nthr = os.cpu_count() or 1
                shares = [sub[i * ns // nthr:(i + 1) * ns // nthr] for i in range(nthr)]
                shares = [x for x in shares if len(x)]

                def _roundtrip(x):
                    o, l, _ = oracle.batch_huffman_encode(0, n_rows, n_cols, x)
                    return oracle.batch_huffman_decode(n_rows, n_cols, o, l)

                a0 = time.perf_counter()
                with cf.ThreadPoolExecutor(len(shares)) as ex:
                    outs = list(ex.map(_roundtrip, shares))
                a1 = time.perf_counter()
                assert all(np.array_equal(o, x) for o, x in zip(outs, shares))
                cpu_baseline["all_cores"] = {"value": round(mb / (a1 - a0), 2), "unit": "MB/s", "cores": len(shares)}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = args.steps
    raw_mb = n_tiles * cells * 4 / 1e6
    value = raw_mb * world * steps / elapsed
    enc_avg, dec_avg = float(np.mean(enc_ms)), float(np.mean(dec_ms))
    # algorithmic bytes per launch: encode reads 4 B/cell and writes c; decode reads c, writes 4
    alg_bytes = (4.0 + c_per_cell) * n_tiles * cells
    # the decode side of the two Huffman codecs is a per-tile pre-pass kernel followed by the decode kernel: both are inside
    # the HIP-event bracket and both are named, so that the rocprofv3 averages under profiles/ add up to avg_launch_ms
    enc_name = {"canon": "k_canon_encode+k_canon_pack", "lsop": "k_lsop_predict+k_canon_pack2", "huffman": "k_huffman_encode+k_huffman_pack"}[args.codec]
    dec_name = {"canon": "k_canon_parse_lengths+k_canon_decode", "lsop": "k_lsop_unpack2+k_lsop_unpack_m32+k_lsop_reconstruct",
                "huffman": "k_huffman_parse_trees+k_huffman_decode"}[args.codec]
    dom_name, dom_ms = (dec_name, dec_avg) if dec_avg >= enc_avg else (enc_name, enc_avg)
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    traffic = _pmc_traffic(args.workload, dom_name)
    out = {
        "metric": METRIC,
        "value": round(value, 1),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.workload, descr), "tile_rows": n_rows, "tile_cols": n_cols,
                   "tiles_per_gpu": n_tiles, "codec": {"canon": "CodecCanonHuffman (Differencing/Linear/Triangle + canonical Huffman)",
                             "lsop": "LSOP12 (12-coefficient optimal predictor + canonical Huffman container)",
                             "huffman": "CodecHuffman (Differencing/Linear/Triangle + M32 + Huffman)"}[args.codec],
                   "sharding": "contiguous tile ranges, no collective"},
        "bit_exact": bit_exact,
        "compressed_bytes_per_cell": round(c_per_cell, 4),
        "encode_ms": round(enc_avg, 4),
        "decode_ms": round(dec_avg, 4),
        "encode_MBps": round(raw_mb / (enc_avg * 1e-3), 1),
        "decode_MBps": round(raw_mb / (dec_avg * 1e-3), 1),
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "avg_launch_ms": round(dom_ms, 4),
                     "roundtrip_frac": round((2 * alg_bytes) / ((enc_avg + dec_avg) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
