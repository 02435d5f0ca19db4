"""N > 1 path on CPU: two gloo ranks shard a tile batch as contiguous ranges with no data-path
collective; the only collectives are the barrier and the MAX of the elapsed time (bench.py's
contract).  The oracle stands in for the device codec here (no GPU on the CPU test box): what is
under test is the partition, the per-rank bookkeeping and that the concatenated shards equal the
single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from gridfour_amd import shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_rows, n_cols, n_tiles, tiles_per_row = 24, 30, 37, 8
    lo, cnt = shard_range(n_tiles, rank, world)
    tiles = oracle.dem_tiles(oracle.DEM_SEED + 9, n_rows, n_cols, tiles_per_row, lo, cnt)
    dist.barrier()
    out, lengths, preds = oracle.batch_huffman_encode(0, n_rows, n_cols, tiles)
    dec = oracle.batch_huffman_decode(n_rows, n_cols, out, lengths)
    assert np.array_equal(dec, tiles)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # max-over-ranks timing
    assert t.item() == float(world)
    # no data-path collective: each rank just reports its own (lengths, bytes)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, lengths=lengths, preds=preds,
             blob=np.concatenate([out[i, :lengths[i]] for i in range(cnt)]) if cnt else np.zeros(0, np.uint8))
    dist.destroy_process_group()


def test_two_rank_tile_sharding(tmp_path):
    import torch.multiprocessing as mp

    import oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n_rows, n_cols, n_tiles, tiles_per_row = 24, 30, 37, 8
    tiles = oracle.dem_tiles(oracle.DEM_SEED + 9, n_rows, n_cols, tiles_per_row, 0, n_tiles)
    out, lengths, preds = oracle.batch_huffman_encode(0, n_rows, n_cols, tiles)
    ref_blob = np.concatenate([out[i, :lengths[i]] for i in range(n_tiles)])
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert int(r0["lo"]) == 0 and int(r1["lo"]) == len(r0["lengths"])
    assert np.array_equal(np.concatenate([r0["lengths"], r1["lengths"]]), lengths)
    assert np.array_equal(np.concatenate([r0["preds"], r1["preds"]]), preds)
    assert np.array_equal(np.concatenate([r0["blob"], r1["blob"]]), ref_blob)
