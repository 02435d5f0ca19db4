"""Statistics of a synthetic workload's tiles on the CPU (test infrastructure: uses the oracle): which predictor wins, how many
M32 bytes the row differences and the winner's residuals need, packed bytes per cell.  tools/rough_stats.py [style] [nSample]
[rows cols tilesPerRow nTilesTotal]"""
import struct
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import oracle  # noqa: E402


def stats(style=1, n_sample=200, n_rows=120, n_cols=150, tpr=144, n_total=12960, seed=oracle.DEM_SEED + 5):
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(n_total, n_sample, replace=False))
    winners = {1: 0, 2: 0, 3: 0}
    nbytes = np.zeros(7, np.int64)
    tot_cells = packed = 0
    multi_tiles = 0
    wide_vals = 0
    for t in idx:
        v = oracle.dem_tiles(seed, n_rows, n_cols, tpr, int(t), 1, style=style)[0]
        g = v.reshape(n_rows, n_cols).astype(np.int64)
        d = np.abs(np.diff(g, axis=1)).ravel()
        for k, (lo, hi) in enumerate(((0, 126), (127, 254), (255, 16638), (16639, 2113790))):
            nbytes[k + 1] += ((d >= lo) & (d <= hi)).sum()
        ref, used = oracle.codec_huffman_encode(0, n_rows, n_cols, v)
        winners[used] += 1
        n_m32 = struct.unpack("<I", ref[6:10])[0]
        wide_vals += n_m32 - (n_rows * n_cols - 1)
        multi_tiles += n_m32 != n_rows * n_cols - 1
        packed += len(ref)
        tot_cells += n_rows * n_cols
    tot = nbytes.sum()
    return {"winners": {k: round(v / n_sample, 3) for k, v in winners.items()},
            "row_diff_bytes_share": {k: round(float(nbytes[k]) / tot, 4) for k in (1, 2, 3, 4)},
            "tiles_with_multibyte": round(multi_tiles / n_sample, 3),
            "extra_m32_bytes_per_cell": round(wide_vals / tot_cells, 4),
            "bytes_per_cell": round(packed / tot_cells, 4)}


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(stats(*a))
