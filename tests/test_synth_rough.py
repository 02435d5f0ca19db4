"""The rough synthetic surface (oracle generator; CPU only): the statistics bench.py's etopo1_rough workload claims."""
import numpy as np

import oracle


def test_rough_surface_statistics():
    """What bench.py's etopo1_rough workload claims about its data (CPU only, oracle generator): each predictor wins at least a
    tenth of a sample of tiles, a few per cent of the row differences need two M32 bytes and a few per mille three."""
    rng = np.random.default_rng(11)
    seed, tpr = oracle.DEM_SEED + 2, 144
    wins = {1: 0, 2: 0, 3: 0}
    d_all = []
    picks = rng.choice(12960, 60, replace=False)
    for t in picks:
        v = oracle.dem_tiles(seed, 120, 150, tpr, int(t), 1, style=oracle.DEM_STYLE_ROUGH)[0]
        wins[oracle.codec_huffman_encode(0, 120, 150, v)[1]] += 1
        d_all.append(np.abs(np.diff(v.reshape(120, 150).astype(np.int64), axis=1)).ravel())
    d = np.concatenate(d_all)
    assert min(wins.values()) >= 6, wins
    two, three = float(((d > 126) & (d <= 254)).mean()), float((d > 254).mean())
    assert 0.02 < two < 0.09 and 0.002 < three < 0.015, (two, three)
