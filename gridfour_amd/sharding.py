"""Tile-range partition for multi-GPU runs.

Tiles are encoded/decoded with no reference to any other tile (reference:
gvrs/RasterTile.java:237-241 loops elements independently; codecs take only
(nRows, nCols, values)), so a batch shards over the GPUs of a node as contiguous tile-index
ranges with NO data-path collective: rank g of G gets tiles [g*T/G, (g+1)*T/G).
"""


def shard_range(n_tiles, rank, world_size):
    """Returns (first_tile, n_local) of the contiguous range owned by `rank`."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    lo = (n_tiles * rank) // world_size
    hi = (n_tiles * (rank + 1)) // world_size
    return lo, hi - lo
