"""gf_multi_partition / shard_range: contiguous, disjoint, complete, balanced (no GPU needed)."""
import ctypes as C

import pytest


@pytest.mark.parametrize("n_tiles", [0, 1, 7, 8, 12960, 93312, 2 ** 40 + 3])
@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_partition(n_tiles, shards):
    from gridfour_amd import lib, shard_range
    prev = 0
    sizes = []
    for i in range(shards):
        t0, t1 = C.c_size_t(1), C.c_size_t(1)
        lib().gf_multi_partition(n_tiles, shards, i, C.byref(t0), C.byref(t1))
        assert t0.value == prev and t1.value >= t0.value
        lo, n = shard_range(n_tiles, i, shards)
        assert (lo, lo + n) == (t0.value, t1.value)            # the Python ranks and the C shards agree
        prev = t1.value
        sizes.append(t1.value - t0.value)
    assert prev == n_tiles and max(sizes) - min(sizes) <= 1


def test_partition_bad_arguments():
    from gridfour_amd import lib
    t0, t1 = C.c_size_t(5), C.c_size_t(5)
    lib().gf_multi_partition(10, 0, 0, C.byref(t0), C.byref(t1))
    assert (t0.value, t1.value) == (0, 0)
    lib().gf_multi_partition(10, 2, 2, C.byref(t0), C.byref(t1))
    assert (t0.value, t1.value) == (0, 0)


def test_multi_create_without_device():
    """no CPU fallback: without a HIP device the multi-context constructor fails like gf_context_create"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gridfour_amd import lib
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib().gf_multi_create(devs, 2, C.byref(h)) == -5 and not h.value
