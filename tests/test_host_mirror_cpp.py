"""The C++ host mirror of the reference plug-in interface (gridfour_amd/host/gvrs_hip_codec.hpp)."""
import os
import subprocess

import numpy as np
import pytest

import gridfour_amd
import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build():
    gridfour_amd.lib()                                        # makes sure libgvrs_hip.so exists
    exe = os.path.join(HERE, "csrc", "host_mirror_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "csrc", "host_mirror_test.cpp"),
                           "-L" + os.path.join(ROOT, "gridfour_amd", "lib"), "-lgvrs_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "gridfour_amd", "lib"), "-L/opt/rocm/lib",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    exe = _build()
    if gridfour_amd.lib().gf_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 10 and "no-device" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_parity():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    parts = r.stdout.split()
    assert parts[0] == "ok"
    got = bytes(int(x, 16) for x in parts[2:])
    i = np.arange(33 * 65, dtype=np.int64)
    v = ((i * 7919) % 211 - 100 + (i // 65) * 3).astype(np.int32)
    ref, _ = oracle.codec_huffman_encode(4, 33, 65, v)
    assert got == ref
