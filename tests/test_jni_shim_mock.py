"""The JNI shim (gridfour_amd/java/gvrs_hip_jni.cpp) compiled and CALLED over a stand-in JNIEnv (tests/csrc/jni_mock/jni.h): the
image holds no JDK, so this is the only compiler and the only caller those lines meet here.  Test infrastructure: it shows that the
shim builds against libgvrs_hip.so, moves arrays and raises the reference's exceptions; a real JVM is not involved."""
import os
import subprocess

import numpy as np
import pytest

import gridfour_amd
import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _build():
    gridfour_amd.lib()
    exe = os.path.join(HERE, "csrc", "jni_shim_test")
    libdir = os.path.join(ROOT, "gridfour_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(HERE, "csrc", "jni_mock"),
                           "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(HERE, "csrc", "jni_shim_test.cpp"),
                           "-L" + libdir, "-lgvrs_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def test_jni_shim_compiles_and_raises_without_gpu():
    exe = _build()
    if gridfour_amd.lib().gf_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 10 and r.stdout.startswith("no-device"), r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(33, 65), (120, 150)])
def test_jni_shim_every_native_method(shape):
    exe = _build()
    r = subprocess.run([exe, str(shape[0]), str(shape[1])], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    parts = r.stdout.split()
    assert parts[0] == "ok"
    got = bytes(int(x, 16) for x in parts[2:])
    # the tile jni_shim_test.cpp builds (tileOf, salt 0)
    n_rows, n_cols = shape
    i = np.arange(n_rows * n_cols, dtype=np.int64)
    v = ((i * 7919) % 211 - 100 + (i // n_cols) * 3 + ((i % n_cols) ** 2) % 17).astype(np.int32)
    ref, _ = oracle.codec_huffman_encode(4, n_rows, n_cols, v)
    assert got == ref
