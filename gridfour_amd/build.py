"""Builds libgvrs_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is the product: there is no Python or CPU implementation behind it.  hipcc
cross-compiles without a GPU, so this also runs on the CPU-only build box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgvrs_hip.so")
SOURCES = ["gvrs_api.hip", "gvrs_encode.hip", "gvrs_decode.hip", "gvrs_aux.hip", "gvrs_float.hip",
           "gvrs_canon_encode.hip", "gvrs_canon_decode.hip", "gvrs_lsop.hip", "gvrs_lsop_decode.hip"]
HEADERS = ["gvrs_common.h", "gvrs_kernels.h", "huff_build.h", "gvrs_encode_layout.h", "gvrs_encode_common.h",
           "gvrs_decode_common.h", "gvrs_canon_common.h", "gvrs_canon_decode_common.h", os.path.join("..", "..", "include", "gvrs_hip_codec.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP codec cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles every HIP source for gfx950 and links gridfour_amd/lib/libgvrs_hip.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for s in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    # the legacy decoder once more with 512-thread workgroups (large tiles, see gvrs_decode.hip)
    obj = os.path.join(LIBDIR, "gvrs_decode_t512.o")
    cmd = [hipcc] + FLAGS + ["-DGF_DEC_THREADS=512", "-DGF_DEC_VARIANT", "-c", os.path.join(CSRC, "gvrs_decode.hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
