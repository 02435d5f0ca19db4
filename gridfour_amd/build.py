"""Builds libgvrs_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The library is the product: there is no Python or CPU implementation behind it.  hipcc
cross-compiles without a GPU, so this also runs on the CPU-only build box.

Two flavours from the same sources:
  libgvrs_hip.so       what ships: no diagnostics in the kernels
  libgvrs_hip_diag.so  -DGF_DIAG: cycle stamps per phase, phase ablation and the gf_internal_* hooks that tools/ use
                       (loaded instead of the shipping library when GVRS_HIP_DIAG=1 is set; never by tests or bench)

Concurrency: several processes may import the package at once (torchrun ranks, pytest-xdist).  A build runs under an
exclusive file lock, compiles into a private directory and moves the finished library into place with os.replace, so a
reader never maps a half-written file and two builders never share object files.
"""
import fcntl
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgvrs_hip.so")
LIB_DIAG = os.path.join(LIBDIR, "libgvrs_hip_diag.so")
SOURCES = ["gvrs_api.hip", "gvrs_multi.hip", "gvrs_encode.hip", "gvrs_decode.hip", "gvrs_aux.hip", "gvrs_float.hip",
           "gvrs_canon_encode.hip", "gvrs_canon_decode.hip", "gvrs_lsop.hip", "gvrs_lsop_decode.hip", "gvrs_inflate.hip",
           "gvrs_readahead.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP codec cannot be built")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "gvrs_hip_codec.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def csrc_digest():
    """Short digest of the kernel sources (csrc/*.hip, *.h): what a replayed PMC measurement (profiles/hbm_traffic.json,
    issue_counts.json) records, so that bench.py can tell a measurement of these kernels from one of an earlier state."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile_all(lib, extra, verbose):
    hipcc = _hipcc()
    tmp = tempfile.mkdtemp(prefix=".build-", dir=LIBDIR)
    try:
        jobs = []
        for s in _sources():
            obj = os.path.join(tmp, os.path.splitext(s)[0] + ".o")
            jobs.append(([hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, s), "-o", obj], obj))
        # the legacy decoder once more with 512-thread workgroups, one Huffman cursor per thread (tiles whose LDS footprint
        # leaves room for at most two 256-thread workgroups per CU, see gvrs_decode.hip / decodeBatchDev)
        for threads in (512, 1024):
            obj = os.path.join(tmp, "gvrs_decode_t%d.o" % threads)
            jobs.append(([hipcc] + FLAGS + extra + ["-DGF_DEC_THREADS=%d" % threads, "-DGF_DEC_VARIANT", "-DGF_DEC_MAXQ=%d" % threads, "-c",
                                                     os.path.join(CSRC, "gvrs_decode.hip"), "-o", obj], obj))
        # ... and the canonical decoder with 512-thread workgroups
        obj = os.path.join(tmp, "gvrs_canon_decode_t512.o")
        jobs.append(([hipcc] + FLAGS + extra + ["-DGF_CD_THREADS=512", "-DGF_CD_VARIANT", "-c", os.path.join(CSRC, "gvrs_canon_decode.hip"), "-o", obj], obj))
        # ... and the legacy encoder's two usual kernels with 1024-thread workgroups, for the one-tile-per-call path
        obj = os.path.join(tmp, "gvrs_encode_t1024.o")
        jobs.append(([hipcc] + FLAGS + extra + ["-DGF_ENC_THREADS=1024", "-DGF_ENC_VARIANT", "-c", os.path.join(CSRC, "gvrs_encode.hip"), "-o", obj], obj))
        # a few compiles at a time: the translation units are independent
        width = max(1, min(4, (os.cpu_count() or 2) // 2))
        running = []
        try:
            for cmd, _ in jobs:
                if verbose:
                    print(" ".join(cmd))
                running.append((cmd, subprocess.Popen(cmd)))
                if len(running) >= width:
                    c, p = running.pop(0)
                    if p.wait() != 0:
                        raise subprocess.CalledProcessError(p.returncode, c)
            while running:
                c, p = running.pop(0)
                if p.wait() != 0:
                    raise subprocess.CalledProcessError(p.returncode, c)
        finally:
            for _, p in running:                    # a compile failed: the others finish before their directory goes away
                p.wait()
        out = os.path.join(tmp, os.path.basename(lib))
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + [o for _, o in jobs] + ["-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(out, lib)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def variant_path(name):
    return os.path.join(LIBDIR, "libgvrs_hip_%s.so" % name)


def build(force=False, verbose=False, diag=False, variant=None, variant_flags=()):
    """Compiles every HIP source for gfx950 and links gridfour_amd/lib/libgvrs_hip[_diag].so.
    variant / variant_flags (tools/ only): an experiment build libgvrs_hip_<variant>.so with extra compiler flags."""
    lib = variant_path(variant) if variant else (LIB_DIAG if diag else LIB)
    if not force and not needs_build(lib):
        return lib
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            # another process may have finished the same build while this one waited for the lock
            if force or needs_build(lib):
                _compile_all(lib, (["-DGF_DIAG"] if diag else []) + list(variant_flags), verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    import sys
    if "--variant" in sys.argv:            # python -m gridfour_amd.build --variant NAME -DFLAG ...
        i = sys.argv.index("--variant")
        print(build(force=True, verbose=True, diag="--diag" in sys.argv, variant=sys.argv[i + 1], variant_flags=sys.argv[i + 2:]))
    else:
        print(build(force=True, verbose=True, diag="--diag" in sys.argv))
