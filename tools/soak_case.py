"""Replays the case tools/soak.py saved (gpurun_out/soak_fail.npz): the damaged CodecDeflate packing through the library and the
oracle, the inflated bytes of the device against the host's zlib (one call with the reference's room, Inflater semantics)."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gridfour_amd, oracle
from test_gpu_inflate import _inflate
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_fail.npz"))
bad = bytes(d["bad"]); nr, nc = map(int, d["shape"])
ctx = gridfour_amd.GvrsHipContext(0)
codec = gridfour_amd.CodecDeflateHip(context=ctx)
vals, st = codec.decode_batch(nr, nc, [bad])
try:
    ref = oracle.codec_deflate_decode(nr, nc, bad)
except IOError as ex:
    ref = None; print("oracle:", ex)
print("device status", st[0], "equal to the oracle:", ref is not None and np.array_equal(vals[0], ref))
if ref is not None:
    bd = np.nonzero(np.asarray(vals[0]) != np.asarray(ref))[0]
    print("cells that differ:", len(bd), bd[:5])
nM32 = int.from_bytes(bad[6:10], "little")
outs, prod, ist = _inflate([bad[10:]], [nM32])
h = zlib.decompressobj()
try:
    host = h.decompress(bad[10:], nM32)
    print("host zlib: %d bytes, eof %s" % (len(host), h.eof))
except zlib.error as ex:
    host = None; print("host zlib error:", ex)
print("device inflate: produced %d status %d" % (prod[0], ist[0]))
if host is not None:
    n = min(len(host), len(outs[0]))
    df = [i for i in range(n) if host[i] != outs[0][i]]
    print("inflated bytes: host %d device %d, first difference %s" % (len(host), len(outs[0]), df[:3]))
