"""The device-resident entry points only enqueue work (include/gvrs_hip_codec.h: "never synchronise, never allocate
(gf_context_reserve first) -> safe for hipGraph capture"): an encode + decode of a batch is captured into a hipGraph on
the context's stream and replayed on new tile data."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("libamdhip64 not loadable")


@pytest.mark.parametrize("codec", ["huffman", "canon"])
def test_encode_decode_replayed_from_a_graph(codec):
    import gridfour_amd
    hip = _hip()
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 60, 80, 256
    ctx.reserve(nr, nc, nt)
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt, codec=codec)
    b.synth_dem(0x9E3779B97F4A7C15 + 1, 16)
    b.encode()                                   # warm-up outside the capture (module load, LDS attributes)
    b.decode()
    ctx.synchronize()
    stream = C.c_void_p(ctx.stream)
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0               # hipStreamCaptureModeGlobal
    b.encode()
    b.decode()
    assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0 and graph.value
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    for k in range(3):
        b.synth_dem(0x9E3779B97F4A7C15 + 10 + k, 16, tile0=1000 * k)           # new tiles, same buffers
        b.decoded.fill(0)
        ctx.synchronize()
        assert hip.hipGraphLaunch(gexec, stream) == 0
        ctx.synchronize()
        assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
        assert np.array_equal(b.get_decoded(), b.get_values()), k
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
