"""The fast decode kernel's roomy run (round 5): tiles whose M32 stream outgrows the usual LDS buffer are listed by the tree
pre-pass and decoded by persistent workgroups -- beside the fast run on the context's second stream when the batch has at least
4,096 tiles and the context's last finished batch listed any, behind the pre-pass on one stream otherwise.  Every combination must
decode every tile: the same context sees smooth and rough batches in turn (the host hint changes sides), large and small ones, and a
captured graph replays the two-stream form."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
SEED = 0x9E3779B97F4A7C15 + 2


def _roundtrip(b, ctx, style, tile0=0, sample=53, check_oracle=True):
    b.synth_dem(SEED, 144, tile0=tile0, style=style)
    b.decoded.fill(0)
    b.encode(codec_index=0)
    b.decode()
    ctx.synchronize()
    assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
    vals = b.get_values()
    assert np.array_equal(b.get_decoded(), vals)
    if check_oracle:
        lengths, preds = b.get_lengths(), b.get_predictors()
        for t in range(0, b.n_tiles, sample):
            ref, used = oracle.codec_huffman_encode(0, b.n_rows, b.n_cols, vals[t])
            assert preds[t] == used and b.get_packing(t, int(lengths[t])) == ref, t
    return vals


def _n_roomy(vals, n_rows, n_cols, lengths, b):
    """tiles of the batch whose packing's nM32 field is beyond the fast run's LDS budget (what the pre-pass lists)"""
    cells = n_rows * n_cols
    lim = (min(max(cells + cells // 8 + 512, 8192), 98304) + 31) // 32 * 32      # gf_huffman_decode_lds_m32 (gvrs_decode.hip)
    n = 0
    for t in range(b.n_tiles):
        hdr = b.get_packing(t, 10)
        n += int.from_bytes(hdr[6:10], "little") > lim
    return n


def test_smooth_and_rough_batches_in_turn_on_one_context():
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 4608                      # >= 4,096 tiles: the two-stream form is allowed
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt)
    smooth, rough = 0, oracle.DEM_STYLE_ROUGH
    # first batch of a context: the hint is unknown (two streams); then it follows what the batches held
    for k, style in enumerate([smooth, smooth, rough, rough, rough, smooth, rough, smooth, smooth]):
        vals = _roundtrip(b, ctx, style, tile0=(k % 3) * 1000, sample=211)
        if style == rough and k == 2:
            assert _n_roomy(vals, nr, nc, b.get_lengths(), b) > 0, "the rough surface no longer has tiles for the roomy run"
    # several decodes queued without a synchronisation in between (the hint lags by a batch or two)
    b.synth_dem(SEED, 144, tile0=500, style=rough)
    b.encode(codec_index=0)
    for _ in range(4):
        b.decode()
    ctx.synchronize()
    assert (b.get_dec_status() == 0).all() and np.array_equal(b.get_decoded(), b.get_values())
    b.free()


def test_small_rough_batch_takes_the_one_stream_form():
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    b = gridfour_amd.DeviceTileBatch(ctx, 120, 150, 1500)
    for style in (oracle.DEM_STYLE_ROUGH, 0, oracle.DEM_STYLE_ROUGH):
        _roundtrip(b, ctx, style, tile0=3000, sample=97)
    b.free()


def test_small_batches_do_without_the_roomy_launch_when_none_is_expected():
    """Round 6: a batch of fewer than 4,096 tiles whose predecessors on the context listed no tile for the roomy run is decoded without
    that launch -- a rough batch behind smooth ones then has its roomy tiles tried by the first run and taken by the general kernel, and
    the batch after it gets the roomy run again."""
    import gridfour_amd
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 1300
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt)
    smooth, rough = 0, oracle.DEM_STYLE_ROUGH
    seen = 0
    for k, style in enumerate([smooth, smooth, rough, rough, smooth, smooth, smooth, rough, smooth, rough, rough]):
        vals = _roundtrip(b, ctx, style, tile0=2000 + 700 * (k % 4), sample=173)
        if style == rough:
            seen += _n_roomy(vals, nr, nc, b.get_lengths(), b)
    assert seen > 0, "the rough surface no longer has tiles for the roomy run"
    # ... and without a synchronisation between the batches (the hint lags)
    for style in (smooth, rough, smooth, rough):
        b.synth_dem(SEED, 144, tile0=2500, style=style)
        b.encode(codec_index=0)
        b.decode()
        b.decode()
    ctx.synchronize()
    assert (b.get_dec_status() == 0).all() and np.array_equal(b.get_decoded(), b.get_values())
    b.free()


def test_two_stream_decode_replayed_from_a_graph():
    """fork / join events inside a stream capture: the side stream joins the capture and leaves it"""
    import gridfour_amd
    hip = None
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = C.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        pytest.skip("libamdhip64 not loadable")
    ctx = gridfour_amd.GvrsHipContext(0)
    nr, nc, nt = 120, 150, 4200
    ctx.reserve(nr, nc, nt)
    b = gridfour_amd.DeviceTileBatch(ctx, nr, nc, nt)
    rough = oracle.DEM_STYLE_ROUGH
    _roundtrip(b, ctx, rough, check_oracle=False)            # warm-up; the context now knows of roomy tiles: two streams
    _roundtrip(b, ctx, rough, tile0=100, check_oracle=False)
    stream = C.c_void_p(ctx.stream)
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0
    b.encode(codec_index=0)
    b.decode()
    assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0 and graph.value
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)) == 0
    for k, style in enumerate((rough, 0, rough)):
        b.synth_dem(SEED, 144, tile0=2000 + 700 * k, style=style)
        b.decoded.fill(0)
        ctx.synchronize()
        assert hip.hipGraphLaunch(gexec, stream) == 0
        ctx.synchronize()
        assert (b.get_enc_status() == 0).all() and (b.get_dec_status() == 0).all()
        assert np.array_equal(b.get_decoded(), b.get_values()), k
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    b.free()
