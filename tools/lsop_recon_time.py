#!/usr/bin/env python3
"""Times k_lsop_reconstruct (and k_lsop_predict) alone on the etopo1-shaped batch: predict once, then reconstruct N times
between HIP events.  Honours GVRS_HIP_VARIANT (experiment builds of build.py --variant).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gridfour_amd  # noqa: E402
from gridfour_amd import DeviceTileBatch, GpuTimer, lib  # noqa: E402
from gridfour_amd._lib import check  # noqa: E402


def main():
    nr, nc, nt = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (120, 150, 12960)))
    reps = 10
    ctx = gridfour_amd.GvrsHipContext(0)
    b = DeviceTileBatch(ctx, nr, nc, nt, slot_stride=16, codec="lsop")
    b.synth_dem(0x9E3779B97F4A7C15 + 2, 144)

    def predict():
        check(lib().gf_lsop12_predict_dev(ctx.handle, None, nr, nc, nt, b.values.ptr, b.residuals.ptr, b.res_stride, b.coefs.ptr,
                                          b.scratch_status.ptr), "predict")

    def recon():
        check(lib().gf_lsop12_reconstruct_dev(ctx.handle, None, nr, nc, nt, b.residuals.ptr, b.res_stride, b.coefs.ptr, None,
                                              b.decoded.ptr, b.dec_status.ptr), "reconstruct")
    out = {}
    for name, fn in (("predict", predict), ("reconstruct", recon)):
        fn()
        tm = [GpuTimer(ctx) for _ in range(reps)]
        for t in tm:
            t.start()
            fn()
            t.stop()
        ctx.synchronize()
        out[name + "_ms"] = round(float(np.mean([t.elapsed_ms() for t in tm])), 4)
    out["exact"] = bool(np.array_equal(b.get_decoded(), b.get_values()))
    out["variant"] = os.environ.get("GVRS_HIP_VARIANT", "")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
