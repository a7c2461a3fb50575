"""Device-side pieces tested on their own (through the test entries of libwsa that are not part of include/wsa.h)."""
import ctypes
import json
import os
import struct

import numpy as np
import pytest

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def test_device_log10_and_pow_are_bit_exact_with_v8():
    """csrc/jsmath_device.hpp (fdlibm log10 / V8's variant of e_pow) on every vector of tests/golden/jsmath_v8.json — what Node's own
    Math.log10 / Math.pow returned (tests/golden/gen/make_jsmath.js), incl. the arguments around the `parseInt(pow(10, t - 3) / 20)`
    boundaries of the noise gate (ref dist/main.js:2 @B28615) — plus a dense sweep of the gate's own argument range against the CPU oracle
    (itself pinned to the same vectors, tests/test_oracle_backend.py)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from webspeechanalyzer_amd import capi
    L = capi.lib()
    L.wsa_debug_jsmath.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.wsa_debug_jsmath.restype = ctypes.c_int
    d = json.load(open(os.path.join(GOLDEN, "jsmath_v8.json")))
    h2d = lambda h: struct.unpack(">d", bytes.fromhex(h))[0]

    def run(fn, x, y=None):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y, np.float64) if y is not None else None
        out = np.zeros_like(x)
        assert L.wsa_debug_jsmath(0, fn, x.ctypes.data, y.ctypes.data if y is not None else None, out.ctypes.data, len(x)) == 0
        return out

    lx = np.array([h2d(a) for a, _ in d["log10"]]); lw = np.array([h2d(b) for _, b in d["log10"]])
    assert len(lx) > 1000 and np.array_equal(run(0, lx).view(np.uint64), lw.view(np.uint64))
    pw = [(h2d(a), h2d(b), h2d(c)) for a, b, c in d["pow"] if h2d(a) > 0 and np.isfinite(h2d(c)) and h2d(c) > 1e-300]     # pow_pos: x > 0, normal results
    px, py, pz = (np.array(v) for v in zip(*pw))
    assert len(px) > 4000 and np.array_equal(run(1, px, py).view(np.uint64), pz.view(np.uint64))
    # the gate's arguments: y = ctx_max in [1, 2^33], t = log10(y), then 10 ** (t - 3), 10 ** (t - 2), 10 ** (t / 3)
    rng = np.random.default_rng(5)
    ys = np.concatenate([np.arange(1, 200001, dtype=np.float64), np.floor(10 ** rng.uniform(0, 9.9, 300000)),
                         np.array([10.0 ** k for k in range(10)]), np.array([2e4 * k for k in range(1, 5000)], dtype=np.float64)])
    Lo = pyoracle.lib()
    t_ref = np.array([Lo.wsa_or_log10(float(v)) for v in ys[:60000]])
    t_dev = run(0, ys)
    assert np.array_equal(t_dev[:60000].view(np.uint64), t_ref.view(np.uint64))
    for shift in ("m3", "m2", "d3"):
        e = t_dev - 3 if shift == "m3" else (t_dev - 2 if shift == "m2" else t_dev / 3)
        got = run(1, np.full_like(e, 10.0), e)
        ref = np.array([Lo.wsa_or_pow(10.0, float(v)) for v in e[:60000]])
        assert np.array_equal(got[:60000].view(np.uint64), ref.view(np.uint64)), shift


def test_integer_floor_law_equals_the_f64_evaluation_for_every_ctx_max():
    """gate_floor.hpp: the gate kernel evaluates the noise floor v(ctx_max) (ref dist/main.js:2 @B28615) in integer arithmetic and
    only takes the V8 log10 / pow route where the f64 result's last bit can matter (y a multiple of the arm's divisor, a perfect
    cube, a power of ten).  Compared on the device against the f64 evaluation for ALL 2^32 values of ctx_max."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd import capi
    L = capi.lib()
    L.wsa_debug_floor_law.argtypes = [ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    L.wsa_debug_floor_law.restype = ctypes.c_int
    out = (ctypes.c_uint64 * 3)()
    total_exact = 0
    for lo in range(0, 1 << 32, 1 << 30):
        assert L.wsa_debug_floor_law(0, lo, lo + (1 << 30), out) == 0
        assert out[0] == 0, f"{out[0]} values of ctx_max in [{lo}, {lo + (1 << 30)}) differ, the first is {out[1]}"
        total_exact += out[2]
    assert 0 < total_exact < (1 << 32) // 1000          # the f64 route is the exception


def test_device_match_score_equals_the_reference_function():
    """csrc/tracker_score.hpp on the rows of tests/golden/score_expected.json (what the reference's own `_` returned under Node), bit for bit:
    the x / 1, x / 2, 10 / gap special cases of the device version included."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from webspeechanalyzer_amd import capi
    L = capi.lib()
    L.wsa_debug_score.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.wsa_debug_score.restype = ctypes.c_int
    d = json.load(open(os.path.join(GOLDEN, "score_expected.json")))
    args = np.ascontiguousarray(np.array(d["args"], dtype=np.float64))
    want = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in d["expected_f64_hex"]])
    out = np.zeros(len(args))
    assert L.wsa_debug_score(0, args.ctypes.data, out.ctypes.data, len(args)) == 0
    bad = np.flatnonzero(out.view(np.uint64) != want.view(np.uint64))
    assert len(bad) == 0, f"{len(bad)} rows differ, first {args[bad[0]].tolist()}: {out[bad[0]]!r} vs {want[bad[0]]!r}"
