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
    # jsm::log10_fin (fn 2): the branch-free form the feature reductions use, for positive normal finite arguments — the V8 vectors of that kind, every
    # power of two and its neighbours (log_e's |f| < 2^-20 arm), fp32 energies as the features see them, and a dense random sweep against jsm::log10
    ok = np.isfinite(lx) & (lx >= 2.3e-308)
    assert ok.sum() > 800 and np.array_equal(run(2, lx[ok]).view(np.uint64), lw[ok].view(np.uint64))
    rng2 = np.random.default_rng(11)
    p2 = 2.0 ** np.arange(-1000, 1000, dtype=np.float64)
    near = np.concatenate([p2, np.nextafter(p2, np.inf), np.nextafter(p2, 0), p2 * (1 + 2.0 ** -21), p2 * (1 - 2.0 ** -22), p2 * (1 + 2.0 ** -19), p2 * 1.4142135623730951, p2 * 1.41421356237])
    f32 = np.abs(rng2.standard_normal(400000).astype(np.float32) * np.float32(10.0) ** rng2.integers(-3, 12, 400000).astype(np.float32)).astype(np.float64)
    f32 = f32[f32 > 0]
    wide = np.exp(rng2.uniform(-700, 700, 400000))
    ints = np.arange(1, 300001, dtype=np.float64)
    for v in (near, f32, wide, ints):
        assert np.array_equal(run(2, v).view(np.uint64), run(0, v).view(np.uint64))
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


def _peak_scan_reference(e):
    """The reference's scan of one u32 frame (ref dist/main.js:2 @B25827) with the noise floor left out
    (every candidate is kept; the gate applies `e[l] > v` later): candidates (i, s, l, last) after the /10 shoulder shrink, g."""
    B = len(e)
    e = [int(x) for x in e]
    out = []
    i = l = s = c = u = 0

    def emit(last):
        nonlocal i, s
        qi, qs = i, s
        while qi < l and 10 * e[qi] < e[l]:
            qi += 1
        while qs > l and 10 * e[qs] < e[l]:
            qs -= 1
        out.append((qi, qs, l, last))
        i, s = qi, qs

    g = 0
    for a in range(1, B):
        g += e[a]
        R = e[a] > e[a - 1] and (a < 2 or e[a] > e[a - 2]) and (a < 3 or e[a] > e[a - 3])
        F = e[a] < e[a - 1] and (a < 2 or e[a] < e[a - 2]) and (a < 3 or e[a] < e[a - 3])
        if R:
            if u in (-1, 0):
                if u == -1 and i <= l and l < s:
                    emit(0)
                i = a - 1; l = a
            else:
                l = a
            u = 1
        elif F:
            if u in (1, -1):
                s = a; u = -1
        elif u == -1:
            c += 1
            if c > 2:
                c = 0
                if i <= l and l < s:
                    emit(0)
                u = 0
        elif u == 1 and e[a] > e[a - 1]:
            l = a
        if a == B - 1 and u == 1:
            s = a; l = a
            if i < l and l <= s:
                emit(1)
    return out, g


def _peak_frames(rng, n, B):
    """frames that reach every corner of the scan: smooth humps, noise, plateaus, long ramps (candidates wider than the LDS
    ring of the lane-per-frame kernel), amplitudes up to 2^32 - 1 (prefix sums above 2^32), zeros"""
    rows = []
    x = np.arange(B)
    for k in range(n):
        kind = k % 8
        if kind == 0:
            r = rng.integers(0, 1 << rng.integers(4, 33), B, dtype=np.uint64)
        elif kind == 1:
            r = sum(rng.uniform(10, 1e6) * np.exp(-0.5 * ((x - rng.uniform(0, B)) / rng.uniform(0.7, 6)) ** 2) for _ in range(rng.integers(1, 12))) + rng.uniform(0, 30, B)
        elif kind == 2:
            r = np.repeat(rng.integers(0, 50, (B + 3) // 4), 4)[:B] * rng.integers(1, 1000)          # plateaus: flat runs
        elif kind == 3:
            r = np.cumsum(rng.integers(0, 3, B)) * rng.integers(1, 1 << 20)                          # long weak ramps
            if k % 16 == 3: r = r[::-1].copy()
        elif kind == 4:
            r = np.full(B, (1 << 32) - 1, dtype=np.uint64) - rng.integers(0, 5, B).astype(np.uint64)   # saturated: P crosses 2^32 every bin
        elif kind == 5:
            r = np.where(rng.random(B) < 0.3, rng.integers(0, 1 << 31, B), rng.integers(0, 20, B))
        elif kind == 6:
            r = np.zeros(B) if k % 16 == 6 else np.abs(np.sin(x * rng.uniform(0.05, 1.5))) * 10 ** rng.uniform(1, 9)
        else:
            base = np.cumsum(rng.integers(1, 4, B)).astype(np.float64) ** rng.uniform(1, 4)          # strictly rising over the whole frame
            r = base if k % 16 == 7 else np.concatenate([base[: B // 2], base[: B - B // 2][::-1]])
        rows.append(np.clip(np.asarray(r, dtype=np.float64), 0, 2.0 ** 32 - 1).astype(np.uint32))
    return np.stack(rows)


@pytest.mark.parametrize("bands", [128, 96, 33, 1, 2, 3, 4, 31, 32, 64, 65, 127, 200, 256])
def test_peak_scan_kernels_equal_the_reference_scan(bands):
    """Frame records of the peak-scan kernels (modes 4 / 3: lane per frame — bit masks + event-driven state machine — in rounds of 32 / 16
    bins; mode 2: wave per frame) against a plain restatement of the reference's scan, field by field: candidates, shrunk shoulders, exact prefix sums,
    g, n, the largest candidate."""
    from webspeechanalyzer_amd import capi
    L = capi.lib()
    L.wsa_debug_peaks.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4
    L.wsa_debug_peaks.restype = ctypes.c_int
    rng = np.random.default_rng(100 + bands)
    n = 64 * 5 + 17
    spec = np.ascontiguousarray(_peak_frames(rng, n, bands))
    want = [_peak_scan_reference(row) for row in spec]
    for mode in (4, 3, 2):
        if mode == 2 and bands > 128:
            continue
        hdr = np.zeros((n, 4), np.uint32); amp = np.zeros((n, 64), np.uint32); ent = np.zeros((n, 64, 4), np.uint32); flags = np.zeros(1, np.uint32)
        assert L.wsa_debug_peaks(0, spec.ctypes.data, n, bands, mode, hdr.ctypes.data, amp.ctypes.data, ent.ctypes.data, flags.ctypes.data) == 0
        for f in range(n):
            cands, g = want[f]
            row = [int(v) for v in spec[f]]
            P = np.concatenate([[0], np.cumsum(np.asarray(row, dtype=object))])          # P[x + 1] = sum e[0..x]
            if len(cands) > 64:
                assert flags[0] & 1
                continue
            hy = int(hdr[f, 1])
            assert (int(hdr[f, 0]) | ((hy & 0xff) << 32)) == g, (mode, f)
            assert (hy >> 8) & 0xff == len(cands), (mode, f, (hy >> 8) & 0xff, len(cands))
            assert int(hdr[f, 3]) == f * 64
            best_amp, best_bin = 0, 0
            for k, (qi, qs, l, last) in enumerate(cands):
                w = int(ent[f, k, 0])
                assert (w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, w >> 24) == (qi, qs, l, last), (mode, f, k)
                assert int(amp[f, k]) == row[l]
                hi = int(ent[f, k, 3])
                assert int(ent[f, k, 1]) | ((hi & 0xff) << 32) == int(P[qi]), (mode, f, k)
                assert int(ent[f, k, 2]) | (((hi >> 8) & 0xff) << 32) == int(P[qs + 1]), (mode, f, k)
                if not last and row[l] > best_amp:
                    best_amp, best_bin = row[l], l
            assert (int(hdr[f, 2]), (hy >> 16) & 0xff) == (best_amp, best_bin), (mode, f)
        assert flags[0] == 0 or bands > 128
