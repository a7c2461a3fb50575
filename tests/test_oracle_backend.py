"""Pins the ORACLE's back end (oracle/backend.c) to the reference: the committed fixtures hold what
formantanalyzer@1.1.6 (dist/main.js module 584) itself produced under Node for the same u32
spectra (tests/golden/gen/make_golden.py).  Segment / syllable indices, timestamps, the per-frame
state trace and all 53 doubles must be bit-identical."""
import json
import os
import shutil
import struct

import numpy as np
import pytest

from oracle import pyoracle
from tests.util import GOLDEN, callbacks_equal, jsnum, jsvec, load_backend_golden, same_f64

SPECTRA, CASES = load_backend_golden()


def cfg_of(case):
    s = case["settings"]
    return pyoracle.default_cfg(level=case["level"], bands=int(SPECTRA[case["key"]].shape[1]), window_step=s["window_step"],
                                pause_length=s["pause_length"], min_seg_length=s["min_seg_length"],
                                auto_noise_gate=s["auto_noise_gate"], voiced_max_dB=s["voiced_max_dB"],
                                voiced_min_dB=s["voiced_min_dB"])


@pytest.mark.parametrize("idx", range(len(CASES)), ids=lambda i: f"{CASES[i]['key']}-L{CASES[i]['level']}-ws{CASES[i]['settings']['window_step']}")
def test_backend_matches_reference(idx):
    case = CASES[idx]
    out = pyoracle.run_backend(SPECTRA[case["key"]], cfg_of(case), trace=True)
    assert out["segments_ci"] == case["segments_ci"]
    if case["level"] in (10, 13):
        got = [s for s, f in zip(out["syllables_ci"], out["flags"]) if f >= 0]
        assert got == [s for s in case["syllables_ci"] if s is not None][:len(got)]
    if case.get("trace"):
        ref = np.array([[jsnum(x) for x in row] for row in case["trace"]], dtype=np.float64)
        assert same_f64(ref, out["trace"])
    if case["level"] in (3, 4, 5, 13):
        ok, why = callbacks_equal(case["level"], case["callbacks"], out["callbacks"], exact=True)
        assert ok, why


def test_golden_covers_the_dropped_segment_quirk():
    # at least one fixture clip has a segment the reference lists in segments_ci but never reports
    n = sum(len(c["segments_ci"]) - len(c["callbacks"]) for c in CASES if c["level"] == 5)
    assert n >= 1


def test_formant_features_function_level():
    cases = json.load(open(os.path.join(GOLDEN, "features_expected.json")))["cases"]
    for c in cases:
        got = pyoracle.formant_features(np.array(c["fr"], np.float32), c["ctx_max"], c["floor"], float("nan"))
        assert same_f64(jsvec(c["expected"]), got)


def test_jsmath_bit_exact_with_v8():
    d = json.load(open(os.path.join(GOLDEN, "jsmath_v8.json")))
    L = pyoracle.lib()
    h2d = lambda h: struct.unpack(">d", bytes.fromhex(h))[0]
    d2h = lambda x: struct.pack(">d", x).hex()
    for a, b in d["log10"]:
        assert d2h(L.wsa_or_log10(h2d(a))) == b
    for a, b, c in d["pow"]:
        r = d2h(L.wsa_or_pow(h2d(a), h2d(b)))
        assert r == c or (c[1:4] == "ff8" and r[1:4] == "ff8")


@pytest.mark.reference
@pytest.mark.skipif(not (os.path.exists("/root/reference/dist/main.js") and shutil.which("node")),
                    reason="live reference run needs /root/reference and node (build container)")
def test_live_fuzz_against_reference(tmp_path):
    """Fresh random clips through the reference itself, right now, vs the oracle."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(GOLDEN, "gen"))
    from synth_spectra import synth_clip
    clips, specs = [], []
    for seed in range(9000, 9012):
        sp = synth_clip(seed, 300)
        f = tmp_path / f"{seed}.bin"
        sp.tofile(f)
        for lv in (5, 13):
            clips.append(dict(spectra=str(f), frames=300, bands=128, level=lv, window_step=25, pause_length=200,
                              min_seg_length=50, auto_noise_gate=True, voiced_max_dB=100, voiced_min_dB=10, trace=False))
            specs.append(sp)
    job = tmp_path / "job.json"
    json.dump({"bundle": "/root/reference/dist/main.js", "clips": clips}, open(job, "w"))
    subprocess.run(["node", os.path.join(GOLDEN, "gen", "ref_driver.js"), str(job), str(tmp_path / "out.json")],
                   check=True, stderr=subprocess.DEVNULL)
    res = json.load(open(tmp_path / "out.json"))["results"]
    for c, sp, r in zip(clips, specs, res):
        out = pyoracle.run_backend(sp, pyoracle.default_cfg(level=c["level"]))
        assert out["segments_ci"] == r["segments_ci"]
        ok, why = callbacks_equal(c["level"], r["callbacks"], out["callbacks"], exact=True)
        assert ok, why


def test_match_score_function_level():
    """G2: the tracker's match score `_` (ref dist/main.js:2 @B37340) on 6 000 argument rows around every branch (amplitude ratio .1 / .001 / 1,
    bin distance against velocity, track length 10, gaps 0..3 inside their windows) — the reference's own function under Node
    (tests/golden/gen/make_golden.py) against the C restatement, bit for bit; the device function is held to the same rows in tests/test_gpu_units.py."""
    import struct
    from oracle import pyoracle
    d = json.load(open(os.path.join(GOLDEN, "score_expected.json")))
    L = pyoracle.lib()
    want = np.array([struct.unpack(">d", bytes.fromhex(h))[0] for h in d["expected_f64_hex"]])
    got = np.array([L.wsa_or_match_score(*[float(v) for v in row]) for row in d["args"]])
    assert len(want) > 5000 and np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert (want > 1).sum() > 2000 and (want == 0).sum() > 1000 and np.isinf(want).sum() > 10
