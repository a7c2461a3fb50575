#!/usr/bin/env python3
"""Regenerates the committed back-end fixtures under tests/golden/ by running the REFERENCE's own
code (formantanalyzer@1.1.6 inside /root/reference/dist/main.js) under Node through ref_driver.js.

Build-container only (needs /root/reference and node).  Outputs are data: input spectra + what the
reference returned.  Nothing of the reference's source is written anywhere.

    python3 tests/golden/gen/make_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from synth_spectra import designed_clip, synth_clip  # noqa: E402

BUNDLE = "/root/reference/dist/main.js"
DEFAULT = dict(window_step=25, pause_length=200, min_seg_length=50, auto_noise_gate=True,
               voiced_max_dB=100, voiced_min_dB=10)
APP = dict(DEFAULT, window_step=15, pause_length=250, min_seg_length=100)      # src/index.js:21
FIXED_GATE = dict(DEFAULT, auto_noise_gate=False, voiced_max_dB=120, voiced_min_dB=60)

# (seed, frames, settings name, levels).  Seed 8 holds a segment whose straighten step throws.
CASES = [(s, 400, "default", (5, 13)) for s in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55)]
CASES += [(s, 400, "default", (3, 4, 10, 11, 12)) for s in (1, 8)]
CASES += [(s, 400, "app", (11, 12)) for s in (101,)] + [(5, 400, "default", (12,))] + [(s, 400, "default", (11,)) for s in (2, 5, 5021, 5027)]   # 5021 / 5027: a histogram left as raw counts
CASES += [(s, 400, "app", (5, 13)) for s in (101, 102, 103)]
CASES += [(s, 400, "fixed_gate", (5, 13)) for s in (201, 202)]
CASES += [(301, 1000, "default", (5, 13)), (302, 40, "default", (5, 13)), (303, 3, "default", (5, 13))]
# G2: clips built for single rules (synth_spectra.designed_clip): the noise gate's piecewise floor, the run / gap rule of sep_syllables
CASES += [("gate_sweep", 900, "default", (5, 13)), ("gate_sweep", 900, "app", (5,)), ("syl_edges", 400, "default", (5, 13, 10)), ("syl_edges", 400, "app", (13,))]
SETTINGS = {"default": DEFAULT, "app": APP, "fixed_gate": FIXED_GATE}


def feature_cases():
    rng = np.random.default_rng(7)
    cases = []
    for n in (1, 2, 3, 17, 60):
        fr = np.zeros((n, 9), np.float32)
        for k in range(3):
            on = rng.random(n) < (0.0 if (k == 2 and n == 17) else 0.8)
            fr[:, 3 * k] = np.where(on, np.round(rng.uniform(8 + 25 * k, 30 + 25 * k, n)), 0)
            fr[:, 3 * k + 1] = np.where(on, rng.uniform(0, 1, n) ** 4 * 10.0 ** rng.uniform(1, 8), 0)
            fr[:, 3 * k + 2] = np.where(on, rng.integers(1, 9, n), 0)
        cases.append(dict(fn="formant_features", fr=fr.astype(float).tolist(),
                          ctx_max=float(10.0 ** rng.uniform(2, 8)), floor=float(rng.integers(1, 500))))
    cases.append(dict(fn="formant_features", fr=np.zeros((5, 9)).tolist(), ctx_max=50.0, floor=2.0))
    return cases


def score_args():
    """Arguments of the match score `_` (ref @B37340) as accumulate_fm calls it (@B36100): gap 0..3 inside the search windows 3 / 4 / 6 / 9,
    integer bins and amplitudes, velocities in thirds and halves — on a grid around every branch (s = .1, .001, 1; t = 0, 1; n = 10) plus random ones."""
    rng = np.random.default_rng(11)
    rows = []
    amps = [1, 2, 9, 10, 11, 99, 100, 101, 999, 1000, 1001, 4095, 65536, 1000000, 4294967295]
    for gap, win in ((0, 3), (1, 4), (2, 6), (3, 9)):
        for dist in range(0, win):
            for ta in amps:
                for pa in amps:
                    for n in (1, 2, 9, 10, 11, 40):
                        tb = 30
                        for vel in (0.0, 1.0, -2.0, 1 / 3, -5 / 3, 2.5, 9.0, -9.5):
                            rows.append([gap, dist, n, tb, tb + dist, ta, pa, vel])
                            if dist:
                                rows.append([gap, dist, n, tb, tb - dist, ta, pa, vel])
    rows = [rows[i] for i in rng.choice(len(rows), 4000, replace=False)]
    for _ in range(2000):
        gap = int(rng.integers(0, 4)); win = (3, 4, 6, 9)[gap]
        tb = int(rng.integers(8, 100)); d = int(rng.integers(-(win - 1), win))
        ta = int(10 ** rng.uniform(0, 9.6)); pa = int(max(1, ta * 10 ** rng.uniform(-3.5, 3.5)))
        vel = float(rng.choice([0.0, int(rng.integers(-12, 13)) / 3, int(rng.integers(-12, 13)) / 2, float(int(rng.integers(-12, 13)))]))
        rows.append([gap, abs(d), int(rng.integers(1, 60)), tb, tb + d, min(ta, 4294967295), min(pa, 4294967295), vel])
    rows.append([1, 0, 3, 20, 20, 5, 0, 0.0])          # peak amplitude 0: `if(!(o>0))return 0`
    return rows


def main():
    tmp = tempfile.mkdtemp(prefix="wsa_golden_")
    spectra, clips, meta = {}, [], []
    for seed, frames, sname, levels in CASES:
        key = f"s{seed}_f{frames}" if not isinstance(seed, str) else f"{seed}_f{frames}"
        if key not in spectra:
            spectra[key] = synth_clip(seed, frames) if not isinstance(seed, str) else designed_clip(seed, frames)
            spectra[key].tofile(os.path.join(tmp, key + ".bin"))
        for lv in levels:
            c = dict(SETTINGS[sname], spectra=os.path.join(tmp, key + ".bin"), frames=frames, bands=128,
                     level=lv, trace=(lv == 5))
            clips.append(c)
            meta.append(dict(key=key, settings=SETTINGS[sname], level=lv))
    # a captured spectrum (HIP front end output of one synthetic clip, 96 mel bands, 15 ms hop) on which numeric.uncmin throws
    # inside make_coeffs ("f(x0) is a NaN" after a singular normal matrix): the reference reports the rows collected so far
    cap = np.load(os.path.join(HERE, "captured_l12_throw.npy"))
    spectra["captured_l12_throw"] = cap
    cap.tofile(os.path.join(tmp, "captured_l12_throw.bin"))
    CAP = dict(window_step=15, pause_length=100, min_seg_length=100, auto_noise_gate=False, voiced_max_dB=140, voiced_min_dB=10)
    for lv in (12, 10):
        clips.append(dict(CAP, spectra=os.path.join(tmp, "captured_l12_throw.bin"), frames=int(cap.shape[0]), bands=int(cap.shape[1]), level=lv, trace=False))
        meta.append(dict(key="captured_l12_throw", settings=CAP, level=lv))
    fcases = feature_cases()
    sargs = score_args()
    job = os.path.join(tmp, "job.json")
    out = os.path.join(tmp, "out.json")
    json.dump({"bundle": BUNDLE, "clips": clips + fcases + [dict(fn="score", args=sargs)]}, open(job, "w"))
    subprocess.run(["node", os.path.join(HERE, "ref_driver.js"), job, out], check=True)
    res = json.load(open(out))
    nclip = len(clips)
    expected = [dict(m, **r) for m, r in zip(meta, res["results"][:nclip])]
    np.savez_compressed(os.path.join(GOLD, "backend_spectra.npz"), **spectra)
    json.dump({"generator": "tests/golden/gen/make_golden.py", "node": res["node"],
               "reference": "formantanalyzer@1.1.6 (dist/main.js module 584)", "cases": expected},
              open(os.path.join(GOLD, "backend_expected.json"), "w"), separators=(",", ":"))
    json.dump({"generator": "tests/golden/gen/make_golden.py", "node": res["node"],
               "cases": [dict(c, expected=r) for c, r in zip(fcases, res["results"][nclip:nclip + len(fcases)])]},
              open(os.path.join(GOLD, "features_expected.json"), "w"), separators=(",", ":"))
    json.dump({"generator": "tests/golden/gen/make_golden.py", "node": res["node"], "reference": "match score `_` of formantanalyzer@1.1.6 (dist/main.js:2 @B37340)",
               "columns": ["gap", "dist", "track_len", "track_bin", "peak_bin", "track_amp", "peak_amp", "velocity"],
               "args": sargs, "expected_f64_hex": res["results"][nclip + len(fcases)]},
              open(os.path.join(GOLD, "score_expected.json"), "w"), separators=(",", ":"))
    subprocess.run(["node", os.path.join(HERE, "make_jsmath.js"), os.path.join(GOLD, "jsmath_v8.json")], check=True)
    for f in ("backend_spectra.npz", "backend_expected.json", "features_expected.json", "score_expected.json", "jsmath_v8.json"):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
