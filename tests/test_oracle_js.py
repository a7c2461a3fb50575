"""The plain-JS oracle (oracle/js/wsa_oracle.js: the whole path under Node, no GPU — BASELINE config
"wav_file_segment_features") against the C oracle (front end: u32 frames bit-exact) and against the
reference's own outputs (tests/golden/backend_expected.json: indices, timestamps and features bit-exact,
since Math.pow / Math.log10 are the engine's own there)."""
import json
import os
import shutil
import subprocess
import wave

import numpy as np
import pytest

from oracle import pyoracle
from tests import util

NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")
RUN = os.path.join(util.ROOT, "oracle", "js", "run.js")


def node(job, tmp_path):
    jf = tmp_path / "job.json"
    jf.write_text(json.dumps(job))
    r = subprocess.run([NODE, RUN, str(jf)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def speechlike(n, fs, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    f0 = 110 + 40 * rng.random()
    x = np.zeros(n)
    for fc, bw in ((500 + 400 * rng.random(), 80), (1500 + 600 * rng.random(), 120), (2500 + 500 * rng.random(), 160)):
        for h in range(1, int(3800 / f0)):
            x += np.exp(-((h * f0 - fc) / bw) ** 2) * np.sin(2 * np.pi * h * f0 * t + rng.random() * 6.28)
    env = (np.sin(2 * np.pi * (1.5 + rng.random()) * t + rng.random() * 6.28) > -0.2).astype(np.float64)
    env = np.convolve(env, np.ones(int(fs * 0.02)) / int(fs * 0.02), mode="same")
    x = 0.25 * x / np.abs(x).max() * env + 1e-4 * rng.standard_normal(n)
    return x.astype(np.float32)


@pytest.mark.parametrize("fs,kw", [(16000, {}), (16000, {"spec_type": 2}), (16000, {"spec_type": 3, "high_f_emph": 0.01}),
                                   (44100, {}), (48000, {"window_step": 15.0}), (8000, {}), (22050, {}), (11025, {"window_width": 40.0}), (6000, {}), (16000, {"N_mel_bins": 64, "f_max": 3000.0})])
def test_js_front_end_matches_c_oracle_bit_for_bit(tmp_path, fs, kw):
    pcm = speechlike(int(fs * 1.2), fs, 7)
    pf = tmp_path / "pcm.f32"
    pcm.tofile(pf)
    out = tmp_path / "spec.u32"
    g = node({"mode": "fe", "pcm": str(pf), "fs": fs, "settings": kw, "out": str(out)}, tmp_path)
    ckw = {k.replace("N_", "n_"): v for k, v in kw.items()}
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=float(fs), **ckw))
    assert (g["nfft"], g["win"], g["hop"], g["bands"], g["kmax"]) == (fe.nfft, fe.win, fe.hop, fe.bands, fe.kmax)
    ref = fe.run(pcm)
    got = np.fromfile(out, dtype=np.uint32).reshape(ref.shape)
    assert ref.any()
    assert np.array_equal(ref, got), f"{(ref != got).sum()} of {ref.size} words differ"
    assert np.allclose(np.array(g["bins_hz"]), fe.bins_hz(), rtol=1e-12, atol=0)      # libm vs V8 pow in the band centres


def _spectra_for(spectra, case):
    return np.ascontiguousarray(spectra[case["key"]], dtype=np.uint32)


def test_js_back_end_reproduces_reference_callbacks(tmp_path):
    spectra, cases = util.load_backend_golden()
    done = 0
    for case in cases:
        if case["level"] not in (3, 5, 11, 12, 13):
            continue
        sp = _spectra_for(spectra, case)
        sf = tmp_path / "s.u32"
        sp.tofile(sf)
        cfg = dict(case["settings"], level=case["level"], bands=int(sp.shape[1]))
        g = node({"mode": "be", "spectra": str(sf), "frames": int(sp.shape[0]), "cfg": cfg}, tmp_path)
        assert g["segments_ci"] == case["segments_ci"], case["key"]
        flat = case["level"] in (5, 11)
        if case["level"] == 3:
            got = g["callbacks"]
        else:
            got = [[c[0], c[1], (np.array(c[2]) if flat else c[2]),
                    (util.jsvec(c[3]) if flat else [util.jsvec(v) for v in c[3]])] for c in g["callbacks"]]
        ok, why = util.callbacks_equal(case["level"], case["callbacks"], got, exact=True)
        assert ok, f"{case['key']}: {why}"
        done += 1
    assert done >= 4


def test_js_whole_path_on_a_wav_file_matches_c_oracle(tmp_path):
    """BASELINE config 'wav_file_segment_features': a 44.1 kHz WAV through the pure-JS CPU path."""
    from webspeechanalyzer_amd.synth import synth_clips
    fs = 44100
    pcm = synth_clips(1, fs * 6, fs=fs, seed=5)[0].numpy()
    i16 = np.round(pcm * 32767).astype(np.int16)
    wf = tmp_path / "clip.wav"
    with wave.open(str(wf), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs); w.writeframes(i16.tobytes())
    g = node({"mode": "e2e", "wav": str(wf), "settings": {"output_level": 5}}, tmp_path)
    assert g["fs"] == fs and g["nfft"] == 3072
    x = (i16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=float(fs)))
    o = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=5))
    assert g["segments_ci"] == o["segments_ci"] and len(o["segments_ci"]) > 0
    assert len(g["callbacks"]) == len(o["callbacks"]) > 0
    for a, b in zip(g["callbacks"], o["callbacks"]):
        assert a[0] == b[0] and util.same_f64(np.array(a[2]), np.array(b[2]))
        assert util.same_f64(util.jsvec(a[3]), b[3])          # C oracle's V8 math ports == the engine's own


def test_feature_db_files_match_the_reference_app_byte_for_byte():
    """the JSON / CSV feature-DB files (and the import of one) written by
    webspeechanalyzer_amd/js/featuredb.js equal what the reference app's own localstore.js / call_backed wrote for the
    same callbacks and imported files (tests/golden/featuredb_expected.json; levels 5, 13, 12, 11, 10, re-stored and refused
    samples, labeled files through import and both exports, the CSV writer's TypeError, invalid and empty inputs)."""
    r = subprocess.run([NODE, os.path.join(util.ROOT, "tests", "js", "featuredb_check.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout)
    assert res["checked"] >= 25 and res["mismatches"] == []


def test_js_match_score_equals_the_reference_function():
    """G2 for the JS restatement: `matchScore` on the rows of tests/golden/score_expected.json (the reference's own `_` under Node), bit for bit."""
    prog = ("const o=require(process.argv[1]);const d=require(process.argv[2]);let bad=0;const b=Buffer.alloc(8);"
            "d.args.forEach((a,i)=>{b.writeDoubleBE(o.matchScore.apply(null,a));if(b.toString('hex')!==d.expected_f64_hex[i])bad++;});"
            "process.stdout.write(JSON.stringify({bad:bad,n:d.args.length}));")
    r = subprocess.run([NODE, "-e", prog, os.path.join(util.ROOT, "oracle", "js", "wsa_oracle.js"), os.path.join(util.GOLDEN, "score_expected.json")],
                       capture_output=True, text=True, check=True)
    out = json.loads(r.stdout)
    assert out["n"] > 5000 and out["bad"] == 0
