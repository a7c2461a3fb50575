#!/usr/bin/env python3
"""Fixture G4 and the range check of the front-end specification FE-1 — BASELINE config 1.

Build container only: reads /root/reference/samples/263771femaleprotagonist.wav (the file the reference app plays by default,
src/index.js:23 / :291) and /root/reference/dist/nnmodel/1/cats_emotion/model_meta.json (per-feature min / max of the 53 inputs over
the 74 249 real syllables the reference's own model was trained on) AT RUN TIME, and drives the reference's own back end
(formantanalyzer@1.1.6 inside dist/main.js) through ref_driver.js.  Nothing of the reference's source is written anywhere; the
outputs are data:

  tests/golden/config1_excerpt.npz    int16 PCM of a 5 s excerpt of the sample file at its own 44.1 kHz (input)
  tests/golden/config1_expected.json  what the chain  RS-1 (44.1 -> 48 kHz, the reference's offline context rate, ref @B18765)
                                      -> FE-1 (3072-point) -> REFERENCE back end  produced for the excerpt with the app's
                                      settings (src/index.js:21) in Segment Features (5) and Syllable Features (13) mode,
                                      the crc32 of the u32 frames in between, and for the WHOLE file in Syllable Features mode the
                                      range of every one of the 53 features next to the range in model_meta.json.

The front end's source is not in the reference tree (parity unpinned); the range check is the evidence the tree offers for FE-1's
scale choices: a front end whose 4|X|^2 * gain scale were decades off would leave the ranges a real corpus produced.

    python3 tests/golden/gen/make_config1_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile
import wave
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.dirname(HERE)
ROOT = os.path.dirname(os.path.dirname(GOLD))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

WAV = "/root/reference/samples/263771femaleprotagonist.wav"
META = "/root/reference/dist/nnmodel/1/cats_emotion/model_meta.json"
BUNDLE = "/root/reference/dist/main.js"
APP = dict(window_step=15.0, pause_length=200.0, min_seg_length=50.0, auto_noise_gate=True, voiced_max_dB=100.0, voiced_min_dB=10.0)   # src/index.js:21
FS_CTX = 48000.0                                        # new OfflineAudioContext(1, 48e6, 48e3), ref dist/main.js:2 @B18765
EXCERPT = (2.0, 7.0)                                    # seconds of the file


def spectra_of(i16, fs):
    x = (i16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    y = pyoracle.resample(x, fs, FS_CTX)
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=FS_CTX, window_step=APP["window_step"]))
    return fe.run(y), fe


def reference_backend(spec, levels, tmp):
    path = os.path.join(tmp, "spec_%d.bin" % len(os.listdir(tmp)))
    spec.tofile(path)
    clips = [dict(APP, spectra=path, frames=int(spec.shape[0]), bands=int(spec.shape[1]), level=lv, trace=False) for lv in levels]
    job, out = os.path.join(tmp, "job.json"), os.path.join(tmp, "out.json")
    json.dump({"bundle": BUNDLE, "clips": clips}, open(job, "w"))
    subprocess.run(["node", os.path.join(HERE, "ref_driver.js"), job, out], check=True)
    res = json.load(open(out))
    return res["results"], res["node"]


def main():
    w = wave.open(WAV)
    fs = w.getframerate()
    assert w.getnchannels() == 1 and w.getsampwidth() == 2
    i16 = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    tmp = tempfile.mkdtemp(prefix="wsa_cfg1_")
    # ---- excerpt: input + expected
    a, b = int(EXCERPT[0] * fs), int(EXCERPT[1] * fs)
    ex = i16[a:b].copy()
    spec, fe = spectra_of(ex, fs)
    res, node = reference_backend(spec, (5, 13), tmp)
    np.savez_compressed(os.path.join(GOLD, "config1_excerpt.npz"), pcm_i16=ex, fs=np.int32(fs))
    # ---- whole file, Syllable Features: ranges against the trained model's input ranges
    spec_all, _ = spectra_of(i16, fs)
    res_all, _ = reference_backend(spec_all, (13, 5), tmp)
    syl = [f for cb in res_all[0]["callbacks"] for f in cb[3]]
    F = np.array([[float(v) if not isinstance(v, str) else float(v) for v in f] for f in syl], dtype=np.float64)
    meta = json.load(open(META))["inputs"]
    ranges = []
    for k in range(53):
        lo, hi = float(meta[str(k)]["min"]), float(meta[str(k)]["max"])
        col = F[:, k]
        ranges.append(dict(feature=k, ours=[float(np.nanmin(col)), float(np.nanmax(col))], corpus=[lo, hi],
                           outside=int(np.sum((col < lo) | (col > hi)))))
    expected = dict(
        generator="tests/golden/gen/make_config1_golden.py", node=node,
        source="samples/263771femaleprotagonist.wav of the reference (44.1 kHz mono int16, %d samples), excerpt %g .. %g s" % (len(i16), EXCERPT[0], EXCERPT[1]),
        reference="formantanalyzer@1.1.6 (dist/main.js module 584) back end on the u32 frames of RS-1 + FE-1 (oracle/)",
        settings=dict(APP, fs_context=FS_CTX), geometry=dict(nfft=fe.nfft, win=fe.win, hop=fe.hop, kmax=fe.kmax, bands=fe.bands),
        excerpt=dict(frames=int(spec.shape[0]), spectra_crc32=int(zlib.crc32(spec.tobytes())), level5=res[0], level13=res[1]),
        whole_file=dict(frames=int(spec_all.shape[0]), spectra_crc32=int(zlib.crc32(spec_all.tobytes())), segments=len(res_all[1]["segments_ci"]),
                        syllables=int(F.shape[0]), corpus="dist/nnmodel/1/cats_emotion/model_meta.json: 74 249 syllables (details.txt:17)", ranges=ranges,
                        values_outside=int(sum(r["outside"] for r in ranges)), values_total=int(F.size)))
    json.dump(expected, open(os.path.join(GOLD, "config1_expected.json"), "w"), separators=(",", ":"))
    print("excerpt: %d frames, %d segments (level 5), %d callbacks (level 13)" % (spec.shape[0], len(res[0]["segments_ci"]), len(res[1]["callbacks"])))
    print("whole file: %d frames, %d segments, %d syllables; %d of %d feature values outside the corpus ranges" %
          (spec_all.shape[0], len(res_all[1]["segments_ci"]), F.shape[0], expected["whole_file"]["values_outside"], F.size))
    for r in ranges:
        if r["outside"]:
            print("  x%-2d ours %s corpus %s outside %d" % (r["feature"], r["ours"], r["corpus"], r["outside"]))
    for f in ("config1_excerpt.npz", "config1_expected.json"):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
