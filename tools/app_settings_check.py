#!/usr/bin/env python3
"""One-off check at the reference APPLICATION's settings (src/index.js:21: 25 ms windows every 15 ms; 48 kHz as the offline path runs it): the batch path against the
oracle on clips of a minute — long segments through all three forms of the out-of-LDS finalize and the generic path.  usage: python tools/app_settings_check.py [seconds]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
from oracle import pyoracle
from tests.util import callbacks_equal
secs, n = int(sys.argv[1]) if len(sys.argv) > 1 else 60, 4
for fs in (16000, 48000):
    pcm = synth_clips(n, secs * fs, fs=fs, seed=91, device="cuda")
    host = pcm.cpu().numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs, window_step=15.0))
    spectra = [fe.run(host[c]) for c in range(n)]
    for level in (5, 13, 10):
        an = Analyzer(Config(output_level=level, window_step=15.0)); b = an.batch([secs * fs] * n, fs)
        st = torch.cuda.current_stream().cuda_stream
        b.run(pcm.data_ptr(), pcm.stride(0), st); got = b.callbacks(st)
        lens = []
        for c in range(n):
            ref = pyoracle.run_backend(spectra[c], pyoracle.default_cfg(level=level, window_step=15.0))
            assert ref["segments_ci"] == got[c]["segments_ci"], (fs, level, c)
            ok, why = callbacks_equal(level, ref["callbacks"], got[c]["callbacks"], exact=False, tol=1e-4); assert ok, (fs, level, c, why)
            lens += [s[1] for s in ref["segments_ci"]]
        lens = np.array(lens)
        print(f"{fs} Hz level {level}: {n} clips x {secs} s ok, {len(lens)} segments, length max {lens.max()}, over 100 frames {(lens > 100).sum()}, over 128 {(lens > 128).sum()}")
        b.close(); an.close()
