#!/usr/bin/env python3
"""One-off check: very long clips (minutes) through the batch path against the oracle."""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
from oracle import pyoracle
from tests.util import callbacks_equal
fs, secs, n = 16000, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 3
pcm = synth_clips(n, secs * fs, fs=fs, seed=77, device="cuda")
for level in (5, 13):
    an = Analyzer(Config(output_level=level)); b = an.batch([secs * fs] * n, fs)
    st = torch.cuda.current_stream().cuda_stream
    b.run(pcm.data_ptr(), pcm.stride(0), st); got = b.callbacks(st)
    t0 = time.perf_counter(); b.run(pcm.data_ptr(), pcm.stride(0), st); b.device_result(st); dt = time.perf_counter() - t0
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs)); host = pcm.cpu().numpy(); nseg = 0
    for c in range(n):
        ref = pyoracle.run_backend(fe.run(host[c]), pyoracle.default_cfg(level=level))
        assert ref["segments_ci"] == got[c]["segments_ci"], c
        ok, why = callbacks_equal(level, ref["callbacks"], got[c]["callbacks"], exact=False, tol=1e-4); assert ok, why
        nseg += len(ref["segments_ci"])
    print(f"level {level}: {n} clips x {secs} s ok, {nseg} segments, {dt * 1e3:.1f} ms, info {b.info}")
    b.close(); an.close()
