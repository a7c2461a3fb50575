#!/usr/bin/env python3
"""Tuning helper: time of the rate converter K0 in front of the path (1024 clips x 10 s -> 48 kHz), per-kernel via rocprofv3:
   rocprofv3 --kernel-trace --stats -d /tmp/p -o r -- python3 tools/resample_probe.py [fs_in]; tools/rocprof_summary.py /tmp/p/.../r_results.db"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

fs_in = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n, ns = 1024, 10 * fs_in
pcm = synth_clips(n, ns, fs=fs_in, seed=1, device="cuda:0")
an = Analyzer(Config(output_level=5), device=0)
b = an.batch([ns] * n, fs_in, resample_to=48000)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
    b.device_result(s)
t0 = time.perf_counter()
for _ in range(5):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
    r = b.device_result(s)
dt = (time.perf_counter() - t0) / 5
print(f"{fs_in} -> 48000 Hz: {dt * 1e3:.3f} ms per batch, rows {r.n_rows}, stage ms {b.stage_ms()}")
