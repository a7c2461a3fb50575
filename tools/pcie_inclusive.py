#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch path: wsa_batch_run_host with the clips in pinned host memory."""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
fs, ns, n = 16000, 160000, 1024
host = synth_clips(n, ns, fs=fs, seed=1000, device="cuda").cpu().pin_memory()
clips = [host[i].numpy() for i in range(n)]
an = Analyzer(Config(output_level=5)); b = an.batch([ns] * n, fs)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    b.run_host(clips, st); b.device_result(st)
t0 = time.perf_counter(); K = 5
for _ in range(K):
    b.run_host(clips, st); b.device_result(st)
dt = (time.perf_counter() - t0) / K
print(f"run_host (H2D of {n * ns * 4 / 1e6:.0f} MB from pinned memory + whole path): {dt * 1e3:.2f} ms per batch = {n * 400 / dt:.3e} frames/s")
