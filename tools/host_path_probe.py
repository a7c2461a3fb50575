#!/usr/bin/env python3
"""Where the time of a host-memory batch goes: plan creation, H2D from pageable / pinned memory, compute, row copies."""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
fs, ns, n = 16000, 160000, 1024
dev = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
pageable = dev.cpu().numpy()
pinned = dev.cpu().pin_memory().numpy()
an = Analyzer(Config(output_level=5))
st = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter(); b = an.batch([ns] * n, fs); t1 = time.perf_counter()
print(f"wsa_batch_create: {(t1 - t0) * 1e3:.1f} ms")
for name, host in (("pageable", pageable), ("pinned", pinned)):
    clips = [host[i] for i in range(n)]
    b.run_host(clips, st); b.device_result(st)
    t0 = time.perf_counter(); b.run_host(clips, st); b.device_result(st); t1 = time.perf_counter()
    print(f"run_host from {name} memory: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); r = b.rows(st); t1 = time.perf_counter()
print(f"copy rows ({len(r['meta'])}): {(t1 - t0) * 1e3:.2f} ms")
t0 = time.perf_counter(); b.close(); t1 = time.perf_counter()
print(f"wsa_batch_destroy: {(t1 - t0) * 1e3:.1f} ms")
