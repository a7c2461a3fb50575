#!/usr/bin/env python3
"""The reference APPLICATION's settings (src/index.js:21: 25 ms windows every 15 ms, level 13) on the bench batch, steps strictly back to back — for
   rocprofv3 (--kernel-trace --stats: which kernels run and for how long; --pmc FETCH_SIZE / WRITE_SIZE: do overlapping windows re-read PCM from HBM?):
   python3 tools/app_defaults_probe.py [fs=16000] [steps=5] [window_step=15] [level=13]      prints segment-length statistics and the step time"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wstep = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
level = int(sys.argv[4]) if len(sys.argv) > 4 else 13
n, ns = 1024, 10 * fs
pcm = synth_clips(n, ns, fs=fs, seed=3, device="cuda:0")
an = Analyzer(Config(output_level=level, window_step=wstep), device=0)
b = an.batch([ns] * n, fs)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
    b.device_result(s)
t0 = time.perf_counter()
for _ in range(steps):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
    r = b.device_result(s)
dt = (time.perf_counter() - t0) / steps
rows = b.rows(s)
seg = np.asarray(rows["segments"]) if "segments" in rows else None
g = an.geometry(fs)
print(f"{fs} Hz, window {g['win']} every {g['hop']} samples, level {level}: {dt * 1e3:.3f} ms per batch alone, frames {b.info['n_frames_total']}, rows {r.n_rows}, segments {r.n_segments}, "
      f"stage ms {b.stage_ms()}, back-end reruns {b.backend_reruns()}")
if seg is not None and len(seg):
    ln = seg[:, 2]
    print("segment length (frames): mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f; longer than 64: %.1f %%, than 128: %.2f %%" % (ln.mean(), *np.percentile(ln, [50, 90, 99]), ln.max(), 100 * (ln > 64).mean(), 100 * (ln > 128).mean()))
