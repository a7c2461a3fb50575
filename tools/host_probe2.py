#!/usr/bin/env python3
"""Tuning probe: what the host thread spends per pipelined step — duration of Slot.launch (11 kernel launches + 5 event records through ctypes),
of the blocking part of finish (stream synchronize inside wsa_batch_result) and of its non-blocking rest — and when each happens on a common clock.
usage: tools/host_probe2.py [depth] [steps]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
fs, ns, n = 16000, 160000, 1024
pcm = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
an = Analyzer(Config(output_level=5))
bs = [an.batch([ns] * n, fs) for _ in range(depth)]
ss = [torch.cuda.Stream() for _ in range(depth)]
busy = [False] * depth
log = []
def step(k):
    i = k % depth
    t0 = time.perf_counter()
    if busy[i]:
        ss[i].synchronize()
        t1 = time.perf_counter()
        r = bs[i].device_result(ss[i].cuda_stream); bs[i].stage_ms()
        t2 = time.perf_counter()
    else:
        t1 = t2 = t0
    bs[i].run(pcm.data_ptr(), pcm.stride(0), ss[i].cuda_stream)
    busy[i] = True
    t3 = time.perf_counter()
    log.append((t0, t1 - t0, t2 - t1, t3 - t2))
for k in range(depth + 6): step(k)
torch.cuda.synchronize(); log.clear()
T0 = time.perf_counter()
for k in range(steps): step(k)
torch.cuda.synchronize()
T1 = time.perf_counter()
a = np.array(log)
print(f"depth {depth}: {1e3 * (T1 - T0) / steps:.3f} ms per step")
print("wait (stream sync)   us: p10 %.0f p50 %.0f p90 %.0f" % tuple(np.percentile(a[:, 1] * 1e6, [10, 50, 90])))
print("result + stage_ms    us: p10 %.0f p50 %.0f p90 %.0f" % tuple(np.percentile(a[:, 2] * 1e6, [10, 50, 90])))
print("launch (wsa_batch_run) us: p10 %.0f p50 %.0f p90 %.0f" % tuple(np.percentile(a[:, 3] * 1e6, [10, 50, 90])))
print("first 12 steps: start_us wait result launch")
for r in a[:12]:
    print("  %8.0f %6.0f %6.0f %6.0f" % ((r[0] - T0) * 1e6, r[1] * 1e6, r[2] * 1e6, r[3] * 1e6))
