#!/usr/bin/env python3
"""Tuning probe for the kernel timeline: D streams, whole path, stage-timing events on or off, 40 steps (run under rocprofv3 --kernel-trace).
usage: tools/timeline_probe.py [depth] [timing 0|1] [sync: stream|event|none]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
timing = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "stream"
fs, ns, n = 16000, 160000, 1024
pcm = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
an = Analyzer(Config(output_level=5))
bs = [an.batch([ns] * n, fs) for _ in range(depth)]
ss = [torch.cuda.Stream() for _ in range(depth)]
for b in bs:
    b.enable_timing(bool(timing))

def loop(K):
    for k in range(K):
        i = k % depth
        if mode == "stream":
            ss[i].synchronize()
        bs[i].run(pcm.data_ptr(), pcm.stride(0), ss[i].cuda_stream)
    torch.cuda.synchronize()
loop(6)
t0 = time.perf_counter(); K = 40; loop(K); dt = time.perf_counter() - t0
print(f"depth {depth} timing {timing} sync {mode}: {dt / K * 1e3:.3f} ms/step")
