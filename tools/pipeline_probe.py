#!/usr/bin/env python3
"""Tuning probe: steady-state ms per step of parts of the pipeline with D batches in flight (one HIP stream each).
usage: tools/pipeline_probe.py [depth]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fs, ns, n = 16000, 160000, 1024
pcm = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
for name, level, fe_only in (("front end only", 5, True), ("front end + peaks + gate (level 3)", 3, False), ("whole path (level 5)", 5, False)):
    an = Analyzer(Config(output_level=level))
    bs = [an.batch([ns] * n, fs) for _ in range(depth)]
    ss = [torch.cuda.Stream() for _ in range(depth)]
    for b in bs:
        b.enable_timing(False)

    def loop(K):
        for k in range(K):
            i = k % depth
            ss[i].synchronize()
            (bs[i].run_frontend if fe_only else bs[i].run)(pcm.data_ptr(), pcm.stride(0), ss[i].cuda_stream)
        torch.cuda.synchronize()
    loop(6)
    t0 = time.perf_counter(); K = 30; loop(K); dt = time.perf_counter() - t0
    print(f"{name:40s} depth {depth}: {dt / K * 1e3:.3f} ms/step")
    for b in bs:
        b.close()
    an.close()
