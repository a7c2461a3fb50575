#!/usr/bin/env python3
"""Tuning probe: does a high-priority HIP stream for the front end shorten its in-pipeline duration?
Three batches in flight; per slot the front end goes to `fe_stream` (priority -1 or 0), the back end to the slot's
own stream behind an event.  Prints ms per step and the front end's mean duration (events on its stream)."""
import os, sys, time
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
n_clips, ns, fs = 1024, 160000, 16000
pcm = synth_clips(n_clips, ns, fs=fs, seed=0, device="cuda:0")
an = Analyzer(Config(output_level=5), device=0)
for mode in ("same-stream", "split prio 0", "split prio -1", "one shared fe stream prio -1"):
    depth = 3
    batches = [an.batch([ns] * n_clips, fs) for _ in range(depth)]
    be = [torch.cuda.Stream() for _ in range(depth)]
    if mode == "split prio 0": fe = [torch.cuda.Stream(priority=0) for _ in range(depth)]
    elif mode == "split prio -1": fe = [torch.cuda.Stream(priority=-1) for _ in range(depth)]
    elif mode.startswith("one shared"): fe = [torch.cuda.Stream(priority=-1)] * depth
    else: fe = be
    for b, s in zip(batches, be):
        b.run(pcm.data_ptr(), pcm.stride(0), s.cuda_stream); spec = b.device_result(s.cuda_stream).d_spectra
    specs = [b.device_result(s.cuda_stream).d_spectra for b, s in zip(batches, be)]
    fe_ms = []
    def launch(k, timed):
        b, sf, sb = batches[k % depth], fe[k % depth], be[k % depth]
        if mode == "same-stream":
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(sb); b.run_frontend(pcm.data_ptr(), pcm.stride(0), sb.cuda_stream); e1.record(sb)
            b.run_backend(specs[k % depth], sb.cuda_stream)
        else:
            sf.wait_stream(sb)                                  # the slot's previous back end has read the spectra
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(sf); b.run_frontend(pcm.data_ptr(), pcm.stride(0), sf.cuda_stream); e1.record(sf)
            sb.wait_event(e1)
            b.run_backend(specs[k % depth], sb.cuda_stream)
        if timed: fe_ms.append((e0, e1))
    for k in range(6): launch(k, False)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for k in range(K):
        if k >= depth: be[k % depth].synchronize()
        launch(k, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    print(f"{mode:32s} {dt:.3f} ms/step   front end {np.mean([a.elapsed_time(b) for a, b in fe_ms]):.3f} ms")
    for b in batches: b.close()
