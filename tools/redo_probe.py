#!/usr/bin/env python3
"""Tuning probe (TUNING=1 build, WSA_DBG=16): how many spans the quad / pair tracking kernel declines, and why (peaks per frame / live tracks).
usage: WSA_LIB_DIR=.../lib_tune WSA_DBG=16 [WSA_QUAD=1] tools/redo_probe.py [clips]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import ctypes, os, sys
os.environ.setdefault("WSA_NO_FUSE", "1")          # (the fused compaction clears the batch's counters at the end of a run: the separate kernels leave them readable)
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
fs, ns = 16000, 160000
pcm = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
an = Analyzer(Config(output_level=5))
b = an.batch([ns] * n, fs)
b.enable_trace(True)
b.run(pcm.data_ptr(), pcm.stride(0), 0)
r = b.device_result(0)
c = (ctypes.c_uint32 * 16)()
an.L.wsa_debug_batch_counters(b.h, c)
c = list(c)
print(f"clips {n}: spans {c[5]}, redo list {c[6]}, declined for peaks {c[10]}, for live tracks {c[11]}, rows {r.n_rows}")
