#!/usr/bin/env python3
"""Tuning helper (TUNING=1 build, WSA_DBG=16): per-span finalize cycle counts and phases at given settings — which spans are slow, and where.
usage (GPU box): WSA_LIB_DIR=.../lib_tune WSA_DBG=16 python tools/fin_probe.py [fs=16000] [window_step=15] [level=13]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
wstep = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
level = int(sys.argv[3]) if len(sys.argv) > 3 else 13
n, ns = 1024, 10 * fs
pcm = synth_clips(n, ns, fs=fs, seed=3, device="cuda:0")
an = Analyzer(Config(output_level=level, window_step=wstep), device=0)
b = an.batch([ns] * n, fs)
b.enable_trace(True)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
rows = b.rows(s)
tr = b.trace(s)[:, :12]
k = 0
while k < len(tr) and (tr[k, 0] > 0 or tr[k, 1] > 0):
    k += 1
sp = tr[:k]
print("trace rows", k, "result rows", len(rows["meta"]), "stage ms", b.stage_ms())
fin = sp[sp[:, 0] == 0]          # finalize-only rows (the split finalize kernel and the tail's long-span role write tk1 == tk0)
print("finalize rows", len(fin), "cycles mean %.0f p99 %.0f max %.0f" % (fin[:, 1].mean(), np.percentile(fin[:, 1], 99), fin[:, 1].max()))
order = np.argsort(-fin[:, 1])[:12]
print("slowest: cycles, len, frames, tracks, points, wave, phases (rank+keys, straighten, copy+rows, features)")
for i in order:
    r = fin[i]
    print("%9.0f  len %4.0f frames %4.0f tr %4.0f pt %5.0f wave %5.0f  ph %8.0f %8.0f %8.0f %8.0f" % (r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]))
lng = fin[fin[:, 2] > 128]
if len(lng):
    print("spans of more than 128 frames: %d, finalize cycles mean %.0f max %.0f" % (len(lng), lng[:, 1].mean(), lng[:, 1].max()))
