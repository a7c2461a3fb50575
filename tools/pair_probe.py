#!/usr/bin/env python3
"""Tuning helper: per-pair cycle counts of the paired tracker kernel (TUNING=1 build, WSA_DBG bit 16).
usage (GPU box): WSA_DBG=16 python tools/pair_probe.py"""
import os, sys
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
n_clips, ns, fs = 1024, 160000, 16000
pcm = synth_clips(n_clips, ns, fs=fs, seed=0, device="cuda:0")
an = Analyzer(Config(output_level=5), device=0)
b = an.batch([ns] * n_clips, fs)
b.enable_trace(True)
s = torch.cuda.current_stream().cuda_stream
b.run(pcm.data_ptr(), pcm.stride(0), s)
rows = b.rows(s)
tr = b.trace(s)[:, :12]
k = 0
while k < len(tr) and tr[k, 2] >= 1 and tr[k, 0] > 0:
    k += 1
sp = tr[:k]
print("pairs", k, "rows", len(rows["meta"]))
steps, on = sp[:, 2], sp[:, 3]
print("steps per pair mean %.1f max %.0f; frames with accumulate per pair %.1f; total steps %.0f" % (steps.mean(), steps.max(), on.mean(), steps.sum()))
print("accumulate cycles per pair mean %.0f max %.0f; per active step %.0f" % (sp[:, 0].mean(), sp[:, 0].max(), sp[:, 0].sum() / on.sum()))
print("finalize (both spans) cycles mean %.0f max %.0f" % (sp[:, 1].mean(), sp[:, 1].max()))
print("steps with two track chunks: %.1f %%; pair passes per active step %.2f" % (100 * sp[:, 4].sum() / on.sum(), sp[:, 5].sum() / on.sum()))
ph = sp[:, 7:12].sum(axis=0) / on.sum()
print("cycles per active step: peaks %.0f, compaction %.0f, scoring %.0f, update %.0f, new tracks %.0f" % tuple(ph))
print("stage ms", b.stage_ms())
