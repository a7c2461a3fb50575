#!/usr/bin/env python3
"""Tuning helper: per-span cycle counts of the tracker (WSA_DBG bit 16 writes them into the trace buffer).
usage (GPU box): WSA_DBG=16 python tools/span_probe.py   (add bits 1 / 2 to switch finalize / accumulate off)"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
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
for _ in range(3):
    b.run(pcm.data_ptr(), pcm.stride(0), s)
rows = b.rows(s)
tr = b.trace(s)
n_spans = None
# spans are the rows whose slot 3 (frames of the span) is a positive integer and slot 6 a block id; take the leading run
cand = tr[:, :12]
k = 0
while k < len(cand) and cand[k, 3] >= 1 and cand[k, 0] > 0:
    k += 1
sp = cand[:k]
print("spans", k, "rows", len(rows["meta"]))
fr = sp[:, 3]; cyc_t = sp[:, 0]; cyc_f = sp[:, 1]
print("frames per span: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f  total %.0f" % (fr.mean(), *np.percentile(fr, [50, 90, 99]), fr.max(), fr.sum()))
print("seg len        : mean %.1f max %.0f" % (sp[:, 2].mean(), sp[:, 2].max()))
print("track cycles/span: mean %.0f p50 %.0f p99 %.0f max %.0f ; per frame %.0f" % (cyc_t.mean(), *np.percentile(cyc_t, [50, 99]), cyc_t.max(), cyc_t.sum() / fr.sum()))
print("finalize cycles  : mean %.0f p50 %.0f p99 %.0f max %.0f" % (cyc_f.mean(), *np.percentile(cyc_f, [50, 99]), cyc_f.max()))
if int(os.environ.get("WSA_DBG", "0")) & 512:
    print("accumulate phases (mean cycles per span): accept %.0f, retire %.0f, score %.0f, hand+update %.0f, new %.0f" % tuple(sp[:, 7:12].mean(axis=0)))
else:
  print("finalize phases (mean cycles): rank+keys %.0f, straighten %.0f, copy+rows %.0f, features %.0f, rest %.0f" % (sp[:, 7].mean(), sp[:, 8].mean(), sp[:, 9].mean(), sp[:, 10].mean(), (cyc_f - sp[:, 7:11].sum(axis=1)).mean()))
print("tracks/span mean %.1f max %.0f points mean %.1f max %.0f" % (sp[:, 4].mean(), sp[:, 4].max(), sp[:, 5].mean(), sp[:, 5].max()))
blk = sp[:, 6].astype(int)
per = np.bincount(blk, weights=cyc_t + cyc_f)
cnt = np.bincount(blk)
print("waves used %d; spans per wave mean %.2f max %d; busy cycles per wave mean %.0f max %.0f" % ((cnt > 0).sum(), cnt[cnt > 0].mean(), cnt.max(), per[cnt > 0].mean(), per.max()))
print("stage ms", b.stage_ms())
if len(sys.argv) > 1:
    np.save(sys.argv[1], sp)          # per-span rows (cycles, frames, ...) for offline what-if scheduling
