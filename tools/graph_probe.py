#!/usr/bin/env python3
"""Tuning helper: the pipelined step with every planned batch's run captured into a hipGraph (torch.cuda.CUDAGraph) against plain launches.
usage (GPU box): tools/graph_probe.py [steps]"""
import os, sys, time
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from webspeechanalyzer_amd import Analyzer, Config
from webspeechanalyzer_amd.synth import synth_clips
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fs, ns, n = 16000, 160000, 1024
pcm = synth_clips(n, ns, fs=fs, seed=1000, device="cuda")
an = Analyzer(Config(output_level=5), device=0)
streams = [torch.cuda.Stream() for _ in range(3)]
side = torch.cuda.Stream()
class Slot:
    def __init__(self, st):
        self.b = an.batch([ns] * n, fs); self.st = st; self.ev = torch.cuda.Event(); self.busy = False; self.g = None
    def launch(self, graph):
        if graph and self.g is not None:
            with torch.cuda.stream(self.st):
                self.g.replay()
        else:
            self.b.run(pcm.data_ptr(), pcm.stride(0), self.st.cuda_stream)
        self.ev.record(self.st); self.busy = True
    def finish(self):
        self.ev.synchronize(); r = self.b.device_result(side.cuda_stream); self.busy = False; return r.n_rows
slots = [Slot(streams[j % 3]) for j in range(6)]
def run(k, graph):
    rows = 0
    for i in range(k):
        s = slots[i % 6]
        if s.busy: rows = s.finish()
        s.launch(graph)
    for s in slots:
        if s.busy: rows = s.finish()
    return rows
run(12, False); torch.cuda.synchronize()
for s in slots:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s.st):
        s.b.run(pcm.data_ptr(), pcm.stride(0), s.st.cuda_stream)
    s.g = g
torch.cuda.synchronize()
for graph in (False, True, False, True):
    run(12, graph); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); rows = run(steps, graph); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps * 1e3)
    ts.sort()
    print(f"graph={graph}: median {ts[2]:.4f} ms per step (min {ts[0]:.4f}, max {ts[-1]:.4f}), rows {rows}")
