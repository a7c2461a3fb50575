#!/usr/bin/env python3
"""Time of the peak-candidate scan alone on the bench spectra (1024 clips x 10 s), optionally with parts switched off
(TUNING=1 build: dbg 1 no emission, 2 no state machine, 4 no mask pass), in rounds of 32 and of 16 bins.  usage: tools/peaks_probe.py [dbg ...]"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webspeechanalyzer_amd import Analyzer, Config, capi
from webspeechanalyzer_amd.synth import synth_clips
fs, ns, n_clips = 16000, 160000, 1024
pcm = synth_clips(n_clips, ns, fs=fs, seed=1000, device="cuda")
an = Analyzer(Config(output_level=5), device=0)
b = an.batch([ns] * n_clips, fs); b.keep_spectra(True)
st = torch.cuda.current_stream().cuda_stream
b.run(pcm.data_ptr(), pcm.stride(0), st)
spec, foff = b.spectra(st)
spec = np.ascontiguousarray(spec)
L = capi.lib()
L.wsa_debug_peaks_time.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
L.wsa_debug_peaks_time.restype = ctypes.c_int
for dbg in [int(x) for x in sys.argv[1:]] or [0]:
    for mode, name in ((4, "rounds of 32 bins"), (3, "rounds of 16 bins")):
        ms = ctypes.c_float(0)
        assert L.wsa_debug_peaks_time(0, spec.ctypes.data, spec.shape[0], spec.shape[1], mode, dbg, 20, ctypes.byref(ms)) == 0
        print(f"dbg {dbg}, {name}: {ms.value * 1000:.1f} us per launch ({spec.shape[0]} frames)")
