#!/usr/bin/env python3
"""BASELINE config "streaming": N concurrent real-time mono streams, one hipGraph-captured step per
frame period; reports p50 / p99 step latency (samples in pinned host memory -> this step's feature rows
visible to the host), one JSON line.

    python bench_stream.py [--streams 512] [--fs 48000] [--frames-per-step 1] [--steps 10000] [--level 5]

A step = wsa_stream_step_host + wsa_stream_collect: pull of the new samples over PCIe, front end (3072-point
FFT at 48 kHz), peak candidates, gate state machines, tracker + finalize of the segments that closed,
compaction, push of the rows to the host — a single graph launch of kernels (stream_api.hip).  Filling the
pinned input buffer (the audio "arriving") is outside the timed region."""
import argparse
import ctypes
import json
import sys
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--frames-per-step", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=20.0, help="length of the synthetic signal each stream loops over")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        print("bench_stream.py needs a GPU (libwsa has no CPU path)", file=sys.stderr)
        sys.exit(2)
    import webspeechanalyzer_amd as wsa
    from webspeechanalyzer_amd import capi
    from webspeechanalyzer_amd.synth import synth_clips

    n, fs, F = args.streams, args.fs, args.frames_per_step
    an = wsa.Analyzer(wsa.Config(output_level=args.level))
    g = an.geometry(fs)
    st = an.streams(n, fs, frames_per_step=F, max_span_frames=1024)
    st.enable_graph(not args.no_graph)
    sps = st.samples_per_step
    loop_steps = max(1, int(args.seconds * fs) // sps)
    pcm = synth_clips(n, loop_steps * sps, fs=fs, seed=5, device="cuda").cpu().numpy().reshape(n, loop_steps, sps)
    hin = st.host_input()
    L, h = st.L, st.h
    rows_out = capi._StreamRows()
    lat = np.zeros(args.steps)
    rows = segs = 0
    t_all0 = None
    for k in range(-args.warmup, args.steps):
        hin[:] = pcm[:, (k + args.warmup) % loop_steps, :]
        t0 = time.perf_counter()
        rc = L.wsa_stream_step_host(h, None, None)
        rc2 = L.wsa_stream_collect(h, None, ctypes.byref(rows_out))
        t1 = time.perf_counter()
        if rc or rc2:
            raise RuntimeError(L.wsa_last_error(an.h).decode())
        if k == 0:
            t_all0 = t0
        if k >= 0:
            lat[k] = t1 - t0
            rows += rows_out.n_rows
            segs += rows_out.n_segments
    wall = time.perf_counter() - t_all0
    ms = lat * 1e3
    period_ms = 1e3 * sps / fs
    out = {
        "metric": "p99 frame latency", "value": float(np.percentile(ms, 99)), "unit": "ms", "higher_is_better": False,
        "p50_ms": float(np.percentile(ms, 50)), "mean_ms": float(ms.mean()), "max_ms": float(ms.max()),
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "data": "synthetic",
        "dtype": "f32 front end / f64 tracker",
        "config": {"workload": f"{n} concurrent {fs} Hz mono streams, {F} frame(s) per step ({period_ms:g} ms of audio), "
                               f"{g['nfft']}-pt FFT, level {args.level}, hipGraph step = {'on' if not args.no_graph else 'off'}",
                   "samples_per_step_per_stream": sps, "h2d_bytes_per_step": int(n * sps * 4)},
        "real_time_budget_ms": period_ms, "budget_used_p99": float(np.percentile(ms, 99) / period_ms),
        "frames_per_s_sustained": float(n * F * args.steps / lat.sum()),
        "streams_in_real_time_at_p99": int(n * period_ms / np.percentile(ms, 99)),
        "rows": int(rows), "segments": int(segs), "loop_wall_s": wall,
    }
    print(json.dumps(out))
    st.close(); an.close()


if __name__ == "__main__":
    main()
