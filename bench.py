#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on synthetic 16 kHz mono PCM.

A "step" = one pass of the whole pipeline (PCM -> Hann -> FFT -> mel -> u32 -> peak scan -> gate -> tracker
-> 53-feature rows) over one batch per GPU, resident in HBM before the timed region.
  N = 1   the batch is BASELINE.json configs[1]: 1024 clips x 10 s, 1024-pt FFT, 25 ms hop, Segment Features.
  N > 1   BASELINE.json configs[3]: every rank runs its shard of the 100 000-clip job (12 500 clips x 10 s per GPU, weak
          scaling) and the feature matrices are gathered to rank 0 with one RCCL gather per step.
Steps are software-pipelined over S = --in-flight HIP streams (default 3) with --slots-per-stream planned batches each (default 2 at
N = 1): step k runs on slot k % (number of slots), slot j on stream j % S with its own planned batch, so the back end of one step
overlaps the front end of the next, and a stream's next step is already queued while the host reads the finished step's counters —
every step still is one full pass over one batch whose counters the host reads before the slot is used again, and `value` = the K
steps' frames over the wall time between the two synchronisation points.  `roofline` is the whole step against the HBM roofline (algorithmic bytes per step / ms_per_step);
`roofline.dominant_kernel` names the largest kernel alone and pipelined with its stage times: alone = HIP events between the stages of
three steps strictly back to back just before the timed region (`single_batch`; what a rocprofv3 profile of `bench.py --in-flight 1`
shows, profiles/*_kernel_stats_in_flight_1.txt), pipelined = the same events inside the timed region, where kernels share the GPU.

The same JSON line carries (rank 0, N = 1): `extra.level13` = BASELINE configs[2] (Syllable Features on the same batch),
`extra.streaming` = configs[4] (512 x 48 kHz streams, one hipGraph step per 25 ms frame, p50 / p99 timed inside libwsa),
`extra.offline_48k` = the headline batch at the reference's offline context rate (48 kHz, 3072-point FFT),
`cpu_baseline` with `cpu_parity` (the CPU rows of the sampled clips compared with the GPU's rows of the same clips).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level 5|13] [--clips C] [--no-cpu-baseline] [--no-extra]
With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a child process, before anything touches a GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s float4 copy)
LEVEL_NAME = {5: "Segment Features (level 5)", 13: "Syllable Features (level 13)", 10: "Syllable Formants (level 10)",
              11: "Utterance Features (level 11)", 12: "Syllable Polynomials (level 12)"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=9,
                    help="the timed region (W warm-up + K timed steps between two synchronisation points) is run this many times in one invocation; "
                         "ms_per_step / value are the MEDIAN region, min / max are reported beside it (the first one or two regions of an invocation run "
                         "~5 %% slower — the GPU has been idle through the set-up —, so the median wants more than five of them)")
    ap.add_argument("--level", type=int, default=5, choices=(5, 13, 10, 11, 12))
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default: 1024 at N = 1, the 12 500-clip shard of BASELINE config 4 at N > 1)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--fs", type=int, default=16000, help="sample rate of the synthetic clips (BASELINE configs use 16000)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="HIP streams the steps are dealt over (default 3): step k runs on slot k %% (number of slots), slot j on stream j %% S; "
                         "1 = strictly back to back (what profiles/*_kernel_stats_in_flight_1.txt is taken with)")
    ap.add_argument("--slots-per-stream", type=int, default=0,
                    help="planned batches per stream (default 2 at N = 1, 1 at N > 1): with 2 a stream's next step is already queued behind the running one while "
                         "the host reads the finished step's counters, so no stream waits for the host's round trip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the level-13 and streaming blocks")
    ap.add_argument("--cpu-clips", type=int, default=768)
    ap.add_argument("--master-port", type=int, default=29533)
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child job and pass its exit code on (this process never touches a GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    # test hook: WSA_BENCH_BACKEND=gloo runs the N > 1 code path with several ranks on ONE GPU (no RCCL, numbers meaningless)
    backend = os.environ.get("WSA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend, device_id=dev) if backend == "nccl" else dist.init_process_group(backend)

    from webspeechanalyzer_amd import Analyzer, Config
    from webspeechanalyzer_amd.gather import gather_rows
    from webspeechanalyzer_amd.synth import synth_clips

    fs = args.fs
    ns = int(args.seconds * fs)
    n_clips = args.clips or (1024 if world == 1 else 12500)
    n_streams = max(1, args.in_flight or 3)
    spp = max(1, args.slots_per_stream or (2 if world == 1 and n_streams > 1 else 1))
    depth = n_streams * spp                                   # planned batches (slots); slot j runs on stream j % n_streams
    pcm = synth_clips(n_clips, ns, fs=fs, seed=1000 + rank, device=dev)          # HBM resident before timing
    an = Analyzer(Config(output_level=args.level), device=local_rank)
    geo = an.geometry(fs)

    side = torch.cuda.Stream(device=dev)                      # never has work: what wsa_batch_result synchronises when the slot's own event has been waited for
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

    class Slot:
        def __init__(self, analyzer, stream=None):
            self.batch = analyzer.batch([ns] * n_clips, fs)
            self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
            self.done = torch.cuda.Event()
            self.rows_cap = self.batch.info["rows_cap"]
            # gather buffers (rank 0 receives): features and metadata keep their own dtypes
            self.feat = torch.empty((self.rows_cap, 53), dtype=torch.float64, device=dev) if world > 1 else None
            self.meta = torch.empty((self.rows_cap, 8), dtype=torch.int32, device=dev) if world > 1 else None
            # rank 0 receives every peer's rows straight into these tables (exact-size sends, webspeechanalyzer_amd/gather.py)
            self.recv = ((torch.empty((self.rows_cap * world, 8), dtype=torch.int32, device=dev), torch.empty((self.rows_cap * world, 53), dtype=torch.float64, device=dev))
                         if world > 1 and rank == 0 else None)
            self.busy = False
            self.gathered = 0

        def launch(self):
            self.batch.run(pcm.data_ptr(), pcm.stride(0), self.stream.cuda_stream)
            self.done.record(self.stream)
            self.busy = True

        def finish(self):
            """Wait for this slot's step, read its row counters; multi-GPU: the single exchange of the job — feature
            matrices to rank 0 over RCCL (xGMI), SURVEY.md 8e."""
            b, st = self.batch, self.stream.cuda_stream
            self.done.synchronize()                          # this slot's step is through (the stream may already hold the next slot's step)
            r = b.device_result(side.cuda_stream if world == 1 else st)
            if world > 1:
                b.an._check(b.L.wsa_batch_copy_rows(b.h, st, self.meta.data_ptr(), self.feat.data_ptr(), self.rows_cap, None, 0, None, None))
                with torch.cuda.stream(self.stream):
                    g = gather_rows(self.meta, self.feat, r.n_rows, rank * n_clips, out=self.recv)
                if rank == 0 and g[0] is not None:
                    self.gathered = int(g[0].shape[0])
            self.busy = False
            return r.n_rows, b.stage_ms()

    def run_steps(slots, k_steps, d, stamps=None):
        """k_steps steps, slot k % d each; returns (rows of the last finished step, summed stage ms)."""
        stage = np.zeros(4)
        rows = 0
        for k in range(k_steps):
            sl = slots[k % d]
            if sl.busy:
                rows, ms = sl.finish()
                stage += ms
                if stamps is not None:
                    stamps.append(time.perf_counter())
            sl.launch()
        for j in range(d):                                   # drain in launch order
            sl = slots[(k_steps + j) % d]
            if sl.busy:
                rows, ms = sl.finish()
                stage += ms
                if stamps is not None:
                    stamps.append(time.perf_counter())
        return rows, stage

    slots = [Slot(an, streams[j % n_streams]) for j in range(depth)]
    frames = slots[0].batch.info["n_frames_total"]
    # set-up, not warm-up: every slot's buffers are touched once (a planned batch's device memory is mapped on first use), so that
    # with --warmup smaller than the pipeline depth no slot's first pass falls into the timed region
    run_steps(slots, depth, depth)
    torch.cuda.synchronize()
    run_steps(slots, args.warmup, depth)
    # strictly back to back (no overlap, warm): per-kernel times of kernels that have the GPU to themselves
    solo_ms = np.zeros(4)
    solo_wall = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        slots[0].launch()
        _, ms1 = slots[0].finish()
        solo_wall += time.perf_counter() - t_s
        solo_ms += ms1
    solo_ms /= 3
    solo_wall /= 3
    # the timed region, `repeats` times: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over
    # ranks; a single 100-step region lasts ~65 ms and two of them differ by a per cent or two, so the line reports the median region
    regions = []
    for rep_i in range(max(1, args.repeats)):
        if rep_i:
            run_steps(slots, args.warmup, depth)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        stamps_r = []
        t0 = time.perf_counter()
        rows_r, stage_r = run_steps(slots, args.steps, depth, stamps_r)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_r = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_r], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_r = float(tmax.item())
        regions.append((dt_r, rows_r, stage_r, stamps_r))
    order_r = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, rows, stage, stamps = regions[order_r[len(order_r) // 2]]          # the median region (the upper one of an even count)
    region_ms = [r[0] / max(args.steps, 1) * 1e3 for r in regions]
    stage = stage / max(args.steps, 1)
    reruns = sum(s.batch.backend_reruns() for s in slots)

    def copy_ceiling():
        """measured device-to-device copy rate on this box (bytes read + bytes written per second), the practical HBM ceiling"""
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)        # 1 GiB
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record(); e1.synchronize()
        return 2 * a.numel() * 4 * 5 / (e0.elapsed_time(e1) / 1e3) / 1e9

    if rank == 0:
        fe_name = "fe_kernel_r8" if geo["nfft"] == 1024 else ("fe_kernel_r3" if geo["nfft"] % 3 == 0 else "fe_kernel_rx")
        traffic, traffic_per_kernel, traffic_src = pmc_traffic(n_clips, fs, args.level, args.seconds)
        total_frames = frames * world * args.steps
        value = total_frames / dt
        # roofline of the pipeline step: algorithmic bytes per step = 4 * hop samples per frame (PCM read once) + the 53-feature rows
        # (53 doubles + 8 ints) leaving the pipeline, over the measured time per step of the timed region
        alg_bytes = frames * 4 * geo["hop"] + rows * (53 * 8 + 8 * 4)
        step_s = dt / args.steps
        achieved = alg_bytes / step_s / 1e9
        # the kernels alone (three back-to-back steps before the timed region, HIP events between the stages): what
        # profiles/*_kernel_stats_in_flight_1.txt shows as per-kernel averages
        alone = {fe_name + " (PCM->Hann->FFT->mel->u32)": float(solo_ms[0]), "peaks_kernel + gate_kernel + span order": float(solo_ms[1]),
                 "tracker_kernel (formant tracking + finalize)": float(solo_ms[2]), "compaction": float(solo_ms[3])}
        dom = max(alone, key=alone.get)
        piped = {fe_name: float(stage[0]), "peaks + gate + span order": float(stage[1]), "tracker": float(stage[2]), "compaction": float(stage[3])}
        gaps = np.diff(np.array(stamps)) * 1e3 if len(stamps) > 2 else np.zeros(1)
        out = {
            "metric": "53-feat frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "repeats": {"n": len(regions), "what": "the timed region (warm-up + K steps between two synchronisation points) repeated inside this invocation; "
                                                   "ms_per_step / value = the median region",
                        "ms_per_step_all": region_ms, "min": min(region_ms), "median": dt / args.steps * 1e3, "max": max(region_ms)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 front end / f64 tracker",
            "data": "synthetic",
            # what the numbers' correctness rests on: the back end is pinned to the reference itself; the front end (and the rate converter K0) to
            # this build's own specification, because the reference's worklet is not in its tree
            "front_end_parity": "self-specified (FE-1): bit-exact against oracle/frontend.c, which no reference source can pin",
            "back_end_parity": "pinned: oracle/backend.c == module 584 of the reference under Node (tests/golden), HIP == oracle",
            "config": {"workload": f"{n_clips} clips x {args.seconds:g} s @{fs / 1000:g} kHz mono per GPU"
                                   + (f" (the {n_clips * world}-clip job of BASELINE config 4 sharded over {world} GPUs)" if world > 1 else "")
                                   + f", {geo['nfft']}-pt FFT, 25 ms hop, " + LEVEL_NAME[args.level],
                       "frames_per_step_per_gpu": frames, "feature_rows_per_step_per_gpu": rows,
                       "parallelism": f"clip-sharded x{world}, RCCL gather of feature rows" if world > 1 else "1 GPU",
                       "batches_in_flight": depth, "streams": n_streams, "planned_batches_per_stream": spp},
            "rccl_ranks": world if world > 1 and backend == "nccl" else None,
            "rows_gathered_on_rank0_last_step": max((s.gathered for s in slots), default=0) if world > 1 else None,
            "step_completion_interval_ms": {"p10": float(np.percentile(gaps, 10)), "p50": float(np.percentile(gaps, 50)), "p90": float(np.percentile(gaps, 90))},
            "backend_reruns": int(reruns),
            "stage_ms": {"frontend_fft_mel": float(stage[0]), "backend_peaks_gate_tracker": float(stage[1] + stage[2]),
                         "compaction": float(stage[3])},
            # primary object: the whole step against the HBM roofline (frac = achieved / peak can be recomputed from this line alone:
            # algorithmic_bytes_per_step / ms_per_step); the dominant kernel's own figures sit below it
            "roofline": {"bound": "hbm", "scope": "pipeline step: front end + peak scan + gate + tracker + compaction over one batch",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_per_kernel": traffic_per_kernel, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_step": alg_bytes, "ms_per_step": step_s * 1e3,
                         "copy_ceiling_GBps_measured": copy_ceiling(),
                         "dominant_kernel": {
                             "alone": dom, "alone_ms": alone[dom], "kernels_alone_ms": alone,
                             "pipelined": max(piped, key=piped.get), "kernels_pipelined_ms": piped,
                             "front_end_alone_hbm_frac": float(alg_bytes / (float(solo_ms[0]) / 1e3) / 1e9 / HBM_PEAK_GBS) if solo_ms[0] > 0 else 0.0,
                             "note": "alone = HIP events between the stages of 3 back-to-back steps (one batch has the GPU to itself; agrees with the "
                                     "per-kernel averages of profiles/*_kernel_stats_in_flight_1.txt); pipelined = the same events inside the timed "
                                     "region, where the kernels of the batches in flight share the CUs (an interval also contains the time a "
                                     "launch waited for them; profiles/*_kernel_stats_default.txt has the profiler's own begin/end times)"}},
            "issue_bound": issue_bound(n_clips, fs, args.level, args.seconds, step_s * 1e3, alg_bytes),
            "single_batch": {"what": "3 steps strictly back to back before the timed region (one rank, includes the gather when n_gpus > 1)",
                             "ms_per_step": solo_wall * 1e3, "value": frames / solo_wall,
                             "frontend_fft_mel_ms": float(solo_ms[0]), "backend_ms": float(solo_ms[1] + solo_ms[2]),
                             "peaks_gate_ms": float(solo_ms[1]), "tracker_ms": float(solo_ms[2]),
                             "compaction_ms": float(solo_ms[3]),
                             "frontend_hbm_frac": float(alg_bytes / (float(solo_ms[0]) / 1e3) / 1e9 / HBM_PEAK_GBS) if solo_ms[0] > 0 else 0.0,
                             # SURVEY.md 8d asks for the fp32 FLOP fraction next to the HBM one (the FFT sits near the ridge):
                             # algorithmic flops = 2.5 N log2 N (real FFT) + ~2 k (power, mel) per frame, vector fp32 peak 157.3 TFLOP/s
                             "frontend_fp32_flop_frac": float(frames * (2.5 * geo["nfft"] * np.log2(geo["nfft"]) + 2000.0) / (float(solo_ms[0]) / 1e3) / 157.3e12) if solo_ms[0] > 0 else 0.0},
        }
        if world == 1 and not args.no_extra:
            gpu_rows_last = slots[(args.steps - 1) % depth].batch.rows(slots[(args.steps - 1) % depth].stream.cuda_stream)
            for s in slots:
                s.batch.close()
            slots.clear()
            out["extra"] = {"level13": extra_level13(args, pcm, ns, n_clips, fs, dev, depth, Slot, run_steps, streams, geo["hop"]),
                            "streaming": extra_streaming(local_rank)}
            out["extra"]["offline_48k"] = extra_offline_48k(local_rank, n_clips, args.seconds, n_streams)
            # BASELINE config 1's chain as the reference's offline path runs it (ref dist/main.js:2 @B18765: a 44.1 kHz file decoded into a 48 kHz context):
            # 44.1 kHz clips -> K0 -> 48 kHz -> 3072-point front end -> rows
            out["extra"]["config1_chain"] = extra_offline_48k(local_rank, n_clips, args.seconds, n_streams, fs_in=44100)
            out["extra"]["host_path"] = extra_host_path(n_clips, args.seconds, args.level)
            # the configuration the APPLICATION ships (ref src/index.js:21: window_width 25, window_step 15, output_level 13) at the offline path's
            # 48 kHz context (ref dist/main.js:2 @B18765), and the same settings on the headline's 16 kHz clips
            out["extra"]["app_defaults"] = {
                "what": "the reference application's own settings (src/index.js:21): 25 ms windows every 15 ms, Syllable Features (level 13); headline protocol "
                        "(median of 3 regions of 20 steps between two synchronisation points)",
                "48k": extra_offline_48k(local_rank, n_clips, args.seconds, n_streams, level=13, window_step=15.0),
                "16k": extra_offline_48k(local_rank, n_clips, args.seconds, n_streams, fs=16000, level=13, window_step=15.0)}
        else:
            gpu_rows_last = slots[(args.steps - 1) % depth].batch.rows(slots[(args.steps - 1) % depth].stream.cuda_stream) if world == 1 else None
        if not args.no_cpu_baseline and world == 1:          # the CPU figure is taken once, at N = 1
            out["cpu_baseline"] = cpu_baseline(pcm, fs, args.level, min(args.cpu_clips, n_clips), gpu_rows_last)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def median_regions(run_region, repeats):
    """`repeats` timed regions (each: two synchronisation points around run_region()); returns (median dt, all dt, result of the median region)"""
    import torch
    outs = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run_region()
        torch.cuda.synchronize()
        outs.append((time.perf_counter() - t0, r))
    order = sorted(range(len(outs)), key=lambda i: outs[i][0])
    dt, r = outs[order[len(order) // 2]]
    return dt, [o[0] for o in outs], r


def hbm_roofline(alg_bytes, step_s, what):
    ach = alg_bytes / step_s / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_step": int(alg_bytes), "what": what}


def extra_level13(args, pcm, ns, n_clips, fs, dev, depth, Slot, run_steps, streams, hop):
    """BASELINE configs[2]: Syllable Features (segmenter state machine + per-syllable reduction) on the same 1024-clip batch — the headline's protocol:
    `repeats` regions of max(20, --steps) steps, each between two synchronisation points after a warm-up; the median region is reported."""
    import torch
    from webspeechanalyzer_amd import Analyzer, Config
    an = Analyzer(Config(output_level=13), device=dev.index)
    slots = [Slot(an, streams[j % len(streams)]) for j in range(depth)]
    frames = slots[0].batch.info["n_frames_total"]
    steps = max(20, args.steps)
    reps = max(3, min(5, args.repeats))
    run_steps(slots, depth, depth)
    torch.cuda.synchronize()

    def region():
        return run_steps(slots, steps, depth)[0]
    run_steps(slots, max(args.warmup, 3), depth)
    dt, all_dt, rows = median_regions(region, reps)
    for s in slots:
        s.batch.close()
    an.close()
    alg = frames * 4 * hop + int(rows) * (53 * 8 + 8 * 4)
    return {"workload": f"{n_clips} clips x {ns / fs:g} s @{fs / 1000:g} kHz, Syllable Features (level 13), {depth} batches in flight",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": frames * steps / dt, "unit": "frames/s", "syllable_rows_per_step": int(rows),
            "repeats": {"n": reps, "ms_per_step_all": [d / steps * 1e3 for d in all_dt], "what": "median of n regions of `steps` steps, as the headline"},
            "roofline": hbm_roofline(alg, dt / steps, "PCM read once (4 x hop B per frame) + 456 B per syllable row, over ms_per_step")}


def extra_host_path(n_clips, seconds, level):
    """What a JavaScript caller of LaunchBatch gets (webspeechanalyzer_amd/js/bench_host.js as a child process): 16-bit PCM clips in host
    memory -> N-API worker thread -> upload (PCIe) + device-side int16 -> float + the kernels -> rows -> callbacks on the JS thread.
    PCIe-inclusive, so never `value`."""
    import shutil
    node = shutil.which("node")
    js = os.path.join(ROOT, "webspeechanalyzer_amd", "js", "bench_host.js")
    if node is None or not os.path.exists(os.path.join(ROOT, "webspeechanalyzer_amd", "lib", "wsa_napi.node")):
        return {"skipped": "node or the N-API addon is not available"}
    out = {}
    for kind in ("i16", "f32", "i16p", "i16ps"):
        try:
            r = subprocess.run([node, js, str(n_clips), str(seconds), str(level), kind], capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            key = "i16p_sustained" if kind == "i16ps" else kind
            out[key] = {"value": d["value"], "unit": "frames/s", "ms_per_batch": d["best_s"] * 1e3, "clips": d["clips"], "rows": d["rows"]}
            if kind == "i16ps":
                out[key].update(batches=d["batches"], one_at_a_time_ms_per_batch=d["one_at_a_time_s"] * 1e3, rows_equal_one_at_a_time=d["rows_equal"])
        except (OSError, ValueError, IndexError, KeyError, subprocess.SubprocessError) as e:
            out[kind] = {"error": str(e)[:200]}
    out["what"] = ("LaunchBatch through the Node host, best of 5: Int16Array clips (i16: what WAV files hold; converted on the device) "
                   "and Float32Array clips (f32) in ordinary (pageable) host memory, and 16-bit clips in page-locked buffers from allocPinned (i16p: DMA "
                   "straight out of the caller's buffers); PCIe + marshalling + callbacks included.  i16p_sustained: 12 such batches back to back through LaunchBatches "
                   "(two contexts, each with its own planned batch and HIP stream: batch k + 1 uploads while batch k computes and batch k - 1's callbacks run), ms per batch; "
                   "beside it the same batches one LaunchBatch at a time, awaited (what the app's file loop does, ref src/index.js:277-296).  PCIe floor: 327 MB at ~50 GB/s = 6.5 ms")
    return out


def extra_offline_48k(device, n_clips, seconds, depth, steps=20, fs_in=None, repeats=3, fs=48000, level=5, window_step=None):
    """The same batch at the rate the reference's offline path always analyses at (`new OfflineAudioContext(1, 48e6, 48e3)`, ref dist/main.js:2
    @B18765): 48 kHz, 3072-point FFT, 1200-sample frames — what a real file costs once it has been brought to the context rate.
    The headline's protocol: `repeats` regions of `steps` steps between two synchronisation points, median."""
    import torch
    from webspeechanalyzer_amd import Analyzer, Config
    from webspeechanalyzer_amd.synth import synth_clips
    fs_src = fs_in or fs                                     # fs_in: the clips are at that rate and K0 converts them to 48 kHz in front (spec RS-1)
    ns = int(seconds * fs_src)
    dev = torch.device("cuda", device)
    pcm = synth_clips(n_clips, ns, fs=fs_src, seed=3, device=f"cuda:{device}")
    # window_step: the APPLICATION's settings (extra.app_defaults) — windows of 25 ms every 15 ms overlap, a frame re-reads 40 % of its samples
    an = Analyzer(Config(output_level=level, **({"window_step": window_step} if window_step else {})), device=device)
    geo = an.geometry(fs)
    batches = [an.batch([ns] * n_clips, fs_src, resample_to=fs if fs_in else None) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    busy = [False] * depth
    frames = batches[0].info["n_frames_total"]
    rows = 0

    def run(k_steps):
        nonlocal rows
        for k in range(k_steps + depth):
            i = k % depth
            if busy[i]:
                rows = batches[i].device_result(streams[i].cuda_stream).n_rows
                busy[i] = False
            if k < k_steps:
                batches[i].run(pcm.data_ptr(), pcm.stride(0), streams[i].cuda_stream)
                busy[i] = True
        return rows

    run(3)
    dt, all_dt, rows = median_regions(lambda: run(steps), repeats)
    for b in batches:
        b.close()
    an.close()
    # algorithmic bytes: the clips as they arrive, read once (44.1 kHz samples when K0 sits in front; the converted signal is an intermediate) + the rows;
    # with overlapping windows SURVEY.md 8(d)'s figure: 4 x hop samples per frame (every sample still counted once) + the rows
    alg = (frames * 4 * geo["hop"] if window_step else n_clips * ns * 4) + int(rows) * (53 * 8 + 8 * 4)
    return {"workload": f"{n_clips} clips x {seconds:g} s @{fs_src / 1000:g} kHz" + (" -> K0 rate converter -> 48 kHz" if fs_in else "")
                        + f", {geo['nfft']}-pt FFT, " + (f"{geo['win']}-sample window every {geo['hop']} samples, " if window_step else "")
                        + f"{LEVEL_NAME[level]}, {depth} batches in flight",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": frames * steps / dt, "unit": "frames/s", "frames_per_step": int(frames), "feature_rows_per_step": int(rows),
            "pcm_GBps": n_clips * ns * 4 * steps / dt / 1e9,
            "repeats": {"n": repeats, "ms_per_step_all": [d / steps * 1e3 for d in all_dt], "what": "median of n regions of `steps` steps, as the headline"},
            "roofline": hbm_roofline(alg, dt / steps, ("4 x hop B per frame (SURVEY.md 8d: every sample counted once although windows overlap)" if window_step
                                                       else "the clips as handed over, read once (4 B per input sample)") + " + 456 B per feature row, over ms_per_step")}


def extra_streaming(device, n=512, fs=48000, steps=2000, warmup=200):
    """BASELINE configs[4]: 512 concurrent real-time 48 kHz mono streams, one hipGraph-captured step per 25 ms frame; the step
    (pinned host samples -> feature rows visible to the host) is timed inside libwsa (wsa_stream_time_steps), p50 / p99."""
    import numpy as np
    from webspeechanalyzer_amd import Analyzer, Config
    from webspeechanalyzer_amd.synth import synth_clips
    an = Analyzer(Config(output_level=5), device=device)
    g = an.geometry(fs)
    st = an.streams(n, fs, frames_per_step=1, max_span_frames=1024)
    st.enable_graph(True)
    sps = st.samples_per_step
    loop = 400                                              # 10 s of signal per stream, cycled
    feed = synth_clips(n, loop * sps, fs=fs, seed=5, device=f"cuda:{device}").cpu().numpy().reshape(n, loop, sps).transpose(1, 0, 2).copy()
    st.time_steps(warmup, feed)
    us, rows = st.time_steps(steps, feed)
    st.close(); an.close()
    ms = us / 1e3
    period = 1e3 * sps / fs
    return {"workload": f"{n} concurrent {fs} Hz mono streams, 1 frame ({period:g} ms) per hipGraph step, {g['nfft']}-pt FFT, level 5; "
                        "step = pinned host samples -> rows visible to the host, timed inside libwsa",
            "steps": steps, "p50_ms": float(np.percentile(ms, 50)), "p99_ms": float(np.percentile(ms, 99)), "max_ms": float(ms.max()),
            "real_time_budget_ms": period, "rows": int(rows)}


def issue_bound(n_clips, fs, level, seconds, ms_per_step, alg_bytes):
    """The ruler next to `roofline`: this pipeline is bound by instruction issue, not by HBM.  Wave-instructions per step by class from the newest
    committed profiles/*_pmc_insts.json taken with THIS workload (one `rocprofv3 --pmc SQ_INSTS_*` pass of `bench.py --in-flight 1`, tools/pmc_insts.sh),
    the time the VALU instructions alone need at one per 4 cycles on each of the chip's 1024 SIMDs at the clock measured under this load
    (SQ_BUSY_CYCLES / duration, tools/pmc_util.sh) — THE stated bound —, ms_per_step against it, and the HBM-roofline fraction the step would reach AT
    that floor (the ceiling of today's instruction count).  The sum over all instruction classes is kept under `empirical_fit` only: the mixed-issue
    microbenchmark of round 5 shows the classes do not share one issue port."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_insts.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d.get("workload") or {}
            if (w.get("clips"), w.get("fs"), w.get("level"), w.get("seconds")) != (n_clips, fs, level, seconds):
                continue
            per = d["kernels"]
            tot = {}
            for v in per.values():
                for c, x in v.items():
                    tot[c] = tot.get(c, 0.0) + x
            clock = float(d.get("clock_GHz_under_load") or 2.2)
            simds = 1024
            valu = tot.get("SQ_INSTS_VALU", 0.0)
            allc = sum(tot.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
            valu_ms = valu * 1e6 * 4 / simds / (clock * 1e9) * 1e3
            all_ms = allc * 1e6 * 4 / simds / (clock * 1e9) * 1e3
            return {"bound": "valu_issue",
                    # the headline of this object: the step against the VALU-issue floor of today's instruction count
                    "frac_of_valu_issue_rate": valu_ms / ms_per_step if ms_per_step > 0 else None,
                    "valu_issue_ms": valu_ms,
                    "hbm_frac_at_valu_issue_floor": alg_bytes / (valu_ms / 1e3) / 1e9 / HBM_PEAK_GBS if valu_ms > 0 else None,
                    "wave_instructions_per_step_M": {k.replace("SQ_INSTS_", ""): round(x, 2) for k, x in sorted(tot.items()) if k.startswith("SQ_INSTS_")},
                    "per_kernel_M": {k.replace("wsa::", ""): {c.replace("SQ_INSTS_", ""): x for c, x in v.items() if c.startswith("SQ_INSTS_")} for k, v in per.items()},
                    "clock_GHz_under_load": clock, "simds": simds,
                    "empirical_fit": {"all_issue_ms": all_ms, "frac_of_all_issue_rate": all_ms / ms_per_step if ms_per_step > 0 else None,
                                      "note": "NOT a bound: all wave-instructions (VALU + SALU + LDS + VMEM) x 4 cycles / 1024 SIMDs / clock happens to equal the measured step. "
                                              "tools/microbench/mixed_issue.hip (profiles/r05_notes.md): an SALU / LDS / VMEM instruction of another wave does not take a VALU "
                                              "issue slot (a VALU wave runs at 8.9 - 9.4 cycles per instruction beside an SALU wave, 9.0 alone); what the fit measures is that a WAVE "
                                              "issues one instruction of any kind per 6 - 10 cycles, so a SIMD with the pipeline's 4 - 5 resident waves, a third to a half of them "
                                              "waiting, gets about one instruction per 4 cycles"},
                    "source": os.path.relpath(f, ROOT) + stale_note(f),
                    "note": "valu_issue_ms = VALU wave-instructions x 4 cycles / 1024 SIMDs / clock: the floor of TODAY'S instruction count with perfect overlap, priced at the "
                            "4 cycles a packed-fp32 or fp64 instruction occupies a SIMD (measured: mixed_issue.hip; most of the front end's and the tracker's VALU work) — "
                            "plain fp32 / integer instructions issue in 2, so the true port floor is somewhat lower; frac_of_valu_issue_rate = that floor / ms_per_step"}
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


def stale_note(f):
    """' @ <commit>' of a profile file, plus ' (STALE: csrc changed since)' when a commit touching webspeechanalyzer_amd/csrc is newer than the file's."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %ct", "--", f], capture_output=True, text=True, timeout=10).stdout.split()
        k = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%ct", "--", "webspeechanalyzer_amd/csrc"], capture_output=True, text=True, timeout=10).stdout.split()
        if not h:
            return ""
        return f" @ {h[0]}" + (" (STALE: a commit touching webspeechanalyzer_amd/csrc is newer than this profile)" if k and int(k[0]) > int(h[1]) else "")
    except (OSError, ValueError, subprocess.SubprocessError):
        return ""


def pmc_traffic(n_clips, fs, level, seconds):
    """HBM bytes per step (all kernels of one pass) and per kernel launch from a committed rocprofv3 PMC summary (FETCH_SIZE and WRITE_SIZE
    are collected in separate --pmc passes of this command and corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py writes
    profiles/*_pmc_traffic.json).  Only a summary taken with THIS workload is used, and its file + commit are named; else (None, None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            w = d.get("workload") or {}
            if (w.get("clips"), w.get("fs"), w.get("level"), w.get("seconds")) != (n_clips, fs, level, seconds):
                continue
            per = {k.replace("wsa::", ""): v["hbm_bytes_per_launch"] for k, v in d["kernels"].items()}
            return sum(per.values()), per, os.path.relpath(f, ROOT) + stale_note(f)
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def cpu_parity(host_rows, gpu_rows, n):
    """The CPU restatement's callbacks for the first n clips against the GPU's rows of the same clips (segment indices exact,
    the 53 doubles within the contract's 1e-4 relative / 1e-6 absolute)."""
    import numpy as np
    if gpu_rows is None:
        return "not compared"
    meta, feat, roff = gpu_rows["meta"], gpu_rows["feat"], gpu_rows["row_off"]
    rows = 0
    for c in range(n):
        want = host_rows[c]
        a, b = int(roff[c]), int(roff[c + 1])
        if b - a != len(want):
            return f"MISMATCH: clip {c} has {b - a} GPU rows, {len(want)} CPU rows"
        for k, (t0, tl, f) in enumerate(want):
            m = meta[a + k]
            if (int(m[2]), int(m[3])) != (t0, tl):
                return f"MISMATCH: clip {c} row {k} indices {(int(m[2]), int(m[3]))} vs {(t0, tl)}"
            if not np.allclose(feat[a + k], f, rtol=1e-4, atol=1e-6, equal_nan=True):
                return f"MISMATCH: clip {c} row {k} features"
            rows += 1
    return f"ok ({n} clips, {rows} rows: indices exact, features within 1e-4)"


def cpu_baseline(pcm, fs, level, n, gpu_rows=None):
    """The CPU restatements of the reference algorithm (oracle/ — test infrastructure, used here only as
    the thing timed BESIDE the GPU path) on this box's host cores, single thread, on the first clips of
    the very batch the GPU processed.  Headline = the Node/JS path (oracle/js, what north_star asks
    for: the reference itself is JavaScript); the plain-C port's rate is reported next to it."""
    import shutil
    import tempfile
    from oracle import pyoracle
    host = pcm[:n].cpu().numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    cfg = pyoracle.default_cfg(level=level)
    step = float(cfg.window_step) / 1e3
    t0 = time.perf_counter()
    frames = 0
    host_rows = []
    t_be = 0.0
    for c in range(n):
        sp = fe.run(host[c])
        tb = time.perf_counter()
        r = pyoracle.run_backend(sp, cfg)
        t_be += time.perf_counter() - tb
        frames += sp.shape[0]
        rows_c = []
        if level == 5:
            for cb in r["callbacks"]:
                rows_c.append((int(round(cb[2][0] / step)), int(round(cb[2][1] / step)) - 1, cb[3]))
        host_rows.append(rows_c)
    dt = time.perf_counter() - t0
    parity = cpu_parity(host_rows, gpu_rows, n) if level == 5 else "not compared (level 5 only)"
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    c_port = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
              "sample": f"first {n} clips ({frames} frames) of the GPU batch, C oracle (oracle/), 1 thread, {dt:.1f} s",
              # the stage the reference's own code covers (u32 frames -> rows: SURVEY.md section 6 measured 6 x 10^5 frames/s for it under Node) on its own:
              # most of the port's time is its plain-C front end (FE-1's FFT), which the reference leaves to the browser's worklet
              "back_end_only": {"value": frames / t_be if t_be > 0 else None, "unit": "frames/s", "share_of_port_time": t_be / dt if dt > 0 else None}}
    node = shutil.which("node")
    if node is None:
        return dict(c_port, cpu=cpu, host_cores=os.cpu_count(), node=None, cpu_parity=parity)
    # size the Node sample for ~15 s from the C rate (the JS restatement runs ~10x slower: fp32 via Math.fround)
    nj = max(4, min(n, int(15.0 * c_port["value"] / 10.0 / max(1, frames // n))))
    root = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as d:
        files = []
        for c in range(nj):
            f = os.path.join(d, f"c{c}.f32")
            host[c].tofile(f)
            files.append(f)
        job = os.path.join(d, "job.json")
        with open(job, "w") as fh:
            json.dump({"mode": "time", "files": files, "fs": fs, "settings": {"output_level": level}}, fh)
        r = subprocess.run([node, os.path.join(root, "oracle", "js", "run.js"), job], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return dict(c_port, cpu=cpu, host_cores=os.cpu_count(), node="failed: " + r.stderr[-200:], cpu_parity=parity)
    j = json.loads(r.stdout)
    ver = subprocess.run([node, "--version"], capture_output=True, text=True).stdout.strip()
    # the same Node path on many cores: P worker processes over disjoint clip shards (wall time from the first spawn
    # to the last exit, i.e. including node start-up)
    many = None
    try:
        procs_n = max(1, min(os.cpu_count() or 1, 64))
        per = 32
        with tempfile.TemporaryDirectory() as d:
            distinct = min(n, 256)
            paths = []
            for c in range(distinct):
                f = os.path.join(d, f"c{c}.f32")
                host[c].tofile(f)
                paths.append(f)
            jobs = []
            for w in range(procs_n):
                files = [paths[(w * per + i) % distinct] for i in range(per)]
                job = os.path.join(d, f"job{w}.json")
                with open(job, "w") as fh:
                    json.dump({"mode": "time", "files": files, "fs": fs, "settings": {"output_level": level}}, fh)
                jobs.append(job)
            t0 = time.perf_counter()
            ps = [subprocess.Popen([node, os.path.join(root, "oracle", "js", "run.js"), jb], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for jb in jobs]
            outs = [p_.communicate(timeout=600)[0] for p_ in ps]
            wall = time.perf_counter() - t0
        fr = sum(json.loads(o)["frames"] for o in outs)
        many = {"value": fr / wall, "unit": "frames/s", "cores": procs_n, "kind": "port",
                "sample": f"{procs_n} node processes x {per} clips ({fr} frames), wall {wall:.1f} s incl. start-up"}
    except Exception as e:                                   # a reported extra, never fatal
        many = {"error": str(e)[:200]}
    return {"value": j["frames"] / (j["ms"] / 1e3), "unit": "frames/s", "cores": 1, "kind": "port", "many_cores": many,
            "sample": f"first {nj} clips ({j['frames']} frames) of the GPU batch, JS oracle (oracle/js) under node {ver}, 1 thread, {j['ms'] / 1e3:.1f} s",
            "cpu": cpu, "host_cores": os.cpu_count(), "c_port": c_port, "cpu_parity": parity}


if __name__ == "__main__":
    main()
