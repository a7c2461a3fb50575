#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on synthetic 16 kHz mono PCM.

A "step" = one pass of the whole pipeline (PCM -> Hann -> FFT -> mel -> u32 -> peak scan -> tracker
-> 53-feature rows) over one batch per GPU; the batch is BASELINE.json configs[1]
(1024 clips x 10 s, 1024-pt FFT, 25 ms hop, Segment Features), resident in HBM before the timed
region.  With N GPUs every rank runs the same per-GPU batch (weak scaling) and the feature
matrices are gathered to rank 0 with one RCCL gather per step.  Steps are software-pipelined over D = --in-flight
slots (default 3): step k runs on slot k % D with its own planned batch and HIP stream, so the tracker's
low-occupancy tail of one step and the gather overlap the front end of the next — every step still is one full
pass over one batch, and `value` = the K steps' frames over the wall time between the two synchronisation points.
Kernel durations measured by HIP events inside the timed region are therefore those of kernels SHARING the GPU
(`roofline` follows the contract and uses them); the same line carries `single_batch`: three steps run strictly
back to back just before the timed region, whose per-kernel times are the ones a rocprofv3 profile of
`bench.py --in-flight 1` shows (profiles/*_kernel_stats.txt) and DESIGN.md quotes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level 5|13] [--clips C] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--level", type=int, default=5, choices=(5, 13, 10, 11, 12))
    ap.add_argument("--clips", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--fs", type=int, default=16000, help="sample rate of the synthetic clips (BASELINE configs use 16000)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="batches in flight: step k runs on slot k %% D (own planned batch + HIP stream), so the tracker's tail of one "
                         "step and the RCCL gather overlap the front end of the next; every step still is one full pass over one batch. "
                         "1 = strictly back to back (what profiles/*_kernel_stats.txt is taken with)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=768)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})", file=sys.stderr)
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
    from webspeechanalyzer_amd.synth import synth_clips

    fs = args.fs
    ns = int(args.seconds * fs)
    n_clips = args.clips
    pcm = synth_clips(n_clips, ns, fs=fs, seed=1000 + rank, device=dev)          # HBM resident before timing
    an = Analyzer(Config(output_level=args.level), device=local_rank)
    depth = max(1, args.in_flight)
    geo = an.geometry(fs)
    from webspeechanalyzer_amd.gather import gather_rows

    class Slot:
        def __init__(self):
            self.batch = an.batch([ns] * n_clips, fs)
            self.stream = torch.cuda.Stream(device=dev)
            self.rows_cap = self.batch.info["rows_cap"]
            # gather buffers (rank 0 receives): features and metadata keep their own dtypes
            self.feat = torch.empty((self.rows_cap, 53), dtype=torch.float64, device=dev) if world > 1 else None
            self.meta = torch.empty((self.rows_cap, 8), dtype=torch.int32, device=dev) if world > 1 else None
            self.busy = False

        def launch(self):
            self.batch.run(pcm.data_ptr(), pcm.stride(0), self.stream.cuda_stream)
            self.busy = True

        def finish(self):
            """Wait for this slot's step, read its row counters; multi-GPU: the single exchange of the job — feature
            matrices to rank 0 over RCCL (xGMI), SURVEY.md 8e."""
            b, st = self.batch, self.stream.cuda_stream
            r = b.device_result(st)                          # syncs the slot's stream
            if world > 1:
                b.an._check(b.L.wsa_batch_copy_rows(b.h, st, self.meta.data_ptr(), self.feat.data_ptr(), self.rows_cap, None, 0, None, None))
                with torch.cuda.stream(self.stream):
                    gather_rows(self.meta, self.feat, r.n_rows, rank * n_clips)
            self.busy = False
            return r.n_rows, b.stage_ms()

    slots = [Slot() for _ in range(depth)]
    frames = slots[0].batch.info["n_frames_total"]
    torch.cuda.synchronize()

    def run_steps(k_steps, depth=max(1, args.in_flight)):
        """k_steps steps, slot k % depth each; returns (rows of the last finished step, summed stage ms)."""
        stage = np.zeros(4)
        rows = 0
        for k in range(k_steps):
            sl = slots[k % depth]
            if sl.busy:
                rows, ms = sl.finish()
                stage += ms
            sl.launch()
        for j in range(depth):                               # drain in launch order
            sl = slots[(k_steps + j) % depth]
            if sl.busy:
                rows, ms = sl.finish()
                stage += ms
        return rows, stage

    run_steps(args.warmup)
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, stage = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stage /= max(args.steps, 1)
    depth_t = max(1, args.in_flight)

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
        traffic = pmc_traffic("fe_kernel_r8") if geo["nfft"] == 1024 else None
        total_frames = frames * world * args.steps
        value = total_frames / dt
        # roofline of the dominant kernel (front end, K1): algorithmic bytes per launch =
        # 4 * hop samples per frame (PCM read once) + the 53-feature rows leaving the pipeline
        alg_bytes = frames * 4 * geo["hop"] + rows * (53 * 8 + 8 * 4)
        fe_s = stage[0] / 1e3
        achieved = alg_bytes / fe_s / 1e9 if fe_s > 0 else 0.0
        out = {
            "metric": "53-feat frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 front end / f64 tracker",
            "data": "synthetic",
            "config": {"workload": f"{n_clips} clips x {args.seconds:g} s @{fs / 1000:g} kHz mono per GPU, {geo['nfft']}-pt FFT, 25 ms hop, "
                                   + {5: "Segment Features (level 5)", 13: "Syllable Features (level 13)", 10: "Syllable Formants (level 10)",
                                      11: "Utterance Features (level 11)", 12: "Syllable Polynomials (level 12)"}[args.level],
                       "frames_per_step_per_gpu": frames, "feature_rows_per_step_per_gpu": rows,
                       "parallelism": f"clip-sharded x{world}, RCCL gather of feature rows" if world > 1 else "1 GPU",
                       "batches_in_flight": depth_t},
            "stage_ms": {"frontend_fft_mel": float(stage[0]), "backend_peaks_gate_tracker": float(stage[1] + stage[2]),
                         "compaction": float(stage[3])},
            "whole_pipeline_hbm_frac": (frames * 4 * geo["hop"] + rows * 456) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "kernel": ("fe_kernel_r8" if geo["nfft"] == 1024 else "fe_kernel_rx") + " (PCM->Hann->FFT->mel->u32)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "copy_ceiling_GBps_measured": copy_ceiling(),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": float(stage[0]),
                         "note": "launch duration = HIP events around the kernel on its stream, averaged over the timed steps"
                                 + (": the kernel shares the GPU with the tracker of the step before it (batches_in_flight > 1); "
                                    "alone it takes single_batch.frontend_fft_mel_ms" if depth_t > 1 else "")},
            "single_batch": {"what": "3 steps strictly back to back before the timed region (one rank, includes the gather when n_gpus > 1)",
                             "ms_per_step": solo_wall * 1e3, "value": frames / solo_wall,
                             "frontend_fft_mel_ms": float(solo_ms[0]), "backend_ms": float(solo_ms[1] + solo_ms[2]),
                             "compaction_ms": float(solo_ms[3]),
                             "frontend_hbm_frac": float(alg_bytes / (float(solo_ms[0]) / 1e3) / 1e9 / HBM_PEAK_GBS) if solo_ms[0] > 0 else 0.0,
                             # SURVEY.md 8d asks for the fp32 FLOP fraction next to the HBM one (the FFT sits near the ridge):
                             # algorithmic flops = 2.5 N log2 N (real FFT) + ~2 k (power, mel) per frame, vector fp32 peak 157.3 TFLOP/s
                             "frontend_fp32_flop_frac": float(frames * (2.5 * geo["nfft"] * np.log2(geo["nfft"]) + 2000.0) / (float(solo_ms[0]) / 1e3) / 157.3e12) if solo_ms[0] > 0 else 0.0},
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU figure is taken once, at N = 1
            out["cpu_baseline"] = cpu_baseline(pcm, fs, args.level, min(args.cpu_clips, n_clips))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate --pmc runs of this same command and corrected as MI355X_MICROARCH.md prescribes; the
    summary lives in profiles/*_pmc_traffic.json).  None if no such summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        for k, v in d["kernels"].items():
            if kernel in k:
                return v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(pcm, fs, level, n):
    """The CPU restatements of the reference algorithm (oracle/ — test infrastructure, used here only as
    the thing timed BESIDE the GPU path) on this box's host cores, single thread, on the first clips of
    the very batch the GPU processed.  Headline = the Node/JS path (oracle/js, what north_star asks
    for: the reference itself is JavaScript); the plain-C port's rate is reported next to it."""
    import shutil
    import subprocess
    import tempfile
    from oracle import pyoracle
    host = pcm[:n].cpu().numpy()
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    cfg = pyoracle.default_cfg(level=level)
    t0 = time.perf_counter()
    frames = 0
    for c in range(n):
        sp = fe.run(host[c])
        pyoracle.run_backend(sp, cfg)
        frames += sp.shape[0]
    dt = time.perf_counter() - t0
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    c_port = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
              "sample": f"first {n} clips ({frames} frames) of the GPU batch, C oracle (oracle/), 1 thread, {dt:.1f} s"}
    node = shutil.which("node")
    if node is None:
        return dict(c_port, cpu=cpu, host_cores=os.cpu_count(), node=None)
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
        return dict(c_port, cpu=cpu, host_cores=os.cpu_count(), node="failed: " + r.stderr[-200:])
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
            "cpu": cpu, "host_cores": os.cpu_count(), "c_port": c_port}


if __name__ == "__main__":
    main()
