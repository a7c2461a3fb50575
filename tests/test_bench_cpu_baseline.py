"""bench.py's cpu_baseline leg (the only place outside tests/ and smoke() that may run the oracle) on a tiny sample."""
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_reports_node_and_c_ports():
    import bench
    from webspeechanalyzer_amd.synth import synth_clips
    pcm = synth_clips(4, 3 * 16000, fs=16000, seed=3)
    r = bench.cpu_baseline(pcm, 16000, 5, 4)
    assert r["cpu_parity"] == "not compared"
    assert r["unit"] == "frames/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0
    if shutil.which("node"):
        assert "JS oracle" in r["sample"] and r["c_port"]["value"] > 0
        assert r["many_cores"]["value"] > 0 and r["many_cores"]["cores"] >= 1
    else:
        assert "C oracle" in r["sample"]


@pytest.mark.gpu
def test_bench_multi_rank_code_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (clip shards per rank, pipelined slots, the per-step gather to rank 0, max-over-ranks timing)
    with two ranks sharing ONE GPU over gloo (WSA_BENCH_BACKEND test hook; numbers meaningless): the JSON line must come out
    with n_gpus = 2 and twice the frames of one rank per step."""
    import json
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["WSA_BENCH_BACKEND"] = "gloo"
    # no launcher: `bench.py --gpus 2` starts its own two ranks (torch.distributed.run as a child process)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--clips", "48", "--seconds", "4", "--master-port", "29541"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert out["rows_gathered_on_rank0_last_step"] >= 2 * out["config"]["feature_rows_per_step_per_gpu"] - 40 and out["backend_reruns"] == 0
    assert out["config"]["frames_per_step_per_gpu"] == 48 * 160
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - 2 * 48 * 160) < 1e-6 * 2 * 48 * 160
