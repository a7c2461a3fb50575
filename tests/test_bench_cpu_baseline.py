"""bench.py's cpu_baseline leg (the only place outside tests/ and smoke() that may run the oracle) on a tiny sample."""
import shutil

import pytest


def test_cpu_baseline_reports_node_and_c_ports():
    import bench
    from webspeechanalyzer_amd.synth import synth_clips
    pcm = synth_clips(4, 3 * 16000, fs=16000, seed=3)
    r = bench.cpu_baseline(pcm, 16000, 5, 4)
    assert r["unit"] == "frames/s" and r["cores"] == 1 and r["kind"] == "port" and r["value"] > 0
    if shutil.which("node"):
        assert "JS oracle" in r["sample"] and r["c_port"]["value"] > 0
        assert r["many_cores"]["value"] > 0 and r["many_cores"]["cores"] >= 1
    else:
        assert "C oracle" in r["sample"]
