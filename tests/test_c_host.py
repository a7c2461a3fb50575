"""The C ABI from plain C (examples/c_host.c): the header is valid C99, the program links against libwsa.so alone, and on
a GPU its rows equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from tests import util

EX = os.path.join(util.ROOT, "examples", "c_host.c")
LIBDIR = os.path.join(util.ROOT, "webspeechanalyzer_amd", "lib")


def _build(tmp_path):
    from webspeechanalyzer_amd import capi
    if not os.path.exists(capi.library_path()):
        capi.build_library()
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(util.ROOT, "include"), EX, "-L", LIBDIR, "-lwsa",
                    "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_header_is_c99_and_program_links(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_host_rows_match_oracle(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import pyoracle
    from webspeechanalyzer_amd.synth import synth_clips
    exe = _build(tmp_path)
    fs, n = 16000, 3
    pcm = synth_clips(n, 5 * fs, fs=fs, seed=91, device="cpu").numpy()
    files = []
    for i in range(n):
        f = tmp_path / f"c{i}.f32"; pcm[i].tofile(f); files.append(str(f))
    r = subprocess.run([exe, "5", str(fs)] + files, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = [line.split() for line in r.stdout.strip().splitlines()]
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg(fs=fs))
    k = 0
    for c in range(n):
        ref = pyoracle.run_backend(fe.run(pcm[c]), pyoracle.default_cfg(level=5))
        for cb in ref["callbacks"]:
            row = rows[k]; k += 1
            assert int(row[0]) == c and int(row[1]) == cb[0]
            assert abs(int(row[2]) * 0.025 - cb[2][0]) < 1e-12 and abs((int(row[3]) + 1) * 0.025 - cb[2][1]) < 1e-12
            assert util.rel_err(np.array([float(x) for x in row[4:]]), cb[3]) <= 1e-4
    assert k == len(rows) and k > 5
