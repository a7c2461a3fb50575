"""CPU-side checks of the boundary: the HIP library builds for gfx950, loads, and exports every
symbol include/wsa.h declares.  No compute calls (no GPU here)."""
import os
import re

from webspeechanalyzer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    capi.build_library()
    L = capi.lib()
    assert L.wsa_abi_version() == capi.ABI_VERSION == 5
    header = open(os.path.join(ROOT, "include", "wsa.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(wsa_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/wsa.h but not exported by libwsa.so"
    assert sorted(capi.ABI_SYMBOLS) == declared


def test_config_defaults_are_the_reference_defaults():
    c = capi.Config()
    # ref dist/main.js:2 @B2965
    assert (c["spec_type"], c["output_level"], c["f_min"], c["f_max"], c["N_fft_bins"], c["N_mel_bins"]) == (1, 4, 50, 4000, 256, 128)
    assert (c["window_width"], c["window_step"], c["pause_length"], c["min_seg_length"]) == (25, 25, 200, 50)
    assert (c["auto_noise_gate"], c["voiced_max_dB"], c["voiced_min_dB"], c["pre_norm_gain"], c["high_f_emph"]) == (1, 100, 10, 1000, 0)


def test_no_device_fails_loudly():
    import ctypes
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.WsaError):
        capi.Analyzer(capi.Config())


def test_front_end_asm_blocks_are_the_generators_output():
    """csrc/fe_blocks.inc is generated (tools/gen/fe_blocks.py: operation lists -> register allocation -> asm text, every block replayed on
    random fp32 values against the plain formulas); the committed file must be what the generator prints."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "fe_blocks.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == open(os.path.join(root, "webspeechanalyzer_amd", "csrc", "fe_blocks.inc")).read()
