"""webspeechanalyzer_amd — MI355X-native formantanalyzer hot path.

The product is the HIP library `lib/libwsa.so` behind the C ABI of include/wsa.h; this package is
the thin Python host plumbing used by the tests and bench.py (device memory and streams come from
PyTorch-ROCm).  The JavaScript host (js/formantanalyzer.js + the N-API addon) is the drop-in for
the reference's `require('formantanalyzer')`.
"""
from .capi import ACTIVE, START, STOP, Analyzer, Batch, Config, Gather, Streams, WsaError, build_library, library_path  # noqa: F401
