#!/usr/bin/env python3
"""One-off differential fuzzing on the GPU box: the random-configuration tests of tests/test_gpu_parity.py and
tests/test_gpu_stream.py over a range of seeds.   usage: python tools/fuzz_gpu.py FIRST LAST [stream]
(WSA_FUZZ_LEVEL=3 forces an output level for the batch test)"""
import os
os.environ.setdefault("WSA_TUNING_ENV", "1")   # libwsa reads its tuning switches only when this is set
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webspeechanalyzer_amd as wsa
from tests import test_gpu_parity, test_gpu_stream

first, last = int(sys.argv[1]), int(sys.argv[2])
stream = len(sys.argv) > 3 and sys.argv[3] == "stream"
fn = test_gpu_stream.test_stream_random_settings_vs_oracle if stream else test_gpu_parity.test_random_configurations_vs_oracle
bad = []
for seed in range(first, last + 1):
    try:
        fn(wsa, seed)
    except Exception:
        bad.append(seed)
        print("FAILED seed", seed)
        traceback.print_exc(limit=2)
print(f"{'stream' if stream else 'batch'} seeds {first}..{last}: {last - first + 1 - len(bad)} passed, failed: {bad}")
sys.exit(1 if bad else 0)
