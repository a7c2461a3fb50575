#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
run() { label="$1"; shift; python3 bench.py --clips 12500 --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 tools/bench_field.py "$label"; }
run "shard 2 streams x1" --in-flight 2 --slots-per-stream 1
run "shard 3 streams x1" --in-flight 3 --slots-per-stream 1
run "shard 2 streams x2" --in-flight 2 --slots-per-stream 2
run "shard 3 streams x2" --in-flight 3 --slots-per-stream 2
