#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
run() { label="$1"; shift; env "$@" python3 bench.py --clips 12500 --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline --no-extra $DEPTH 2>/dev/null | python3 tools/bench_field.py "$label"; }
DEPTH=""
run "shard no queue   " WSA_FE_NO_QUEUE=1
run "shard pers 4     " WSA_FE_WGS=4
run "shard pers 3     " WSA_FE_WGS=3
run "shard pers 2     " WSA_FE_WGS=2
DEPTH="--slots-per-stream 1"
run "shard pers 2 spp1" WSA_FE_WGS=2
run "shard pers 4 spp1" WSA_FE_WGS=4
DEPTH="--in-flight 3"
run "shard pers 2 3str" WSA_FE_WGS=2
