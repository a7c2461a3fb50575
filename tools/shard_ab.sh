#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# A/B on the 12 500-clip shard (BASELINE config 4 per GPU): tools/shard_ab.sh "LABEL:VAR=v ..." ...
run() { label="$1"; shift; env "$@" python3 bench.py --clips 12500 --steps 8 --warmup 2 --repeats 3 --no-cpu-baseline --no-extra $BENCH_ARGS 2>/dev/null | python3 tools/bench_field.py "$label"; }
for spec in "$@"; do label="${spec%%:*}"; vars="${spec#*:}"; run "$label" WSA_AB=1 $vars; done
