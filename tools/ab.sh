#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# A/B of environment switches on one box: tools/ab.sh "LABEL_A:VAR=v VAR2=v" "LABEL_B:..." ...   (each run = median of 5 regions of 100 steps; the list is run twice)
run() { label="$1"; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 3 --repeats 5 $BENCH_ARGS 2>/dev/null | python3 tools/bench_field.py "$label"; }
for pass in 1 2; do
  for spec in "$@"; do
    label="${spec%%:*}"; vars="${spec#*:}"
    run "$label" WSA_AB=1 $vars
  done
done
