#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# A/B of environment switches at the driver's region length and at 100 steps: tools/ab_k.sh "LABEL:VAR=v ..." ...   (median of 9 regions each; the list is run twice)
run() { label="$1"; k="$2"; shift 2; env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps $k --warmup 5 --repeats 9 $BENCH_ARGS 2>/dev/null | python3 tools/bench_field.py "$label/K=$k"; }
for pass in 1 2; do
  for spec in "$@"; do
    label="${spec%%:*}"; vars="${spec#*:}"
    run "$label" 20 WSA_AB=1 $vars
    run "$label" 100 WSA_AB=1 $vars
  done
done
