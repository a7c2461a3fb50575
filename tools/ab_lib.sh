#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# A/B of two builds of libwsa on one box: tools/ab_lib.sh [bench args]   (webspeechanalyzer_amd/lib_old = the other build, see tools/README.md "A/B timing of two builds")
run() { label="$1"; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 3 --repeats 5 $BENCH_ARGS 2>/dev/null | python3 tools/bench_field.py "$label"; }
for pass in 1 2 3; do
  run "old" WSA_LIB_DIR=$PWD/webspeechanalyzer_amd/lib_old
  run "new" WSA_AB=1
done
