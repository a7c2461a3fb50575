#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# A/B of the K5 occupancy cap (GPU box, from the repository root): rebuilds coeffs.hip with amdgpu_waves_per_eu(w, w), w = 0 (uncapped), 2, 3, 4, and runs the level-12 bench; restores the source
export TMPDIR=/tmp
cp webspeechanalyzer_amd/csrc/coeffs.hip /tmp/coeffs_orig.hip
# the source is edited in place: put it back however the script ends (an interrupted run must not leave a modified kernel behind)
trap 'cp /tmp/coeffs_orig.hip webspeechanalyzer_amd/csrc/coeffs.hip' EXIT
for w in 0 2 3 4; do
  cp /tmp/coeffs_orig.hip webspeechanalyzer_amd/csrc/coeffs.hip
  if [ $w != 0 ]; then sed -i "s/__global__ __launch_bounds__(64) void coeffs_kernel(CoefParams p) {/__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu($w, $w))) void coeffs_kernel(CoefParams p) {/" webspeechanalyzer_amd/csrc/coeffs.hip; fi
  make -s -C webspeechanalyzer_amd/csrc > /dev/null 2>&1
  echo "waves_per_eu $w:"
  python3 bench.py --level 12 --no-cpu-baseline --no-extra | python3 tools/bench_field.py level12
done
cp /tmp/coeffs_orig.hip webspeechanalyzer_amd/csrc/coeffs.hip
