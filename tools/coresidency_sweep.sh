#!/bin/bash
# Co-residency A/B: the pipelined bench with the front end / peak scan / tracker / finalize capped to fewer waves per CU,
# so that kernels of the other steps in flight fit beside them.  Every line = median of 5 timed regions of 100 steps on THIS box.
# usage (GPU box): tools/coresidency_sweep.sh > gpurun_out/coresidency.txt
run() { label="$1"; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 3 --repeats 5 $DEPTH 2>/dev/null | python3 tools/bench_field.py "$label"; }
for d in "" "--in-flight 2" "--in-flight 4"; do
  DEPTH="$d"
  echo "== depth: ${d:-default 3}"
  run "baseline            " WSA_X=0
  run "FE 3 WG/CU          " WSA_FE_WGS=3
  run "FE 2 WG/CU          " WSA_FE_WGS=2
  run "FE 3 + tracker 12   " WSA_FE_WGS=3 WSA_TRACKER_WPC=12
  run "FE 2 + tracker 8    " WSA_FE_WGS=2 WSA_TRACKER_WPC=8
  run "FE 2 + peaks 4      " WSA_FE_WGS=2 WSA_PEAKS_WPC=4
  run "FE 3 + peaks 6      " WSA_FE_WGS=3 WSA_PEAKS_WPC=6
  run "peaks 4             " WSA_PEAKS_WPC=4
  run "fin 8/CU            " WSA_FIN_WPC=8
  run "FE 2 + trk 8 + fin 8" WSA_FE_WGS=2 WSA_TRACKER_WPC=8 WSA_FIN_WPC=8
  run "FE 2+pk 4+trk 8+fin8" WSA_FE_WGS=2 WSA_PEAKS_WPC=4 WSA_TRACKER_WPC=8 WSA_FIN_WPC=8
  run "baseline again      " WSA_X=0
done
