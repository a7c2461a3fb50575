#!/bin/bash
export WSA_TUNING_ENV=1   # libwsa reads its tuning switches only when this is set (csrc/api.hip Tuning::from_env)
# Co-residency A/B: the pipelined bench with the front end / peak scan / tracker / finalize capped to fewer waves per CU,
# so that kernels of the other steps in flight fit beside them.  Every line = median of 5 timed regions of 100 steps on THIS box.
# usage (GPU box): tools/coresidency_sweep.sh [set] > gpurun_out/coresidency.txt
run() { label="$1"; shift; env "$@" python3 bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 3 --repeats 5 $DEPTH 2>/dev/null | python3 tools/bench_field.py "$label"; }
set=${1:-a}
if [ $set = a ]; then
for d in "" "--in-flight 2" "--in-flight 4"; do
  DEPTH="$d"
  echo "== depth: ${d:-default 3}"
  run "one chunk per WG (r03)  " WSA_FE_NO_QUEUE=1
  run "persistent 4 WG/CU      " WSA_FE_WGS=4
  run "persistent 3 WG/CU      " WSA_FE_WGS=3
  run "persistent 2 WG/CU      " WSA_FE_WGS=2
  run "padded 3 WG/CU          " WSA_FE_WGS=-3
  run "pers 3 + tracker 12     " WSA_FE_WGS=3 WSA_TRACKER_WPC=12
  run "pers 3 + tracker 8      " WSA_FE_WGS=3 WSA_TRACKER_WPC=8
  run "pers 3 + fin 8          " WSA_FE_WGS=3 WSA_FIN_WPC=8
  run "pers 3 + fpw 10         " WSA_FE_WGS=3 WSA_FPW=10
  run "pers 3 + fpw 50         " WSA_FE_WGS=3 WSA_FPW=50
  run "pers 4 + fpw 10         " WSA_FE_WGS=4 WSA_FPW=10
  run "one chunk per WG again  " WSA_FE_NO_QUEUE=1
done
fi
if [ $set = b ]; then
for d in "" "--in-flight 2" "--in-flight 4"; do
  DEPTH="$d"
  echo "== depth: ${d:-default 3}"
  run "one chunk per WG        " WSA_FE_NO_QUEUE=1
  run "persistent 4 WG/CU      " WSA_FE_WGS=4
  run "persistent 3 WG/CU      " WSA_FE_WGS=3
  run "persistent 2 WG/CU      " WSA_FE_WGS=2
  run "padded 3 WG/CU          " WSA_FE_WGS=-3
  run "pers 3 + tracker 8      " WSA_FE_WGS=3 WSA_TRACKER_WPC=8
  run "pers 3 + tracker 12     " WSA_FE_WGS=3 WSA_TRACKER_WPC=12
  run "pers 2 + tracker 12     " WSA_FE_WGS=2 WSA_TRACKER_WPC=12
  run "pers 3 again            " WSA_FE_WGS=3
done
fi
if [ $set = c ]; then
for d in "--in-flight 2" "" "--in-flight 4" "--in-flight 5" "--in-flight 6"; do
  DEPTH="$d"
  echo "== depth: ${d:-default 3}"
  run "one chunk per WG        " WSA_FE_NO_QUEUE=1
  run "persistent 4 WG/CU      " WSA_FE_WGS=4
  run "persistent 3 WG/CU      " WSA_FE_WGS=3
  run "persistent 2 WG/CU      " WSA_FE_WGS=2
  run "pers 2 + tracker 12     " WSA_FE_WGS=2 WSA_TRACKER_WPC=12
  run "pers 3 + tracker 12     " WSA_FE_WGS=3 WSA_TRACKER_WPC=12
done
fi
if [ $set = d ]; then
for d in "--in-flight 2" "--in-flight 3" "--in-flight 4"; do
 for sp in 1 2 3; do
  DEPTH="$d --slots-per-stream $sp"
  echo "== $DEPTH"
  run "one chunk per WG        " WSA_FE_NO_QUEUE=1
  run "persistent 4 WG/CU      " WSA_FE_WGS=4
  run "persistent 3 WG/CU      " WSA_FE_WGS=3
  run "persistent 2 WG/CU      " WSA_FE_WGS=2
 done
done
fi
if [ $set = e ]; then
  DEPTH=""
  echo "== one knob at a time around: persistent 2 WG/CU, 3 streams x 2 planned batches"
  run "base: pers 2           " WSA_FE_WGS=2
  for v in 8 10 12 14; do run "tracker wpc $v          " WSA_FE_WGS=2 WSA_TRACKER_WPC=$v; done
  for v in 8 12 16 24; do run "fin wpc $v              " WSA_FE_WGS=2 WSA_FIN_WPC=$v; done
  for v in 4 6; do run "peaks wpc $v             " WSA_FE_WGS=2 WSA_PEAKS_WPC=$v; done
  for v in 10 15 40 60; do run "fpw $v                  " WSA_FE_WGS=2 WSA_FPW=$v; done
  run "trk 12 + fin 16         " WSA_FE_WGS=2 WSA_TRACKER_WPC=12 WSA_FIN_WPC=16
  run "trk 12 + fin 12         " WSA_FE_WGS=2 WSA_TRACKER_WPC=12 WSA_FIN_WPC=12
  run "trk 12 + fpw 40         " WSA_FE_WGS=2 WSA_TRACKER_WPC=12 WSA_FPW=40
  run "base again              " WSA_FE_WGS=2
  run "GPU_MAX_HW_QUEUES=8     " WSA_FE_WGS=2 GPU_MAX_HW_QUEUES=8
  DEPTH="--in-flight 4"; run "4 streams, 8 hw queues  " WSA_FE_WGS=2 GPU_MAX_HW_QUEUES=8
  DEPTH="--in-flight 5"; run "5 streams, 8 hw queues  " WSA_FE_WGS=2 GPU_MAX_HW_QUEUES=8
fi
