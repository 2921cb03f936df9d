#!/bin/bash
# regression check for the intermittent abort (async copy from a dead stack frame in psl_track_iters): the default
# bench shape, no per-launch sync, repeated
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/hunt_E$i.out 2> gpurun_out/hunt_E$i.err; rc=$?
  echo "E run $i rc=$rc $(tail -1 gpurun_out/hunt_E$i.out | cut -c60-90)"
  if [ $rc -ne 0 ]; then tail -2 gpurun_out/hunt_E$i.err; fi
  rm -f gpurun_out/hunt_E$i.err gpurun_out/hunt_E$i.out
done
