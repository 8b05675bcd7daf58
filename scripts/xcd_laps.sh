#!/bin/bash
# Where an iteration of the persistent launch goes, phase by phase (the stand-in for an instruction-level trace: the image has no ATT decoder).
# Needs `make -C abip_amd/csrc prof` (the library with the lap counters compiled in).  Output: gpurun_out/laps/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/laps
export ABIP_HIP_LIBRARY=$PWD/abip_amd/lib/libabip_hip_prof.so
for wl in c2 c3; do
  python bench.py --workload $wl --no-cpu --no-extra > gpurun_out/laps/${wl}_bench.json 2> gpurun_out/laps/${wl}_laps_raw.txt
  grep "xcd prof" gpurun_out/laps/${wl}_laps_raw.txt | tail -2 > gpurun_out/laps/${wl}_laps.txt
done
