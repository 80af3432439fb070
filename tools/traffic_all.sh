#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE) of bench workloads -> profiles/<tag>_traffic.json (key: workload, or workload_<policy>: what
# bench.py's traffic_bytes() looks up).
# usage: tools/traffic_all.sh <workload>[:<policy>] ...      e.g.  tools/traffic_all.sh full topiq topiq_clip faces ensemble:parity full:parity full:reference_gpu ensemble:bf16
# Counter budget (MI355X_MICROARCH.md, rocprofv3 PMC slots): the TCC block has 4 slots per pass, FETCH_SIZE takes 3 and WRITE_SIZE 2, so
# they cannot share a pass: ONE counter per pass, each pass its own run, --kernel-trace only beside --pmc (what the pool accepts).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${ROUND_TAG:-r03}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  wl=${spec%%:*}; dt=f32; [ "$spec" != "$wl" ] && dt=${spec##*:}
  key=$wl; [ "$dt" != f32 ] && key=${wl}_$dt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${key}_$c
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${key}_$c -- python3 $R/bench.py --workload $wl --dtype $dt --steps 1 --warmup 1 --batch 32 --cpu-sample 0 --no-sub > $R/gpurun_out/pmc_${key}_$c.log 2>&1 || { echo "pass $key $c failed"; tail -3 $R/gpurun_out/pmc_${key}_$c.log; exit 1; }
    echo "done $key $c"
  done
  # images the run pushed through the models: warm-up step + timed step + the per-launch profile pass, 32 each
  python3 $R/tools/collect_traffic.py $R/gpurun_out/pmc_${key}_FETCH_SIZE $R/gpurun_out/pmc_${key}_WRITE_SIZE 96 $key 1024 $TAG "--workload $wl --dtype $dt" && cp $R/profiles/${TAG}_traffic.json $R/gpurun_out/${TAG}_traffic.json
  find $R/gpurun_out/pmc_${key}_FETCH_SIZE $R/gpurun_out/pmc_${key}_WRITE_SIZE -name "*kernel_trace.csv" -delete
done
