#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the bench workloads -> profiles/r01_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${wl}_$c
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${wl}_$c -- python3 $R/bench.py --workload $wl --steps 1 --warmup 1 --batch 32 --cpu-sample 0 > $R/gpurun_out/pmc_${wl}_$c.log 2>&1 || { echo "pass $wl $c failed"; tail -3 $R/gpurun_out/pmc_${wl}_$c.log; exit 1; }
    echo "done $wl $c"
  done
  n=64; [ "$wl" = topiq ] && n=96   # bench.py appends a 32-image per-launch pass for the topiq workload
  python3 $R/tools/collect_traffic.py $R/gpurun_out/pmc_${wl}_FETCH_SIZE $R/gpurun_out/pmc_${wl}_WRITE_SIZE $n $wl 1024 && cp $R/profiles/r01_traffic.json $R/gpurun_out/r01_traffic.json
  find $R/gpurun_out/pmc_${wl}_FETCH_SIZE $R/gpurun_out/pmc_${wl}_WRITE_SIZE -name "*kernel_trace.csv" -delete
done
