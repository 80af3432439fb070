#!/bin/bash
# One profiling session of a round on the GPU box: kernel-trace stats, per-layer / per-GEMM tables, MFMA-busy and HBM-traffic PMC passes.
# Everything lands under gpurun_out/ (and profiles/<tag>_traffic.json through collect_traffic.py); copy what is to be judged into profiles/.
# usage (on the box): bash tools/profile_round.sh [stats] [tables] [mfma] [traffic]      (default: all four)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export ROUND_TAG=${ROUND_TAG:-r03}
T=$ROUND_TAG
what=${*:-stats tables mfma traffic}
cd /tmp && export TMPDIR=/tmp && cd $R
if [[ $what == *stats* ]]; then
  for spec in full:f32 ensemble:parity ensemble:bf16; do
    wl=${spec%%:*}; pol=${spec##*:}
    rm -rf gpurun_out/ks_${wl}_$pol
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_${wl}_$pol -- python3 bench.py --workload $wl --dtype $pol --steps 2 --warmup 1 --batch 64 --cpu-sample 0 --no-sub > gpurun_out/ks_${wl}_$pol.log 2>&1
    f=$(find gpurun_out/ks_${wl}_$pol -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f gpurun_out/${T}_kernel_stats_${wl}_${pol}_b64.csv
    grep -h "^{" gpurun_out/ks_${wl}_$pol.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $pol under the profiler:', d['value'], 'images/s')"
    find gpurun_out/ks_${wl}_$pol -name "*kernel_trace.csv" -delete
  done
fi
if [[ $what == *tables* ]]; then
  python3 tools/perf_topiq.py 64 32 1024 score > gpurun_out/${T}_per_layer_topiq_f32_mb32.txt 2>&1
  python3 tools/perf_topiq.py 64 32 1024 score f16 > gpurun_out/${T}_per_layer_topiq_f16_mb32.txt 2>&1
  python3 tools/perf_clip.py > gpurun_out/${T}_per_gemm_clip_f32.txt 2>&1
  python3 tools/perf_clip.py f16 > gpurun_out/${T}_per_gemm_clip_f16.txt 2>&1
  python3 tools/perf_clip.py f16x3 > gpurun_out/${T}_per_gemm_clip_f16x3.txt 2>&1
  python3 tools/perf_clip_policies.py > gpurun_out/${T}_clip_policies.txt 2>&1
  python3 tools/perf_samp.py 64 > gpurun_out/${T}_per_layer_samp_f32.txt 2>&1
  python3 tools/perf_vlm.py --layers 4 --batches 1,2,4,8,32 > gpurun_out/${T}_vlm_perf.txt 2>&1
  head -3 gpurun_out/${T}_per_layer_topiq_f16_mb32.txt; head -2 gpurun_out/${T}_per_gemm_clip_f16x3.txt; cat gpurun_out/${T}_clip_policies.txt
fi
if [[ $what == *mfma* ]]; then
  bash tools/pmc_mfma.sh full f32 | head -8
  bash tools/pmc_mfma.sh ensemble parity | head -8
fi
if [[ $what == *traffic* ]]; then
  bash tools/traffic_all.sh full topiq topiq_clip faces ensemble:parity full:parity full:reference_gpu ensemble:fast16 ensemble:bf16
fi
