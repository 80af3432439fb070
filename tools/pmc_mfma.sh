#!/bin/bash
# Matrix-core utilisation of a bench workload from one PMC pass -> gpurun_out/<tag>_<workload>[_bf16]_mfma_busy.summary.txt
# usage: tools/pmc_mfma.sh <workload> [policy: f32 | parity | fast16 | bf16 | reference_gpu | f16]
# Counter budget (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ has 8 slots per pass, GRBM 2, independent of each other - this pass takes
# 4 SQ counters + 1 GRBM counter and nothing from TCC; it runs with --kernel-trace only (no other trace domain beside --pmc).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-topiq}
DT=${2:-f32}
TAG=${ROUND_TAG:-r03}
KEY=$WL; [ "$DT" != f32 ] && KEY=${WL}_$DT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mfma
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --workload $WL --dtype $DT --steps 1 --warmup 1 --batch 32 --cpu-sample 0 --no-sub > $R/gpurun_out/pmc_mfma.log 2>&1 || { echo "pass failed"; tail -5 $R/gpurun_out/pmc_mfma.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_mfma/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        calls[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])
tot_b = sum(v["SQ_BUSY_CYCLES"] for _, v in rows); tot_m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for _, v in rows)
tot_g = sum(v["GRBM_GUI_ACTIVE"] for _, v in rows)
# GRBM_GUI_ACTIVE is summed over the 8 XCDs (micro-architecture guide): /8 = elapsed shader cycles of the launches. SQ_VALU_MFMA_BUSY_CYCLES
# sums over the chip; UNITS = how many matrix pipes that sum runs over, calibrated in the summary line against the MFMA count the
# engine reports (one v_mfma_f32_32x32x2_f32 = 4096 FLOP = 64 cycles of one SIMD's pipe).
import json, re
line = [l for l in open("$R/gpurun_out/pmc_mfma.log") if l.startswith("{")][-1]
bench = json.loads(line)
exec_flops = bench["roofline"]["executed_flops_per_image"] * 96      # three passes of 32 images: warm-up, timed, per-launch profile
# one v_mfma_f32_32x32x2_f32 = 4096 FLOP and 64 cycles of its SIMD's pipe; one v_mfma_f32_32x32x16_{bf16,f16} = 32768 FLOP and 32 cycles
by = bench["roofline"].get("executed_flops_per_image_by_dtype")
if by:      # a policy that mixes fp32 and 2-byte models: priced per dtype
    expect_simd_cycles = by["f32"] * 96 / 4096 * 64 + by["2-byte"] * 96 / 32768 * 32
else:
    expect_simd_cycles = exec_flops / 32768 * 32 if isinstance(bench["roofline"]["peak"], (int, float)) and bench["roofline"]["peak"] > 1000 else exec_flops / 4096 * 64
out = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --workload $WL --dtype $DT --steps 1 --warmup 1 --batch 32 --cpu-sample 0 --no-sub",
       "# per kernel (summed over launches): launches, share of elapsed cycles, matrix pipes busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8) / 1024 SIMDs, issue-stall share SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES",
       f"# whole run: sum SQ_VALU_MFMA_BUSY_CYCLES {tot_m:.4g}; MFMA pipe cycles implied by the engine's executed-FLOP counter {expect_simd_cycles:.4g} (ratio {tot_m / expect_simd_cycles:.3f});",
       f"#            elapsed shader cycles of all launches (GRBM_GUI_ACTIVE / 8) {tot_g / 8:.4g} -> matrix pipes busy {tot_m / (tot_g / 8) / 1024:.3f} of the elapsed cycles"]
for k, v in rows[:14]:
    b = v["SQ_BUSY_CYCLES"]
    g = max(v["GRBM_GUI_ACTIVE"] / 8, 1)
    out.append(f"{k[:70]:70s} x{calls[k]:4d}  {v['GRBM_GUI_ACTIVE'] / tot_g * 100:5.1f} % of cycles   matrix pipes busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / g / 1024:.3f} (of 1024)   issue-stall {v['SQ_WAIT_INST_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.3f}")
open("$R/gpurun_out/${TAG}_${KEY}_mfma_busy.summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $R/gpurun_out/pmc_mfma -name "*kernel_trace.csv" -delete
