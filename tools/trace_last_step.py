"""Developer tool: print the launches of the LAST decode step found in a rocprofv3 kernel-trace CSV (start offset, duration, grid, name)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "vlm_" in r["Kernel_Name"] or "conv_bf16" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(sel) if "argmax_final" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(sel[a]["Start_Timestamp"])
for r in sel[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us +{(e - s) / 1e3:7.1f}  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}  {r['Kernel_Name'][:70]}")
print(f"step total {(int(sel[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
