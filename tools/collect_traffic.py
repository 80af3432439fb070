"""Sum FETCH_SIZE / WRITE_SIZE of two rocprofv3 --pmc runs of bench.py into profiles/<tag>_traffic.json.
usage: collect_traffic.py <fetch_dir> <write_dir> <images_in_run> <key> <image_size> <round tag> "<bench flags>"
The face graphs' launches (workloads `full` / `faces`) are part of the same run and are counted: the figure is per image of the
whole step. The per-launch profile pass of bench.py covers the ensemble models only, so for those two workloads the per-image
figure is a slight under-estimate of the face stage's share (two of three passes carry it)."""
import csv, glob, json, os, sys
fd, wd, n, key, hw = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
tag = sys.argv[6] if len(sys.argv) > 6 else "r02"
flags = sys.argv[7] if len(sys.argv) > 7 else f"--workload {key}"
def total(d, name):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", f"{tag}_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out[key] = {"fetch_kb_per_image": total(fd, "FETCH_SIZE") / n, "write_kb_per_image": total(wd, "WRITE_SIZE") / n,
            "image_size": hw, "images_in_run": n,
            "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes, one TCC counter each) -- python3 bench.py {flags} --steps 1 --warmup 1 --batch 32 --cpu-sample 0 --no-sub"}
json.dump(out, open(path, "w"), indent=1)
print(key, out[key])
