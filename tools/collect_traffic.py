"""Sum FETCH_SIZE / WRITE_SIZE of two rocprofv3 --pmc runs of bench.py into profiles/r01_traffic.json.
usage: collect_traffic.py <fetch_dir> <write_dir> <images_in_run> <workload> <image_size>"""
import csv, glob, json, os, sys
fd, wd, n, workload, hw = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
def total(d, name):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "r01_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out[workload] = {"fetch_kb_per_image": total(fd, "FETCH_SIZE") / n, "write_kb_per_image": total(wd, "WRITE_SIZE") / n,
                 "image_size": hw, "images_in_run": n,
                 "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workload {workload} --steps 1 --warmup 1 --batch 32 --cpu-sample 0"}
json.dump(out, open(path, "w"), indent=1)
print(out[workload])
