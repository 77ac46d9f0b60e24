"""Copy what tools/round_profiles.sh left under gpurun_out/round into profiles/rNN_* (the newest file of every kind):
python tools/collect_profiles.py r02"""
import csv, glob, os, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = "gpurun_out/round", "profiles"
def newest(pat):
    f = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    if not f: raise SystemExit("missing " + pat)
    return f[-1]
for d, name in [("stats", "kernel_stats"), ("stats_b8192", "b8192_kernel_stats"), ("stats_x25", "x25_kernel_stats"),
                ("stats_c4", "config4_kernel_stats"), ("stats_c5", "config5_kernel_stats")]:
    shutil.copy(newest(d + "/*/*_kernel_stats.csv"), f"{dst}/{tag}_{name}.csv")
for j in ["bench", "bench_config2", "bench_config4", "bench_config5", "bench_under_rocprof", "bench_b8192_under_rocprof",
          "bench_config4_under_rocprof", "bench_config5_under_rocprof"]:
    lines = [l for l in open(f"{src}/{j}.json").read().splitlines() if l.startswith("{")]
    open(f"{dst}/{tag}_{j}.json", "w").write(lines[-1] + "\n")
notes = {"FETCH_SIZE": "KB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): double it",
         "WRITE_SIZE": "KB"}
rows = []
for d in ("pmc_fetch", "pmc_write", "pmc_sq1"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(newest(d + "/*/*_counter_collection.csv"))):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0] if "at::native" not in r["Kernel_Name"] else r["Kernel_Name"], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, cnt[(k, c)], v / cnt[(k, c)], notes.get(c, "sum over the chip's counter instances")))
with open(f"{dst}/{tag}_pmc.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "dispatches", "avg_value", "note"]); w.writerows(rows)
print("wrote", len(rows), "pmc rows")
