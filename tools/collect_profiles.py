"""Copy what tools/round_profiles.sh left under gpurun_out/round into profiles/rNN_* (the newest file of every kind) and
derive profiles/rNN_pmc_headline.json (what bench.py's roofline block quotes from the counter passes):
python tools/collect_profiles.py r03"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = "gpurun_out/round", "profiles"


def newest(pat, required=True):
    f = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    if not f:
        if required:
            raise SystemExit("missing " + pat)
        return None
    return f[-1]


for d, name in [("stats", "kernel_stats"), ("stats_b8192", "b8192_kernel_stats"), ("stats_x25", "x25_kernel_stats"),
                ("stats_c4", "config4_kernel_stats"), ("stats_c5", "config5_kernel_stats")]:
    shutil.copy(newest(d + "/*/*_kernel_stats.csv"), f"{dst}/{tag}_{name}.csv")
for j in ["bench", "bench_nosplit", "bench_b8192", "bench_weak", "bench_config2", "bench_config4", "bench_config5", "bench_under_rocprof",
          "bench_b8192_under_rocprof", "bench_config4_under_rocprof", "bench_config5_under_rocprof"]:
    lines = [l for l in open(f"{src}/{j}.json").read().splitlines() if l.startswith("{")]
    open(f"{dst}/{tag}_{j}.json", "w").write(lines[-1] + "\n")

notes = {"FETCH_SIZE": "KB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): double it",
         "WRITE_SIZE": "KB", "SQ_VALU_MFMA_BUSY_CYCLES": "cycles (the other SQ cycle counters are quad-cycles)"}


def kname(s):
    if "at::native" in s:
        return s
    s = s.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\((PolParams|[A-Za-z]*Params|mpcqp_[a-z_]*\*).*$", "", s)


MEDIAN = {}      # (run dir, kernel, counter) -> median over the dispatches (the bench also times the other keep_state mode: the
                 # mean of FETCH / WRITE mixes the two, the median is the timed mode's)


def load(d):
    f = newest(d + "/*/*_counter_collection.csv", required=False)
    vals = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            vals[(kname(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        MEDIAN[k] = sorted(v)[len(v) // 2]
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}


rows = []
sets = {"headline (component split)": ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"),
        "headline solved whole (--no-split)": ("pmc_ns_fetch", "pmc_ns_write", "pmc_ns_sq1", "pmc_ns_sq2"),
        "config 4": ("pmc_c4_fetch", "pmc_c4_write", "pmc_c4_sq1", "pmc_c4_sq2"),
        "config 5": ("pmc_c5_fetch", "pmc_c5_write", "pmc_c5_sq2")}
tables = {}
for run, dirs in sets.items():
    t = {}
    for d in dirs:
        t.update(load(d))
    tables[run] = t
    for (k, c), (v, n) in sorted(t.items()):
        rows.append((run, k, c, n, v, MEDIAN[(k, c)], notes.get(c, "sum over the chip's counter instances")))
    MEDIANS = dict(MEDIAN); tables[run + " median"] = {k: (MEDIANS[k], 0) for k in t}; MEDIAN.clear()
with open(f"{dst}/{tag}_pmc.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["run", "kernel", "counter", "dispatches", "avg_value", "median_value", "note"]); w.writerows(rows)
print("wrote", len(rows), "pmc rows")


def stats_ns(name_part, d="stats"):
    """average duration (ns) of the kernel whose name contains name_part, from the kernel-trace run"""
    for r in csv.DictReader(open(newest(d + "/*/*_kernel_stats.csv"))):
        if re.search(name_part, r["Name"].replace("(anonymous namespace)::", "")):
            return float(r["AverageNs"])
    return None


def headline(run, kernel_part, stats_dir, source):
    t = tables[run]
    ks = sorted({k for (k, c) in t if re.search(kernel_part, k)})
    if not ks:
        return None
    k = ks[0]
    g = lambda c: t.get((k, c), (None, 0))[0]
    gm = lambda c: tables[run + " median"].get((k, c), (None, 0))[0]
    out = {"kernel": k, "source": source + "; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, medians over the dispatches (keep_state = 0, the timed mode)"}
    if gm("FETCH_SIZE") is not None and gm("WRITE_SIZE") is not None:
        out["hbm_bytes_per_launch"] = (2.0 * gm("FETCH_SIZE") + gm("WRITE_SIZE")) * 1024.0
        out["fetch_kb"], out["write_kb"] = gm("FETCH_SIZE"), gm("WRITE_SIZE")
    ns = stats_ns(kernel_part, stats_dir)
    if ns and g("SQ_ACTIVE_INST_VALU") is not None:
        cycles = ns * 2.4                                   # shader clock 2.4 GHz
        out["launch_ns"] = ns
        out["valu_active_per_simd"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * cycles)
    if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
        out["waves_per_simd"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES") / 8.0
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES"):
        if g(c) is not None:
            out[c] = g(c)
    return out


hl = {"split": headline("headline (component split)", r"polish_oct_kernel<8, \d, 8, 512, true, true, 4>", "stats",
                        f"profiles/{tag}_pmc.csv, run 'headline (component split)': first active-set pass over the components, 65,536 instances"),
      "unsplit": headline("headline solved whole (--no-split)", r"polish_oct_kernel<8, 8, 16, 512, true", "stats",
                          f"profiles/{tag}_pmc.csv, run 'headline solved whole (--no-split)': first active-set pass, 65,536 instances")}
if hl["unsplit"] is not None:
    hl["unsplit"].pop("launch_ns", None); hl["unsplit"].pop("valu_active_per_simd", None)   # no kernel-trace run of the unsplit chain in this set
json.dump(hl, open(f"{dst}/{tag}_pmc_headline.json", "w"), indent=1)
print(json.dumps(hl, indent=1))
