"""Print kernel averages from a rocprofv3 --stats csv directory: python tools/kstats.py DIR [pattern...]"""
import csv, glob, os, re, sys
pats = sys.argv[2:] or ["polish_oct", "split_", "cold_key", "polish_box", "fillBuffer"]
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])
        if any(p in n for p in pats):
            print(f"  {n[:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
