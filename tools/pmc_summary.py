"""Summarise rocprofv3 --pmc runs: average counter value per kernel.
python tools/pmc_summary.py gpurun_out/pmc_*/ > profiles/rNN_pmc.csv
Each directory holds one pass (rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python bench.py ...)."""
import csv, glob, os, re, sys
from collections import defaultdict

NOTES = {"FETCH_SIZE": "KB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): double it",
         "WRITE_SIZE": "KB; exact for wide stores"}
acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
            name = re.sub(r"\((PolParams|AdmmParams|.*Params)\)$", "", name)
            a = acc[(name, row["Counter_Name"])]
            a[0] += 1; a[1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "avg_value", "note"])
for (k, c), (n, v) in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    w.writerow([k, c, n, v / n, NOTES.get(c, "SQ_* cycle counters are in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles)" if c.startswith("SQ_") else "")])
