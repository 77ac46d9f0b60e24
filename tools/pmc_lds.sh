# LDS counters of the bench's kernels (one --pmc pass): bash tools/pmc_lds.sh [bench.py arguments]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/lds; mkdir -p $O; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-shard-sweep "$@" > $O/log 2> $O/err || { tail -5 $O/err; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
f=glob.glob(R+"/gpurun_out/lds/pmc/**/*counter_collection.csv", recursive=True); f=max(f, key=os.path.getmtime)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_BUSY_CYCLES": n[k]+=1
for k,v in acc.items():
    if v["SQ_BUSY_CYCLES"] * n[k] > 2e7:
        print(k, n[k], {c: round(x/max(n[k],1)) for c,x in v.items()})
PY
