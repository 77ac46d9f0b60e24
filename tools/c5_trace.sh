#!/bin/bash
# kernel sequence of one config-5 solve under rocprofv3 (durations in us): bash tools/c5_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c5prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5prof -- python3 $R/bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/c5p.json 2> $R/gpurun_out/c5p.err || exit 1
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob("$R/gpurun_out/c5prof/*/*_kernel_trace.csv"),key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].replace("(anonymous namespace)::","")[:60] for r in rows]
idx=[i for i,n in enumerate(names) if "rows_qp_kernel" in n]
i0=idx[20]
for r,n in zip(rows[i0-6:i0+3],names[i0-6:i0+3]):
    print("%8.1f  %s"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,n))
PY
