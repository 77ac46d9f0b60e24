#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU session: bash tools/round_profiles.sh   (from the repository root)
# Each GPU step has its own time limit and the script stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
# step NAME OUT ERR CMD...: run CMD with stdout -> OUT, stderr -> ERR; stop the script if it fails
step() { local name=$1 out=$2 err=$3; shift 3; echo "== $name"; "$@" > "$out" 2> "$err" || { echo "FAILED ($name): $*"; tail -5 "$err"; exit 1; }; }
B="$R/bench.py"
step bench $O/bench.json $O/bench.err timeout -k 10 300 python3 $B
step bench_c2 $O/bench_config2.json $O/bench_c2.err timeout -k 10 200 python3 $B --config 2 --no-cpu-baseline
step bench_c4 $O/bench_config4.json $O/bench_c4.err timeout -k 10 300 python3 $B --config 4 --no-cpu-baseline
step bench_c5 $O/bench_config5.json $O/bench_c5.err timeout -k 10 300 python3 $B --config 5 --no-cpu-baseline
step stats $O/bench_under_rocprof.json $O/stats.err timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --steps 20 --warmup 3 --no-cpu-baseline --no-shard-sweep
step stats_8192 $O/bench_b8192_under_rocprof.json $O/stats8192.err timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b8192 -- python3 $B --batch 8192 --steps 50 --warmup 3 --no-cpu-baseline --no-shard-sweep
step stats_x25 $O/x25.log $O/x25.log.err timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_x25 -- python3 $R/tools/scale_run.py 2.5 65536 10
step stats_c4 $O/bench_config4_under_rocprof.json $O/statsc4.err timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $B --config 4 --steps 50 --warmup 5 --no-cpu-baseline
step stats_c5 $O/bench_config5_under_rocprof.json $O/statsc5.err timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $B --config 5 --steps 20 --warmup 3 --no-cpu-baseline
P="python3 $B --steps 20 --warmup 3 --no-cpu-baseline --no-shard-sweep"
step pmc_fetch $O/pmc1.log $O/pmc1.log.err timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P
step pmc_write $O/pmc2.log $O/pmc2.log.err timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P
step pmc_sq1 $O/pmc3.log $O/pmc3.log.err timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- $P
echo ALL DONE
