#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU session: bash tools/round_profiles.sh   (from the repository root)
# then, back in the container: python tools/collect_profiles.py r03
# Each GPU step has its own time limit and the script stops at the first failure.  Counter passes: one block's budget per
# pass (8 SQ_, 4 TCC_: FETCH_SIZE takes 3, WRITE_SIZE 2), never together with a trace.
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
step bench_nosplit $O/bench_nosplit.json $O/bench_nosplit.err timeout -k 10 300 python3 $B --no-split --no-cpu-baseline
step bench_b8192 $O/bench_b8192.json $O/bench_b8192.err timeout -k 10 300 python3 $B --batch 8192 --no-cpu-baseline --no-shard-sweep
step bench_weak $O/bench_weak.json $O/bench_weak.err timeout -k 10 300 python3 $B --scaling weak --no-cpu-baseline --no-shard-sweep
step bench_c2 $O/bench_config2.json $O/bench_c2.err timeout -k 10 200 python3 $B --config 2 --no-cpu-baseline
step bench_c4 $O/bench_config4.json $O/bench_c4.err timeout -k 10 300 python3 $B --config 4 --no-cpu-baseline
step bench_c5 $O/bench_config5.json $O/bench_c5.err timeout -k 10 300 python3 $B --config 5 --no-cpu-baseline
T="rocprofv3 --kernel-trace --stats --output-format csv"
step stats $O/bench_under_rocprof.json $O/stats.err timeout -k 10 300 $T -d $O/stats -- python3 $B --steps 20 --warmup 3 --no-cpu-baseline --no-shard-sweep
step stats_b8192 $O/bench_b8192_under_rocprof.json $O/stats8192.err timeout -k 10 300 $T -d $O/stats_b8192 -- python3 $B --batch 8192 --steps 50 --warmup 3 --no-cpu-baseline --no-shard-sweep
step stats_x25 $O/x25.log $O/x25.log.err timeout -k 10 300 $T -d $O/stats_x25 -- python3 $R/tools/scale_run.py 2.5 65536 10
step stats_c4 $O/bench_config4_under_rocprof.json $O/statsc4.err timeout -k 10 300 $T -d $O/stats_c4 -- python3 $B --config 4 --steps 50 --warmup 5 --no-cpu-baseline
step stats_c5 $O/bench_config5_under_rocprof.json $O/statsc5.err timeout -k 10 300 $T -d $O/stats_c5 -- python3 $B --config 5 --steps 20 --warmup 3 --no-cpu-baseline
C="rocprofv3 --output-format csv"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"
P="python3 $B --steps 20 --warmup 3 --no-cpu-baseline --no-shard-sweep"
# the headline (component split), its set-up kernels included (assemble_kernel, gemm_nt_kernel, split_key_kernel: MFMA rows)
step pmc_fetch $O/pmc1.log $O/pmc1.err timeout -s KILL 120 $C --pmc FETCH_SIZE -d $O/pmc_fetch -- $P
step pmc_write $O/pmc2.log $O/pmc2.err timeout -s KILL 120 $C --pmc WRITE_SIZE -d $O/pmc_write -- $P
step pmc_sq1 $O/pmc3.log $O/pmc3.err timeout -s KILL 120 $C --pmc $SQ1 -d $O/pmc_sq1 -- $P
step pmc_sq2 $O/pmc4.log $O/pmc4.err timeout -s KILL 120 $C --pmc $SQ2 -d $O/pmc_sq2 -- $P
# the same batch solved whole (--no-split): cold_key_kernel + the whole-instance cascade
step pmc_ns_fetch $O/pmc5.log $O/pmc5.err timeout -s KILL 120 $C --pmc FETCH_SIZE -d $O/pmc_ns_fetch -- $P --no-split
step pmc_ns_write $O/pmc6.log $O/pmc6.err timeout -s KILL 120 $C --pmc WRITE_SIZE -d $O/pmc_ns_write -- $P --no-split
step pmc_ns_sq1 $O/pmc7.log $O/pmc7.err timeout -s KILL 120 $C --pmc $SQ1 -d $O/pmc_ns_sq1 -- $P --no-split
step pmc_ns_sq2 $O/pmc8.log $O/pmc8.err timeout -s KILL 120 $C --pmc $SQ2 -d $O/pmc_ns_sq2 -- $P --no-split
# configs 4 and 5: HBM traffic and the MFMA rows of factor_wave_kernel / assemble / the rows path
P4="python3 $B --config 4 --steps 30 --warmup 3 --no-cpu-baseline"
P5="python3 $B --config 5 --steps 10 --warmup 2 --no-cpu-baseline"
step pmc_c4_fetch $O/pmc9.log $O/pmc9.err timeout -s KILL 120 $C --pmc FETCH_SIZE -d $O/pmc_c4_fetch -- $P4
step pmc_c4_write $O/pmc10.log $O/pmc10.err timeout -s KILL 120 $C --pmc WRITE_SIZE -d $O/pmc_c4_write -- $P4
step pmc_c4_sq1 $O/pmc11.log $O/pmc11.err timeout -s KILL 120 $C --pmc $SQ1 -d $O/pmc_c4_sq1 -- $P4
step pmc_c4_sq2 $O/pmc12.log $O/pmc12.err timeout -s KILL 120 $C --pmc $SQ2 -d $O/pmc_c4_sq2 -- $P4
step pmc_c5_fetch $O/pmc13.log $O/pmc13.err timeout -s KILL 180 $C --pmc FETCH_SIZE -d $O/pmc_c5_fetch -- $P5
step pmc_c5_write $O/pmc14.log $O/pmc14.err timeout -s KILL 180 $C --pmc WRITE_SIZE -d $O/pmc_c5_write -- $P5
step pmc_c5_sq2 $O/pmc15.log $O/pmc15.err timeout -s KILL 180 $C --pmc $SQ2 -d $O/pmc_c5_sq2 -- $P5
echo ALL DONE
