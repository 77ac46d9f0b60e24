set -o pipefail
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c4prof -- python3 $R/bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/c4prof.json 2> $R/gpurun_out/c4prof.err || exit 1
echo ok
