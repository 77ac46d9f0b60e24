set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s25 -- python3 $R/tools/scale_run.py 2.5 65536 10 > $R/gpurun_out/s25.log 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b8192 -- python3 $R/tools/scale_run.py 1.0 8192 50 > $R/gpurun_out/b8192.log 2>&1 || exit 3
echo done
