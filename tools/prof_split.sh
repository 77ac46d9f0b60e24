# usage: bash tools/prof_split.sh "<dbg values>" [batch] [scale] [keep]   -- kernel averages of tools/exp_split.py under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in $1; do
  rm -rf $R/gpurun_out/ps_$d
  MPCQP_DBG=$d timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_$d -- python3 $R/tools/exp_split.py ${2:-65536} ${3:-1.0} ${4:-0} > $R/gpurun_out/ps_$d.log 2>&1 || { echo FAILED; tail -5 $R/gpurun_out/ps_$d.log; exit 1; }
  echo "== MPCQP_DBG=$d"; grep -E "us per solve" $R/gpurun_out/ps_$d.log | tr '\n' ' '; echo
  python3 $R/tools/kstats.py $R/gpurun_out/ps_$d
done
