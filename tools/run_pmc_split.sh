cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 2 3; do
MPCQP_DBG=$d timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmcd$d -- python3 $R/tools/exp_split.py 65536 1.0 0 > /dev/null 2>&1 || exit 1
( cd $R && python3 tools/pmc_summary.py gpurun_out/pmcd$d > gpurun_out/pmc_d$d.csv )
done
