set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p1r4a -- python3 $R/tools/trace_single.py > $O/p1r4a.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p1r4g -- python3 $R/tools/trace_single.py > $O/p1r4g.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1r4f -- python3 $R/tools/trace_single.py > $O/p1r4f.log 2>&1 || exit 1
echo done
