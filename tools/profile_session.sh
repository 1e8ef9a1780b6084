#!/bin/bash
# One profiling session on the GPU box: kernel trace + the PMC passes (each its own run, --pmc only) of a bench
# workload, then the bench line itself.
#   usage (inside gpurun): bash tools/profile_session.sh TAG [lbl|repwvl100]
# Afterwards, here: python tools/assemble_profiles.py TAG [lbl|repwvl100]
set -u
TAG=$1
WL=${2:-lbl}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --workload $WL --no-cpu-baseline --no-extra"   # (the extras launch the same kernels at other sizes)
if [ "$WL" = lbl ]; then KT="--steps 4 --warmup 1"; PM="--steps 2 --warmup 1"; CAL="--ncol 4 --steps 2 --warmup 1"
else KT="--steps 200 --warmup 20"; PM="--steps 20 --warmup 5"; CAL="--ncol 1 --steps 20 --warmup 5"; fi
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$TAG -- $B $KT > $O/kt$TAG.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p${TAG}a -- $B $PM > $O/p${TAG}a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/p${TAG}b -- $B $PM > $O/p${TAG}b.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p${TAG}f -- $B $PM > $O/p${TAG}f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p${TAG}w -- $B $PM > $O/p${TAG}w.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p${TAG}g -- $B $PM > $O/p${TAG}g.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p${TAG}f4 -- $B $CAL > $O/p${TAG}f4.log 2>&1 || exit 1
if [ "$WL" = lbl ]; then  # configs[1]: per-kernel times of one column x the full spectrum (60 steps in one rcm_step call)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1$TAG -- python3 $R/tools/trace_single.py > $O/kt1$TAG.log 2>&1 || exit 1
fi
cd $R
timeout -k 10 600 python bench.py --workload $WL > $O/bench_$TAG.json 2> $O/bench_$TAG.err || exit 1
cut -c1-200 $O/bench_$TAG.json
