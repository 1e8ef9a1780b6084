#!/bin/bash
# Instruction counters of the RT kernel for two library builds (product and a variant), same small workload.
#   usage (inside gpurun): bash tools/pmc_compare.sh VARIANT_NAME [ncol=512] [lbl|repwvl100]
set -u
V=$1
N=${2:-512}
WL=${3:-lbl}
if [ $WL = lbl ]; then ST="--steps 2 --warmup 1"; else ST="--steps 20 --warmup 5"; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out
B="python3 $R/bench.py --workload $WL --no-cpu-baseline --ncol $N $ST"
rm -rf $O/pc_product_* $O/pc_${V}_*
for which in product $V; do
  if [ $which = product ]; then unset RCM_LIB_PATH; else export RCM_LIB_PATH=$R/our_first_climate_model_amd/variants/librcm_hip_$which.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pc_${which}_a -- $B > $O/pc_${which}_a.log 2>&1 || exit 1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pc_${which}_g -- $B > $O/pc_${which}_g.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $O/pc_${which}_b -- $B > $O/pc_${which}_b.log 2>&1 || exit 1
done
echo done
