#!/bin/bash
# Static look at the two production RT kernels of a build (no GPU needed): ISA lines, register copies, scratch
# accesses, scalar vs vector loads, VGPRs.  A kernel whose s_load count drops and whose global_load count rises has
# lost the compiler's proof that global memory is not clobbered (a volatile asm, a scheduling builtin, very many LDS
# stores): its schedule and weights are then fetched by vector loads with a full wait each.
#   usage: tools/isa_stats.sh NAME [extra hipcc flags, e.g. -DRCM_PRIO=0]      (writes gpurun_out/isa/NAME*.s)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/isa
mkdir -p $O
n=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$R/include -I$R/our_first_climate_model_amd/csrc -S \
  --cuda-device-only -o $O/$n.s $R/our_first_climate_model_amd/csrc/rcm_capi.hip "$@" 2>/dev/null || exit 1
# the kernels by the start of their mangled names: k_rt_repwvl_flat<CUBE, NACT> has one instantiation per species count
# (NACT = 3: the CO2-sweep ensembles of the benchmarks), so the symbols are looked up, not spelled out
ks=$(grep -oE '^_ZN3rcm(13k_rt_lbl_pairILi1E|16k_rt_repwvl_flatILi1ELi3E)[A-Za-z0-9_]*' $O/$n.s | sort -u)
[ -n "$ks" ] || { echo "isa_stats: no production RT kernel found in $O/$n.s" >&2; exit 1; }
for k in $ks; do
  awk "/^$k:/,/s_endpgm/" $O/$n.s > $O/$n.$k.s
  echo "$n ${k:7:24} lines:$(wc -l < $O/$n.$k.s) mov:$(grep -c 'v_mov_b' $O/$n.$k.s) scratch:$(grep -c 'scratch_' $O/$n.$k.s)" \
       "s_load:$(grep -c 's_load' $O/$n.$k.s) global_load:$(grep -c 'global_load' $O/$n.$k.s)" \
       "vgpr:$(grep -A40 "^\s*.amdhsa_kernel $k" $O/$n.s | grep next_free_vgpr | awk '{print $2}')"
done
