"""Builds librcm_hip.so (HIP kernels + C ABI + host pieces) and the rcm_main driver, in-tree,
for gfx950.  hipcc cross-compiles without a GPU.  Usage: python -m our_first_climate_model_amd.build"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librcm_hip.so")
MAIN = os.path.join(PKG, "rcm_main")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["rcm_capi.hip", "rcm_host.cpp", "rcm_nc.cpp"]
HEADERS = ["rcm_kernels.hpp", "rcm_device.hpp", "rcm_internal.hpp", "rcm_exp_coef.hpp", "rcm_variants.hpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


VARIANTS_LIB = os.path.join(ROOT, "tests", "_variants", "librcm_hip_variants.so")


def _sources():
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "rcm.h")]
    exports = os.path.join(CSRC, "rcm_exports.map")  # the C ABI of include/rcm.h is ALL the library exports
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
              "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wl,--version-script=" + exports]
    return srcs, deps + [exports], common


def build_variants(force=False):
    """TEST-SIDE build of the same sources with -DRCM_WITH_VARIANTS: adds the one-lane-per-point reference
    kernels (library exp, branch-for-branch cplkavg, no cube chains) and their selection by environment
    variable.  Lives under tests/_variants/ and is loaded only by the tests that cross-check the production
    kernels against them; the production library contains neither."""
    srcs, deps, common = _sources()
    if force or _newer(VARIANTS_LIB, deps):
        os.makedirs(os.path.dirname(VARIANTS_LIB), exist_ok=True)
        subprocess.run([HIPCC] + common + ["-DRCM_WITH_VARIANTS", "-shared", "-o", VARIANTS_LIB] + srcs, check=True)
    return VARIANTS_LIB


def build(force=False, verbose=False):
    srcs, deps, common = _sources()
    if force or _newer(LIB, deps):
        cmd = [HIPCC] + common + ["-shared", "-o", LIB] + srcs
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
    main_src = os.path.join(CSRC, "rcm_main.cpp")
    if os.path.exists(main_src) and (force or _newer(MAIN, [main_src, LIB])):
        cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", MAIN, main_src,
               "-L" + PKG, "-lrcm_hip", "-Wl,-rpath,$ORIGIN"]
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        if os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):  # --devices: one RCCL all-gather
            cmd += ["-DRCM_WITH_RCCL", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                    "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
