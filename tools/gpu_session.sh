#!/bin/bash
# One GPU-box session: the GPU test suite, both bench workloads, then (optional) profiling sessions.
#   usage (inside gpurun): bash tools/gpu_session.sh TAG [tests] [bench] [fuzz] [loader] [prof_lbl] [prof_repwvl]
# A step that is killed (timeout: 124/137) ends the session: no further GPU step is started after it.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
for step in "$@"; do
  case $step in
    tests)
      timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/tests_$TAG.log 2>&1; rc=$?
      tail -5 $O/tests_$TAG.log; echo "tests rc=$rc"; killed $rc && exit 1 ;;
    bench)
      timeout -k 10 400 python bench.py > $O/bench_lbl_$TAG.json 2> $O/bench_lbl_$TAG.err; rc=$?
      cut -c1-300 $O/bench_lbl_$TAG.json; echo "bench lbl rc=$rc"; killed $rc && exit 1
      timeout -k 10 400 python bench.py --workload repwvl100 > $O/bench_repwvl_$TAG.json 2> $O/bench_repwvl_$TAG.err; rc=$?
      cut -c1-300 $O/bench_repwvl_$TAG.json; echo "bench repwvl rc=$rc"; killed $rc && exit 1 ;;
    fuzz)
      timeout -k 10 900 python tools/fuzz_parity.py 2500 404 big > $O/fuzz_$TAG.log 2>&1; rc=$?
      tail -4 $O/fuzz_$TAG.log; echo "fuzz rc=$rc"; killed $rc && exit 1 ;;
    loader)
      timeout -k 10 600 python tools/time_lbl_loader.py /tmp/rcm_lbl_asc > $O/lbl_loader_$TAG.txt 2>&1; rc=$?
      tail -12 $O/lbl_loader_$TAG.txt; echo "loader rc=$rc"; killed $rc && exit 1 ;;
    prof_lbl)
      timeout -k 10 900 bash tools/profile_session.sh ${TAG}l lbl; rc=$?; echo "prof lbl rc=$rc"; killed $rc && exit 1 ;;
    prof_repwvl)
      timeout -k 10 900 bash tools/profile_session.sh ${TAG}r repwvl100; rc=$?; echo "prof repwvl rc=$rc"; killed $rc && exit 1 ;;
  esac
done
exit 0
