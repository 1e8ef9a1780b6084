#!/bin/bash
# TEST INFRASTRUCTURE.  Compiles the reference's own code, UNEDITED and from where it lies, into oracle/_ref/
# (git-ignored; the .so files travel with gpurun).  Called by oracle/Makefile when the reference tree exists.
#
#   build_ref.sh REF_DIR            build into oracle/_ref (or into $RCM_REF_OUT: the test of the hash check uses that)
#   build_ref.sh REF_DIR --rehash   print the hash list for ref_slices.sha256 (after LOOKING at a changed reference)
#
# Slices are addressed by line number, so every slice is hashed first and compared with ref_slices.sha256: a
# reference file whose lines have moved fails HERE, loudly, instead of compiling another body.  oracle/_ref is
# removed before anything is built, so a failed build leaves no stale library behind for the pins to test.
# No reference text is written anywhere: slices exist only on pipes.
set -euo pipefail
REF=${1:?usage: build_ref.sh REF_DIR [--rehash]}
HERE=$(cd "$(dirname "$0")" && pwd)
cd "$HERE"
MAIN=$REF/main.cpp
RT=$REF/repwvl_V2.01_cpp/repwvl_thermal.cpp

slice() { sed -n "$2" "$1"; }            # file, sed line program
section() { awk -v s="$1" '/^\/\/@@ /{on=($2==s); next} on' ref_readtau_shim.inc; }

# name | file | lines: what is compiled ...
SLICES=(
  "main_body|$MAIN|12,21p;25,344p"
  "rt_prelude|$RT|2,4p;6p;10p;13,45p"
  "rt_55_111|$RT|55,111p"
  "rt_124_141|$RT|124,141p"
  "rt_153_168|$RT|153,168p"
  "rt_178_206|$RT|178,206p"
  "rt_210_262|$RT|210,262p"
# ... and what is deliberately NOT (the lines that name netCDF; hashed so that a shifted file cannot hide one)
  "main_skipped_22_23|$MAIN|22,23p"
  "rt_skipped|$RT|5p;7p;11p;49,53p;113,123p;143,149p;171,176p;208p"
  "whole_cplkavg|$REF/cplkavg.cpp|1,\$p"
  "whole_ascii|$REF/lbl.arts/ascii.cpp|1,\$p"
)
hashes() {
  for s in "${SLICES[@]}"; do
    IFS='|' read -r name file prog <<<"$s"
    printf '%s  %s\n' "$(slice "$file" "$prog" | sha256sum | cut -d' ' -f1)" "$name"
  done
}
if [ "${2:-}" = "--rehash" ]; then hashes; exit 0; fi

OUT=${RCM_REF_OUT:-_ref}
rm -rf "$OUT"
if ! diff <(hashes) ref_slices.sha256 >/dev/null; then
  echo "oracle/build_ref.sh: the reference under $REF is not the revision the line slices were cut for:" >&2
  diff <(hashes) ref_slices.sha256 >&2 || true
  echo "nothing built ($OUT removed).  Inspect the reference, re-cut the slices, then --rehash." >&2
  exit 1
fi

mkdir -p "$OUT"
CXX="g++ -O2 -w -fPIC -shared"
$CXX -o "$OUT"/libref_cplkavg.so "$REF/cplkavg.cpp" -lm
$CXX -I"$REF/lbl.arts" -o "$OUT"/libref_ascii.so "$REF/lbl.arts/ascii.cpp" -lm
{ slice "$MAIN" '12,21p;25,344p'; cat ref_main_shim.inc; } | $CXX -x c++ -o "$OUT"/libref_main.so - -lm
{ slice "$RT" '2,4p;6p;10p;13,45p'; cat ref_lowerpos_shim.inc; } | $CXX -x c++ -o "$OUT"/libref_lowerpos.so -
{ slice "$RT" '2,4p;6p;10p;13,45p'; section HEAD; section A
  slice "$RT" '55,111p';  section B
  slice "$RT" '124,141p'; slice "$RT" '153,168p'; section C
  slice "$RT" '178,206p'; section D
  slice "$RT" '210,262p'; section E; } | $CXX -x c++ -o "$OUT"/libref_readtau.so -
echo "built $OUT from $REF"
