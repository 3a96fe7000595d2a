#!/bin/bash
# tools/gen_golden/gen_golden.sh -- BUILD CONTAINER ONLY (needs /root/reference).  Compiles the reference's headers
# (where they lie, nothing is copied) over include/ac_types and regenerates tests/golden/ref_hdr/*.json.
# The binaries live in /tmp; only the JSON vectors (data) are committed.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
REF=${ACDSP_REFERENCE:-/root/reference}
[ -d "$REF/include/ac_dsp" ] || { echo "reference tree not found at $REF"; exit 1; }
B=${TMPDIR:-/tmp}/acdsp_gen_golden
OUT=$R/tests/golden/ref_hdr
mkdir -p "$B" "$OUT"
CXX="g++ -std=c++11 -O1 -Wno-unknown-pragmas -I$REF/include -I$R/include/ac_types -I$R/tools/gen_golden"
$CXX "$R/tools/gen_golden/gen_fir.cpp" -o "$B/gen_fir"
$CXX "$R/tools/gen_golden/gen_cfg1.cpp" -o "$B/gen_cfg1"
$CXX "$R/tools/gen_golden/gen_cic.cpp" -o "$B/gen_cic_dec"
$CXX -DGEN_INTR "$R/tools/gen_golden/gen_cic.cpp" -o "$B/gen_cic_intr"
$CXX "$R/tools/gen_golden/gen_reg_share.cpp" -o "$B/gen_reg_share"
$CXX "$R/tools/gen_golden/gen_poly_dec.cpp" -o "$B/gen_poly_dec"
$CXX "$R/tools/gen_golden/gen_poly_intr.cpp" -o "$B/gen_poly_intr"
$CXX "$R/tools/gen_golden/gen_mv_avg.cpp" -o "$B/gen_mv_avg"
$CXX "$R/tools/gen_golden/gen_wide.cpp" -o "$B/gen_wide"
$CXX -DGEN_INTR "$R/tools/gen_golden/gen_wide.cpp" -o "$B/gen_wide_intr"
for g in gen_fir gen_cfg1 gen_cic_dec gen_cic_intr gen_reg_share gen_poly_dec gen_poly_intr gen_wide gen_wide_intr; do "$B/$g" "$OUT"; done
# ac_mv_avg runs over this repo's own ac_window_1d_flag restatement (the class is part of the absent ac_types): its vectors pin the
# reference's MAC loop only and live in their own directory
mkdir -p "$R/tests/golden/ref_hdr_window_unpinned"
"$B/gen_mv_avg" "$R/tests/golden/ref_hdr_window_unpinned"
ls -la "$OUT"
