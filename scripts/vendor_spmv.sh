#!/bin/bash
# lab/vendor_spmv_lab (rocSPARSE csrmv beside this library's SpMV kernels, lab only) + one rocprofv3 PMC pass of the vendor kernels on the headline
# matrix.  Output: gpurun_out/<tag>/vendor_spmv.md, vendor_pmc_l1.md, vendor_pmc_l2.md   
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-vendor}
mkdir -p $OUT
cd $R
CASES=${2:-1000000:f64,200000:f64,4000000:f32}
: > $OUT/vendor_spmv.md; : > $OUT/vendor_spmv.err
# one process per algorithm (a fault inside one vendor kernel must not cost the other rows), this library's kernels last
for alg in adaptive rowsplit lrb nnzsplit default none; do
  echo "### process: --only-vendor $alg" >> $OUT/vendor_spmv.md
  timeout 600 lab/vendor_spmv_lab $CASES --only-vendor $alg >> $OUT/vendor_spmv.md 2>> $OUT/vendor_spmv.err || echo "(process for $alg ended with status $?)" >> $OUT/vendor_spmv.md
done
grep -E "^\||^##|status" $OUT/vendor_spmv.md | head -80
cd /tmp && export TMPDIR=/tmp
# counters in their own runs, kernel trace only beside them (two passes: counter slots)
( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc1 -o v -- lab/vendor_spmv_lab 1000000:f64 --only-vendor adaptive > /dev/null 2> $OUT/pmc1.err )
( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -o v -- lab/vendor_spmv_lab 1000000:f64 --only-vendor adaptive > /dev/null 2> $OUT/pmc2.err )
cd $R
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc1/*results.db | head -1) 20 > $OUT/vendor_pmc_l1.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc2/*results.db | head -1) 20 > $OUT/vendor_pmc_l2.md 2>/dev/null
rm -rf $OUT/pmc1 $OUT/pmc2
head -30 $OUT/vendor_pmc_l1.md; head -30 $OUT/vendor_pmc_l2.md
