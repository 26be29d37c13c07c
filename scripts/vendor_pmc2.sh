#!/bin/bash
# PMC of this library's kernels and of rocSPARSE rowsplit in the SAME lab harness as the vendor adaptive pass (L1 -> L2 read requests)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-vendor2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for alg in none rowsplit; do
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_$alg -o v -- lab/vendor_spmv_lab 1000000:f64 --only-vendor $alg > /dev/null 2> $OUT/pmc_$alg.err )
  ( cd $R; python3 scripts/rocpd_pmc.py $(ls $OUT/pmc_$alg/*results.db | head -1) 20 > $OUT/vendor_pmc_l1_$alg.md 2>/dev/null )
  rm -rf $OUT/pmc_$alg
  head -12 $OUT/vendor_pmc_l1_$alg.md | cut -c1-200
done
