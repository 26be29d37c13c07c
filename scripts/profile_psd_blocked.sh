#!/bin/bash
# rocprofv3 kernel trace + instruction-mix PMC pass of the blocked Jacobi path (psd_big.h) on 1024x1 and 256x8
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_psd_blocked
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SCS_AMD_GRAPH=0
export SCS_AMD_ALLOW_ENV_HOOKS=1
rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $R/scripts/bench_psd_sizes.py --cases 1024x1,512x2,256x8,92x64 --iters 20 > $OUT/psd_traced.out 2> $OUT/psd_traced.err
CTRS="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc -o p -- python $R/scripts/bench_psd_sizes.py --cases 256x8 --iters 10 > $OUT/psd_pmc.out 2> $OUT/psd_pmc.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/tr/*results.db | head -1) 0 > $OUT/psd_kernel_stats.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc/*results.db | head -1) 0 2>/dev/null | grep -E "k_bj_|k_bp_|kernel|---" > $OUT/psd_pmc.md
rm -rf $OUT/tr $OUT/pmc
head -10 $OUT/psd_kernel_stats.md | cut -c1-200; cat $OUT/psd_traced.out | cut -c1-150; grep -E "k_bj" $OUT/psd_pmc.md | cut -c1-160
