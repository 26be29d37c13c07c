#!/bin/bash
# round 3, GPU call D: pipelined wave kernel depths on banded matrices; kernel trace + PMC of the blocked Jacobi path
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_linsys_gpu.py -q --timeout 600 ) > $OUT/pytest_linsys.log 2>&1; tail -3 $OUT/pytest_linsys.log
timeout 900 python scripts/bench_locality.py --bands 1024,4096,65536 --modes auto,0,1,2 > $OUT/locality.jsonl 2> $OUT/locality.err; cat $OUT/locality.jsonl
cd /tmp
export SCS_AMD_GRAPH=0
rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $R/scripts/bench_psd_sizes.py --cases 1024x1,256x8 --iters 20 > $OUT/psd_traced.out 2> $OUT/psd_traced.err
CTRS="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc -o p -- python $R/scripts/bench_psd_sizes.py --cases 256x8 --iters 10 > $OUT/psd_pmc.out 2> $OUT/psd_pmc.err
cd $R
python3 scripts/rocpd_stats.py $(ls $OUT/tr/*results.db | head -1) 0 > $OUT/psd_kernel_stats.md 2>/dev/null
python3 scripts/rocpd_pmc.py $(ls $OUT/pmc/*results.db | head -1) 0 2>/dev/null | grep -E "k_bj_|k_bp_|kernel|---" > $OUT/psd_pmc.md
rm -rf $OUT/tr $OUT/pmc
head -20 $OUT/psd_kernel_stats.md | cut -c1-220; cat $OUT/psd_traced.out; grep -E "MFMA|INSTS_VALU |BUSY|INSTS_LDS|kernel" $OUT/psd_pmc.md | cut -c1-200 | head -50
