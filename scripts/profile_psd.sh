#!/bin/bash
# rocprofv3 PMC passes behind profiles/r2_sdp_psd_kernel_pmc.md (run on the GPU box): instruction mix of the PSD kernels on
# BASELINE configs[2] (LDS Jacobi kernel) and on 8 blocks of 256x256 (chip-wide path, psd_big.h).  SCS_AMD_GRAPH=0: rocprofv3
# crashes on replayed HIP-graph kernel nodes.  Counters only with --kernel-trace (no other trace domain).
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1 # A/B script: measurement variants of scs_amd/csrc/options.h are set through the environment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_psd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SCS_AMD_GRAPH=0
CTRS="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/sdp -o p -- python $R/scripts/bench_sdp.py > $OUT/sdp.out 2> $OUT/sdp.err
rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/big -o p -- python $R/scripts/bench_psd_sizes.py --cases 256x8 --iters 10 > $OUT/big.out 2> $OUT/big.err
cd $R
{
  echo "# instruction mix of the PSD kernels (rocprofv3 --kernel-trace --pmc $CTRS, scripts/profile_psd.sh, one MI355X)"
  echo "## BASELINE configs[2]: 200 PSD blocks 50x50 + box(1001) -- k_psd_jacobi (one workgroup per block, A and V in LDS, warm start)"
  python3 scripts/rocpd_pmc.py $(ls $OUT/sdp/*results.db | head -1) 50 2>/dev/null | grep -E "k_psd|kernel|---"
  tail -3 $OUT/sdp.out
  echo "## 8 PSD blocks 256x256 -- psd_big.h (chip-wide Jacobi steps; MFMA in k_bp_gemm_tn (warm start) and k_bp_gram (reconstruction))"
  python3 scripts/rocpd_pmc.py $(ls $OUT/big/*results.db | head -1) 0 2>/dev/null | grep -E "k_bp_|k_bj_|kernel|---" | grep -E "MFMA|SQ_INSTS_VALU |kernel|---|SQ_INSTS_LDS"
  python3 scripts/rocpd_stats.py $(ls $OUT/big/*results.db | head -1) 0 2>/dev/null | grep -E "k_bp_|k_bj_|kernel \||---"
} > $OUT/psd_pmc.md
rm -rf $OUT/sdp $OUT/big
head -40 $OUT/psd_pmc.md | cut -c1-200
