#!/bin/bash
# The ONE parameterised GPU-box runner (replaces the per-call scripts of earlier rounds).  Usage, through gpurun:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'
# Steps (each writes under gpurun_out/<tag>/ and prints a short tail):
#   tests[:<pytest -k expr>]   pytest -m gpu (optionally a subset)
#   smoke                      __graft_entry__.smoke()
#   bench[:<extra args>]       the default bench line (compact line -> bench.json, full record -> bench_detail.json)
#   profile:<round>            scripts/profile_bench.sh <round>  (rocprofv3 kernel trace + the PMC passes, separate runs)
#   psd                        scripts/bench_psd_sizes.py + the configs[2] SDP (scripts/bench_sdp.py)
#   term:<n>:<threads>         scripts/term_parity.py in the BACKGROUND (its CPU leg runs beside the later steps); waited for at the end
#   sh:<command>               anything else, verbatim
set -u
export SCS_AMD_ALLOW_ENV_HOOKS=1 # A/B script: measurement variants of scs_amd/csrc/options.h are set through the environment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export PYTHONUNBUFFERED=1
TERM_PID=""
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $kind in
    tests)
      if [ -n "$arg" ]; then ( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 900 -k "$arg" ) > "$OUT/pytest.log" 2>&1
      else ( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 900 ) > "$OUT/pytest.log" 2>&1; fi
      tail -6 "$OUT/pytest.log" ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    bench)
      ( time SCS_BENCH_DETAIL="$OUT/bench_detail.json" timeout 900 python bench.py $arg ) > "$OUT/bench.json" 2> "$OUT/bench.err"
      wc -c "$OUT/bench.json"; tail -c 3000 "$OUT/bench.json" ;;
    profile) bash scripts/profile_bench.sh "$arg" 2>&1 | tail -30 ;;
    psd)
      python scripts/bench_psd_sizes.py > "$OUT/psd_sizes.jsonl" 2> "$OUT/psd_sizes.err"; cat "$OUT/psd_sizes.jsonl"
      python scripts/bench_sdp.py > "$OUT/sdp.json" 2> "$OUT/sdp.err"; cat "$OUT/sdp.json" ;;
    term)
      n=${arg%%:*}; thr=${arg#*:}
      ( python scripts/term_parity.py --n "$n" --threads "$thr" --out "$OUT/term_parity_n$n.json" > "$OUT/term_parity.log" 2>&1 ) &
      TERM_PID=$! ;;
    sh) bash -c "$arg" 2>&1 | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
if [ -n "$TERM_PID" ]; then echo "=== waiting for the to-termination reference leg"; wait $TERM_PID; cat "$OUT/term_parity.log"; fi
