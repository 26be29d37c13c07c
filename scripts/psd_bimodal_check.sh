cd "$GRAFT_REPO_ROOT"
for cs in "100x32,100x32,100x32" "92x64,100x32,100x32" "128x32,100x32,92x64,100x32"; do
 echo "cases $cs"; python scripts/bench_psd_sizes.py --cases $cs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l: d=json.loads(l); print(d['k'], d['blocks'], round(d['gpu_ms_per_projection'],2), round(d['wall_s'],2))"
done
