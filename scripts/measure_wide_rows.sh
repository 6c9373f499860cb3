cd $GRAFT_REPO_ROOT
for d in 256 512 1024; do
python scripts/bench_knn.py --nq 100000 --nref 200000 --d $d --k 30 --uniform 0 --reps 2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=r['stage_ms']['phase1_mfma']; print('d=$d: phase1 %.1f ms (%.1f TFLOP/s f32)  knn %.1f ms  fallback %d' % (p, 2*1e5*2e5*$d/p/1e9, r['knn_ms'], r['stats']['fallback_queries']))"
done
