cd $GRAFT_REPO_ROOT
for g in 30000 7500; do for mode in "" "CELLMAP_A3_COMPLETION=1" "CELLMAP_NO_ROW_ORDER=1"; do
env $mode python bench.py --config 5 --nq 1000000 --genes $g --steps 2 --warmup 1 --check 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('genes $g $mode: a3 %.1f ms  frac %.3f  stages %s' % (r['roofline']['kernel_ms'], r['roofline']['frac'], {k: round(v,1) for k,v in r['stage_ms_last_step'].items()}))"
done; done
cd $GRAFT_REPO_ROOT
for v in noprod prodsb; do for r in 1 2; do
CELLMAP_HIP_LIB=$PWD/build_ab/lib_$v.so python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0 --reps 2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d128 $v: phase1 %.1f ms knn %.1f' % (r['stage_ms']['phase1_mfma'], r['knn_ms']))"
python scripts/bench_knn.py --nq 1500000 --nref 1500000 --d 128 --k 30 --uniform 0 --reps 2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d128 prod: phase1 %.1f ms knn %.1f' % (r['stage_ms']['phase1_mfma'], r['knn_ms']))"
done; done
