#!/bin/bash
# skeleton (no appends) timing of the level-1 kernel with parts of the hot loop compiled out (variants: scripts/build_ablation_variants.sh)
for x in 0 1 2 4 8 15 0; do
CELLMAP_HIP_LIB=$GRAFT_REPO_ROOT/ab/libexp_$x.so CELLMAP_FILTER_PRODUCTS=1 CELLMAP_DBG=5 timeout -k 10 100 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 2 2>&1 | grep -E "cellmap dbg|stage_ms" | tail -2 | python -c "
import sys,json,re
w=None
for l in sys.stdin:
    if l.startswith('{'): r=json.loads(l); print('exp $x p1 ms', round(r['stage_ms']['phase1_mfma'],1), w)
    else:
        m=dict(re.findall(r'(\w+)=(\d+)', l)); w='wave_cyc/tile %.0f barrier %.1f%% dma %.1f%%' % (int(m['wave_cycles'])/2048/67069, 100*int(m['barrier_cycles'])/int(m['wave_cycles']), 100*int(m['dma_wait_cycles'])/int(m['wave_cycles']))"
done
