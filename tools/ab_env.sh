#!/bin/bash
# In-model A/B of environment switches on ONE box: bench.py (no legs, no CPU baseline) alternating between the
# settings, 3 rounds.   tools/ab_env.sh OUT.jsonl "" "IRM_NO_APPLY_FUSE=1" [...]   (run on the GPU box from the repo root)
OUT=$1; shift
: > $OUT
for round in 1 2 3; do
  for setting in "$@"; do
    if [ -n "$setting" ]; then export "$setting"; fi
    python3 bench.py --steps 6 --warmup 2 --no-legs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'setting':'$setting','round':$round,'ms_per_frame':d['ms_per_frame'],'abs_dpsnr':d.get('abs_dpsnr'),'kernels':{k:round(v['ms_per_frame'],3) for k,v in d['kernels'].items()}}))" >> $OUT
    if [ -n "$setting" ]; then unset "${setting%%=*}"; fi
  done
done
python3 - $OUT <<'PY'
import json,sys,collections
rows=[json.loads(l) for l in open(sys.argv[1])]
by=collections.defaultdict(list)
for r in rows: by[r['setting']].append(r)
for s,rs in by.items():
    ms=sorted(r['ms_per_frame'] for r in rs)
    ks={k:sorted(r['kernels'][k] for r in rs)[len(rs)//2] for k in rs[0]['kernels']}
    print(repr(s), 'ms_per_frame median %.2f min %.2f'%(ms[len(ms)//2],ms[0]), 'dpsnr', rs[0]['abs_dpsnr'], {k:v for k,v in list(ks.items())[:8]})
PY
