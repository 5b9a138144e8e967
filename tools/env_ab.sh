#!/bin/bash
# one-box A/B of an environment switch of the library:  tools/env_ab.sh NAE_RS_SINGLE [rounds]
R=${2:-2}
for r in $(seq 1 $R); do
  for v in unset set; do
    if [ $v == set ]; then export $1=1; else unset $1; fi
    echo "== $1 $v (round $r)"
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  step %.2f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:5]))
"
  done
done
