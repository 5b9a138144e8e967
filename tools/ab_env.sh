#!/bin/bash
# usage: ab_env.sh "ENV1" "ENV2" ... ; each arg is an env assignment string (may be empty)
for r in 1 2 3; do
for e in "$@"; do
  env $e python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-pcie --no-host-path --total-streams ${N:-1024} | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$e]  step %.3f ms | ' % d['ms_per_step'] + ' '.join('%s %.3f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:3]))
"
done
done
