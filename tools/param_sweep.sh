#!/bin/bash
# the C5 graph at other pitch-node settings than the headline's (+3 semitones): step time and the top kernels
for args in "--semitones 3" "--semitones -3" "--semitones 7" "--semitones -7" "--rate 1.5 --semitones -7.01955" "--rate 0.8 --semitones 0" "--semitones 0"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-host-path $args | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d.get('pitch_node_soundtouch_algorithm',{})
print('%-34s step %6.2f ms  %.3g sf/s | ' % ('$args', d['ms_per_step'], d['value']) + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:5]) + ' | wsola node %.2f ms' % w.get('pitch_node_ms', float('nan')))
"
done
