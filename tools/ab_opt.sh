#!/bin/bash
# usage: tools/ab_opt.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- alternating `bench.py --light` runs, two rounds
for i in $(seq 1 ${ROUNDS:-2}); do
for e in "$@"; do
env $e python bench.py --light --steps ${STEPS:-60} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$e', d['value'], d['ms_per_step'])
"
done; done
