for i in 1 2; do
for f in 1 0; do
PCD_BN_FOLD=$f python bench.py --light --steps 60 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fold=$f', d['value'], d['ms_per_step'])
"
done; done
