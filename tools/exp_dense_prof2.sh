# usage: exp_dense_prof2.sh "<env>" : conv / BatchNorm kernel totals of one replayed --dense-head step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in $1; do export "$kv"; done
rm -rf /tmp/pdh; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pdh -o r -- python3 bench.py --dense-head --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d > /dev/null 2>&1
DB=$(find /tmp/pdh -name "*.db" | head -1)
python tools/timeline.py $DB /tmp/dh_seq.txt > /dev/null
python - <<'PY'
import re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open("/tmp/dh_seq.txt"):
    q, t0, dur, name = l.split(None, 3)
    name = re.sub(r"\(.*", "", name.strip())[:40]
    agg[name][0] += 1; agg[name][1] += float(dur)
for k in sorted(agg):
    if any(s in k for s in ("conv2d_3x3", "bn_")):
        print(f"{agg[k][0]:4d} {agg[k][1]:8.1f} us  avg {agg[k][1]/agg[k][0]:6.1f}  {k}")
PY
