# kernel time by name inside ONE replayed step of bench.py --dense-head (last step of the trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pdh; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pdh -o r -- python3 bench.py --dense-head --com --steps 5 --warmup 2 --light > gpurun_out/dh_bench.log 2>&1
DB=$(find /tmp/pdh -name "*.db" | head -1)
python tools/timeline.py $DB gpurun_out/dh_seq.txt | head -12
python - <<'PY'
import re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open("gpurun_out/dh_seq.txt"):
    q, t0, dur, name = l.split(None, 3)
    name = re.sub(r"<.*", "", name.strip())[:60]
    agg[name][0] += 1; agg[name][1] += float(dur)
tot = sum(v[1] for v in agg.values())
print(f"kernel time in the step: {tot/1e3:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[0]:5d} {v[1]:9.1f} us  {k}")
PY
tail -1 gpurun_out/dh_bench.log | cut -c1-200
python - <<'PY'
import re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in open("gpurun_out/dh_seq.txt"):
    q, t0, dur, name = l.split(None, 3)
    if not any(k in name for k in ("elementwise", "reduce_kernel<", "Cat", "rocclr", "scatter_gather", "index", "SubTensor", "multi_tensor", "FillFunctor")):
        continue
    agg[name.strip()[:140]][0] += 1; agg[name.strip()[:140]][1] += float(dur)
print("torch / runtime glue kernels in the step:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:5d} {v[1]:9.1f} us  {k}")
PY
