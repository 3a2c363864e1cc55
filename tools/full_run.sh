# full GPU suite + the driver's default bench command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1
tail -6 gpurun_out/full_tests.log
timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
tail -c 600 gpurun_out/full_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
r=d.get('roofline',{})
print('roofline', {k:r.get(k) for k in ('kernel','frac','frac_rocprof','avg_launch_us','traffic')})
print('rulebook', r.get('rulebook'))
for k in ('h2d_inclusive','ragged','full_model','config5_300k','stage2','n_gt_1_form','cpu_baseline'):
    v=d.get(k); 
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('what','sample','config','form')}
    print(k, v)
PY
