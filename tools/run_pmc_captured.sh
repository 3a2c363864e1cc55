# HBM-side traffic per kernel of the CAPTURED step (the kernels and table forms the product path runs; the eager passes of
# run_round_profiles.sh see the two-phase builds and 27-wide tables instead): two separate --pmc passes, graph replays.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
rm -rf /tmp/pf /tmp/pw
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r -- python3 bench.py --steps 4 --warmup 2 --light > gpurun_out/${TAG}_pmc_cap.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o r -- python3 bench.py --steps 4 --warmup 2 --light >> gpurun_out/${TAG}_pmc_cap.log 2>&1
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
python tools/pmc_traffic.py $F $W > gpurun_out/${TAG}_pmc_traffic_captured.json
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_pmc_traffic_captured.json'))
for n,v in list(d['kernels'].items())[:45]:
    print(n[:80].ljust(80), v['launches_sampled'], round(v['hbm_bytes_per_launch_corrected']/1e6,1))
PY
