# 32 -> 32 SubM layers with the whole packed weight (55 KB) resident in LDS (no stage barriers, 2 workgroups / CU) against the
# staged default: same-box A/B on the whole step
cd $GRAFT_REPO_ROOT
for kb in 32 64 32 64; do
  PCD_GG_RESIDENT_KB=$kb python bench.py --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-full-model --no-fp8 --no-stage2 --no-regime --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident_kb', $kb, d['value'], d['ms_per_step'])"
done
