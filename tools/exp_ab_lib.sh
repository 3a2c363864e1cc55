# usage: exp_ab_lib.sh <alt .so> "<bench flags>"  -- A/B of the built library against another build, alternating
ALT=$1; FLAGS=${2:---steps 200 --warmup 20}
L=com_amd/lib/libpcdops_hip.so
cp $L /tmp/lib_a.so; cp $ALT /tmp/lib_b.so
for rep in 1 2; do
  for v in a b; do
    cp /tmp/lib_$v.so $L
    echo "== $v ($([ $v = a ] && echo built || echo $ALT))"
    python bench.py --no-cpu-baseline --no-roofline $FLAGS 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
    grep "stamps" /tmp/err.txt
  done
done
cp /tmp/lib_a.so $L
