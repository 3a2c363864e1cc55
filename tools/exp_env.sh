# usage: exp_env.sh "<bench flags>" "<env A>" "<env B>" ...   ("-" = no env) -> frames/s, ms/step (+ stamps) per env
FLAGS=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then for kv in $cfg; do export "$kv"; done; fi
    python bench.py --no-cpu-baseline --no-roofline $FLAGS 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
    grep "stamps" /tmp/err.txt )
done
