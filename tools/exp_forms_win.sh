# which window-kernel options keep the sparse step bit-reproducible?  (one-graph twice + three-graph per option set)
cd $GRAFT_REPO_ROOT
C="python bench.py --steps 4 --warmup 2 --batch 2 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-full-model --no-stage2 --no-fp8 --distinct-batches 3 --same-shard --gpus 1"
run() { tag=$1; shift; env "$@" $C --dump-state /tmp/st_$tag.json > /tmp/out_$tag.json 2>/tmp/err_$tag.txt; python tools/exp_forms_print.py $tag; }
for opt in "7 6" "7 0" "3 0" "1 0" "0 0" "4 0" "4 4" "2 2"; do
  set -- $opt
  echo "== subm_window=$1 subm_window_wgrad=$2"
  run w$1g$2_a PCD_OPT_SUBM_WINDOW=$1 PCD_OPT_SUBM_WINDOW_WGRAD=$2
  run w$1g$2_b PCD_OPT_SUBM_WINDOW=$1 PCD_OPT_SUBM_WINDOW_WGRAD=$2
  run w$1g$2_3 PCD_OPT_SUBM_WINDOW=$1 PCD_OPT_SUBM_WINDOW_WGRAD=$2 PCD_FORCE_3GRAPH=1
done
