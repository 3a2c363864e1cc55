# which of the dense-head forms are bit-reproducible?  (one-graph twice, three-graph, with / without --com)
cd $GRAFT_REPO_ROOT
C="python bench.py --steps 4 --warmup 2 --batch 2 --no-cpu-baseline --no-roofline --no-h2d --no-ragged --no-full-model --no-stage2 --no-fp8 --distinct-batches 3 --same-shard --gpus 1"
run() { tag=$1; shift; env "$@" $C --dump-state /tmp/st_$tag.json $EXTRA > /tmp/out_$tag.json 2>/tmp/err_$tag.txt; python tools/exp_forms_print.py $tag; }
EXTRA="--dense-head"; run dh_a A=1; run dh_b A=1; run dh_3g PCD_FORCE_3GRAPH=1
EXTRA="--dense-head --com"; run com_a A=1; run com_b A=1; run com_3g PCD_FORCE_3GRAPH=1
EXTRA=""; run sparse_a A=1; run sparse_b A=1
