cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -k "sparse_conv or basic_block" 2>&1 | tail -1
for ch in 128 64; do
echo -n "old        "; PCD_GGW=0 python tools/exp_l4.py $ch
for mi in 2 3; do for d in 0 4; do echo -n "GGW=$mi DBG=$d  "; PCD_GGW=$mi PCD_GGW_DBG=$d python tools/exp_l4.py $ch; done; done
done
