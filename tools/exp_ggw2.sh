cd $GRAFT_REPO_ROOT
for ch in 128 64; do
echo -n "old        "; PCD_GGW=0 python tools/exp_l4.py $ch
for mi in 2 3; do for d in 0 64 4; do echo -n "GGW=$mi DBG=$d  "; PCD_GGW=$mi PCD_GGW_DBG=$d python tools/exp_l4.py $ch; done; done
done
