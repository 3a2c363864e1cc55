# A/B of alternative builds of the window kernel: com_amd/lib/alt_*.so against the built library
L=com_amd/lib/libpcdops_hip.so
cp $L /tmp/lib_main.so
for alt in main $(ls com_amd/lib/alt_*.so 2>/dev/null); do
  [ $alt = main ] && cp /tmp/lib_main.so $L || cp $alt $L
  for d in ${2:-0 12}; do echo "== $alt win_dbg=$d"; PCD_OPT_WIN_DBG=$d timeout 200 python tools/exp_subm_win.py ${1:-3} 2>&1 | grep "fwd:" | sed 's/.*generic/generic/'; done
done
cp /tmp/lib_main.so $L
