# ablations of the window kernel (win_dbg bits: 1 no DMA, 4 no MFMA loop, 8 no reduction / epilogue)
for d in ${2:-0 1 4 8 5 12 13}; do echo "win_dbg=$d"; PCD_OPT_WIN_DBG=$d timeout 200 python tools/exp_subm_win.py ${1:-3} 2>&1 | grep "fwd:" | sed 's/.*generic/generic/'; done
