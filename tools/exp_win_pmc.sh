# SQ counters (own --pmc pass, kernel-trace only) of the three gather-GEMM designs on the level-3 SubM 64 -> 64 layer
# (tools/exp_win.py): MFMA-busy cycles, wave cycles, wait cycles, LDS bank conflicts, active instructions per launch.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_pmc_sq_win.txt; : > $OUT
for m in 0 1 4; do
  rm -rf /tmp/psw$m
  PCD_GGWIN=$m timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d /tmp/psw$m -o r -- python3 tools/exp_win.py > /dev/null 2>&1
  S=$(find /tmp/psw$m -name "*.db" | head -1)
  echo "## PCD_GGWIN=$m" >> $OUT
  python tools/pmc_summary.py $S | grep -E "^kernel|gather_gemm_kernel|ggwin_kernel|ggreg_kernel" | cut -c1-260 >> $OUT
done
cat $OUT
