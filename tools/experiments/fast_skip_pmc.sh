export TMPDIR=/tmp
for k in 1 2 0; do
  touch pilotguru_amd/csrc/fast.hip
  if [ $k = 0 ]; then make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1; else make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_FAST_SKIP=$k > /dev/null 2>&1; fi
  rm -rf gpurun_out/pmcs_$k
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/pmcs_$k -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg > /dev/null 2>&1
  echo "SKIP=$k"; python tools/rocpd_summary.py pmc $(find gpurun_out/pmcs_$k -name '*.db' | head -1) | grep -E "kernel|k_fast"
done
