for k in 1 2; do
  touch pilotguru_amd/csrc/fast.hip
  make -C pilotguru_amd/csrc -j8 EXTRA=-DPGORB_FAST_SKIP=$k > /dev/null 2>&1
  echo "SKIP=$k: $(python bench.py --no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --steps 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["fast"])')"
done
touch pilotguru_amd/csrc/fast.hip; make -C pilotguru_amd/csrc -j8 > /dev/null 2>&1
echo "full: $(python bench.py --no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg --steps 10 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["stage_ms_per_step"]["fast"])')"
