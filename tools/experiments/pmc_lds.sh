export TMPDIR=/tmp
rm -rf gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS -d gpurun_out/pmc_lds -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --sustain-seconds 0 --no-upload-leg > gpurun_out/pmc_lds.log 2>&1
python tools/rocpd_summary.py pmc $(find gpurun_out/pmc_lds -name '*.db' | head -1) | grep -E "kernel|k_fast"
