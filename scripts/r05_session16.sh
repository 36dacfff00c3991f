#!/bin/bash
# round 5, session 16: the exact kernel's counters (two PMC passes) and the whole GPU suite on the final libraries
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PMC_GROUPS="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD|SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" timeout 900 bash scripts/pmc_kernel.sh r05_final/exact_pmc exact_wide_kernel -- python $PWD/bench.py --exact --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r05_s16_pmc.log 2>&1
cat gpurun_out/r05_final/exact_pmc/pmc_summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05_final/pytest_gpu.log 2>&1; tail -2 gpurun_out/r05_final/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
