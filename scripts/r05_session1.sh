#!/bin/bash
# round 5, session 1: the benchmarked instantiation against the oracle; the fragment study under three allocation policies
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s1; mkdir -p $OUT
export TMPDIR=/tmp
( mount -t debugfs none /sys/kernel/debug 2>&1; ls /sys/kernel/debug/dri/ 2>&1 | head; for d in /sys/kernel/debug/dri/*; do [ -f $d/amdgpu_vram_mm ] && { echo "== $d"; head -60 $d/amdgpu_vram_mm; }; done ) > $OUT/debugfs.log 2>&1
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vm_size 2>&1 > $OUT/amdgpu_params.log
timeout 600 python -m pytest tests/test_gpu_search_parity.py -q -k "benchmarked_instantiation" > $OUT/pytest_inst.log 2>&1; tail -5 $OUT/pytest_inst.log
timeout 300 python scripts/fragment_study.py --tag default --copies 6 > $OUT/study_default.log 2>&1; grep -E "===|copy" $OUT/study_default.log
USEARCH_AMD_ALIGNED_MAP=1 timeout 300 python scripts/fragment_study.py --tag aligned_map --copies 6 > $OUT/study_aligned.log 2>&1; grep -E "===|copy" $OUT/study_aligned.log
HSA_MAX_VA_ALIGN=18 timeout 300 python scripts/fragment_study.py --tag va_align18 --copies 6 > $OUT/study_va18.log 2>&1; grep -E "===|copy" $OUT/study_va18.log
timeout 300 python scripts/fragment_study.py --tag default_held --copies 6 --hold > $OUT/study_default_held.log 2>&1; grep -E "===|copy" $OUT/study_default_held.log
rocprofv3 --list-avail > $OUT/counters.log 2>&1
rm -f /dev/shm/usearch_amd_fragment_study.img
