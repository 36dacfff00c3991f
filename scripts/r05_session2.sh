#!/bin/bash
# round 5, session 2: what differs between the fast and the slow physical placement? PMC passes over alternating copies
set -u
cd "$(dirname "$0")/.."
REPO=$(pwd)
OUT=$REPO/gpurun_out/r05_s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/fragment_study.py --tag cache --copies 2 > $OUT/study_cache.log 2>&1; grep -E "===|copy" $OUT/study_cache.log
timeout 300 python scripts/fragment_study.py --tag sleep2 --copies 4 --sleep 2 > $OUT/study_sleep2.log 2>&1; grep -E "===|copy" $OUT/study_sleep2.log
PASS=0
for counters in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" \
                "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum" \
                "TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
                "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"; do
  PASS=$((PASS + 1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$PASS -- python $REPO/scripts/fragment_study.py --tag pmc$PASS --copies 4 > $OUT/study_pmc$PASS.log 2>&1 )
  grep -E "===|copy" $OUT/study_pmc$PASS.log
  find $OUT/pmc_$PASS -name "*counter_collection.csv" | head -1 | while read f; do python scripts/pmc_per_dispatch.py "$f" search_kernel > $OUT/pmc_$PASS.txt; cat $OUT/pmc_$PASS.txt; done
  rm -rf $OUT/pmc_$PASS
done
rm -f /dev/shm/usearch_amd_fragment_study.img
