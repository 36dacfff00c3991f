#!/bin/bash
# round 5, session 3: what exactly makes a placement fast — time after the free, idleness, or the driver's late release of freed blocks?
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_s3; mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python scripts/fragment_study.py --tag $tag "$@" > $OUT/study_$tag.log 2>&1; grep -E "===|copy|free memory" $OUT/study_$tag.log | grep -v "HSA_MAX"; }
run cache --copies 2
run settle --copies 4 --settle
run sleep03 --copies 4 --sleep 0.3
run sleep1 --copies 4 --sleep 1
run idle_before_close --copies 4 --sleep-before-close 2
run held_sleep2 --copies 4 --hold --sleep 2
run sleep2_again --copies 4 --sleep 2
rm -f /dev/shm/usearch_amd_fragment_study.img
