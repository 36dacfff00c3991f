#!/bin/bash
# round 5, session 13: the exact bench line on the final exact kernel, its kernel trace, then the whole GPU suite and smoke()
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_final/exact; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python bench.py --exact --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.log; tail -c 900 $OUT/bench.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats -- python $OLDPWD/bench.py --exact --no-cpu-baseline --steps 5 --warmup 1 > $OLDPWD/$OUT/stats_bench.json 2> $OLDPWD/$OUT/stats.log)
find $OUT/stats -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; done; rm -rf $OUT/stats; head -4 $OUT/kernel_stats.csv | cut -c1-220
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r05_final/pytest_gpu.log 2>&1; tail -4 gpurun_out/r05_final/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
