#!/bin/bash
# scripts/gpu_session.sh <tag> <step>… — one GPU-box session made of named steps, every step under its own timeout, logs under
# gpurun_out/<tag>/. Steps: tests | sweep10m | bench | sharded1 | c4small | c5small | profile (see the case arms).
set -u
TAG=$1; shift 1
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
for step in "$@"; do
  echo "=== $step $(date +%T)"
  case $step in
    tests)    timeout 900 python -m pytest tests -m gpu -q --maxfail=8 > "$OUT/pytest.log" 2>&1; tail -15 "$OUT/pytest.log" ;;
    sweep10m) timeout 600 python scripts/sweep.py --n 10000000 --ef 592 256 --modes 2 --waves 0 --variants 1 2 3 4 5 \
                --frontiers 1 2 --steps 3 > "$OUT/sweep10m.log" 2>&1; cat "$OUT/sweep10m.log" ;;
    sweepv)   timeout 600 python scripts/sweep.py --n 10000000 --ef 608 --queries 10000 100000 --modes 2 --waves 0 --variants 3 1 2 4 \
                --frontiers 1 2 --steps 3 > "$OUT/sweepv.log" 2>&1; cat "$OUT/sweepv.log" ;;
    sweepwaves) timeout 600 python scripts/sweep.py --n 10000000 --ef 592 --modes 2 --waves 8 12 16 --variants 1 2 3 4 5 \
                --frontiers 2 --steps 3 > "$OUT/sweepwaves.log" 2>&1; cat "$OUT/sweepwaves.log" ;;
    bench)    timeout 900 python bench.py --steps 20 --warmup 5 --wave-clock > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -25 "$OUT/bench.log"; cat "$OUT/bench.json" ;;
    sharded1) timeout 600 python bench.py --sharded --vectors 2000000 --dim 128 --dtype b1 --queries 100000 --steps 5 --warmup 2 \
                > "$OUT/sharded1.json" 2> "$OUT/sharded1.log"; tail -12 "$OUT/sharded1.log"; cat "$OUT/sharded1.json" ;;
    c4small)  timeout 600 python bench.py --vectors 20000000 --dim 96 --dtype i8 --queries 100000 --no-stress-rows --cpu-seconds 4 \
                > "$OUT/c4small.json" 2> "$OUT/c4small.log"; tail -8 "$OUT/c4small.log"; cat "$OUT/c4small.json" ;;
    c5small)  timeout 600 python bench.py --vectors 20000000 --dim 128 --dtype b1 --queries 100000 --no-stress-rows --cpu-seconds 4 \
                > "$OUT/c5small.json" 2> "$OUT/c5small.log"; tail -8 "$OUT/c5small.log"; cat "$OUT/c5small.json" ;;
    c5ab)     for inline in 0 1; do USEARCH_AMD_INLINE_ROWS=$inline timeout 300 python bench.py --vectors 20000000 --dim 128 --dtype b1 \
                --queries 100000 --expansion 64 --recall-queries 1000 --no-stress-rows --no-cpu-baseline --steps 5 \
                > "$OUT/c5_inline$inline.json" 2> "$OUT/c5_inline$inline.log"; cat "$OUT/c5_inline$inline.json"; done ;;
    c4ab)     for dense in 1 0; do USEARCH_AMD_DENSE_ROWS=$dense timeout 300 python bench.py --vectors 20000000 --dim 96 --dtype i8 \
                --queries 100000 --expansion 96 --recall-queries 1000 --no-stress-rows --no-cpu-baseline --steps 5 \
                > "$OUT/c4_dense$dense.json" 2> "$OUT/c4_dense$dense.log"; cat "$OUT/c4_dense$dense.json"; done ;;
    sharded2) timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_exact.py -q > "$OUT/sharded2.log" 2>&1; tail -15 "$OUT/sharded2.log" ;;
    prof10m)  timeout 1500 bash scripts/profile_round.sh $TAG/headline ;;
    profc4)   PROFILE_TRAFFIC_ONLY=1 timeout 1500 bash scripts/profile_round.sh $TAG/c4 --vectors 100000000 --dim 96 --dtype i8 --queries 100000 ;;
    profc5)   PROFILE_TRAFFIC_ONLY=1 timeout 1500 bash scripts/profile_round.sh $TAG/c5 --vectors 125000000 --dim 128 --dtype b1 --queries 100000 ;;
    w5ab)     for lib in "" "$REPO/usearch_amd/lib_waves5/libusearch_amd.so"; do for shape in "--dim 128 --dtype b1 --expansion 64" "--dim 96 --dtype i8 --expansion 64"; do
                echo "--- library ${lib:-product} $shape"
                USEARCH_AMD_LIBRARY=$lib timeout 300 python bench.py --vectors 20000000 $shape --queries 100000 \
                  --recall-queries 1000 --no-stress-rows --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'QPS', d['config']['persistent_waves'], 'waves', d['roofline']['kernel_ms'], 'ms')"
              done; done ;;
    variance) for run in 1 2; do echo "--- process $run"; timeout 400 python scripts/variance_probe.py 2>&1 | grep -v "amdgpu.ids"; done | tee "$OUT/variance.log" ;;
    order)    # files that load the engine / the drop-in BEFORE anything imports torch: one HIP runtime either way (index.py)
              timeout 600 python -m pytest tests/test_gpu_search_parity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py tests/test_gpu_build.py \
                -m gpu -q --maxfail=5 > "$OUT/order_pytest.log" 2>&1; tail -4 "$OUT/order_pytest.log" ;;
    bucketsab) L=$REPO/usearch_amd/lib_buckets/libusearch_amd.so
              USEARCH_AMD_LIBRARY=$L timeout 600 python -m pytest tests/test_gpu_search_parity.py tests/test_gpu_fullsize.py tests/test_gpu_build.py \
                tests/test_gpu_golden.py -m gpu -q --maxfail=5 > "$OUT/buckets_pytest.log" 2>&1; tail -6 "$OUT/buckets_pytest.log"
              for shape in "--vectors 20000000 --dim 128 --dtype b1 --expansion 64 --queries 100000" "--vectors 20000000 --dim 96 --dtype i8 --expansion 80 --queries 100000" \
                           "--vectors 10000000 --expansion 608"; do for lib in "" "$L"; do
                echo "--- library ${lib:-product} $shape"
                USEARCH_AMD_LIBRARY=$lib timeout 400 python bench.py $shape --recall-queries 1000 --no-stress-rows --no-cpu-baseline --no-host-api --steps 5 2>/dev/null \
                  | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'QPS', d['roofline']['kernel_ms'], 'ms', 'recall', d['config'].get('recall_at_k'))"
              done; done 2>&1 | tee "$OUT/buckets_ab.log" ;;
    exactcheck) timeout 900 python scripts/exact_check.py --vectors 1000000 20000000 40000000 100000000 --dim 96 --dtype i8 > "$OUT/exactcheck.log" 2>&1; cat "$OUT/exactcheck.log" ;;
    exact10m) timeout 900 python scripts/exact_check.py --vectors 10000000 --dim 768 --dtype f16 --queries 1000 > "$OUT/exact10m.log" 2>&1; cat "$OUT/exact10m.log" ;;
    c2)       timeout 600 python bench.py --vectors 1000000 --dim 768 --dtype f32 --steps 10 --warmup 2 > "$OUT/c2.json" 2> "$OUT/c2.log"; tail -6 "$OUT/c2.log"; cat "$OUT/c2.json" ;;
    c5sharded) timeout 900 python bench.py --sharded --vectors 125000000 --dim 128 --dtype b1 --queries 100000 --steps 10 --warmup 2 \
                > "$OUT/c5sharded.json" 2> "$OUT/c5sharded.log"; tail -6 "$OUT/c5sharded.log"; cat "$OUT/c5sharded.json" ;;
    tailhist) USEARCH_AMD_WAVE_CLOCK=2 timeout 600 python scripts/sweep.py --n 10000000 --ef 608 --queries 10000 --modes 2 --waves 0 --variants 4 1 \
                --frontiers 2 --steps 2 > "$OUT/tailhist.log" 2>&1; cat "$OUT/tailhist.log" ;;
    visits)   timeout 600 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 2 1 --waves 0 --steps 3 \
                --env "" USEARCH_AMD_HASH_CAP=4096 USEARCH_AMD_HASH_CAP=2048 > "$OUT/visits_b1.log" 2>&1; cat "$OUT/visits_b1.log"
              timeout 600 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 --queries 100000 --modes 2 1 --waves 0 --steps 3 \
                --env "" USEARCH_AMD_HASH_CAP=4096 > "$OUT/visits_i8.log" 2>&1; cat "$OUT/visits_i8.log" ;;
    phases)   export USEARCH_AMD_LIBRARY=$REPO/usearch_amd/lib_phases/libusearch_amd.so USEARCH_AMD_PHASES=1
              timeout 600 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 2 1 --waves 0 --steps 1 > "$OUT/phases_b1.log" 2>&1
              timeout 600 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 --queries 100000 --modes 2 --waves 0 --steps 1 > "$OUT/phases_i8.log" 2>&1
              timeout 600 python scripts/sweep.py --n 10000000 --ef 608 --queries 10000 --modes 2 --waves 0 --variants 4 --frontiers 2 1 --steps 1 > "$OUT/phases_f16.log" 2>&1
              unset USEARCH_AMD_LIBRARY USEARCH_AMD_PHASES
              grep -h "phases ef=64 grid=6144\|phases ef=64 grid=1024\|phases ef=80 grid=4096\|phases ef=608 grid=2048\|^ef=" "$OUT"/phases_*.log | tail -30 ;;
    hashcap)  timeout 600 python scripts/sweep.py --n 20000000 --dim 128 --dtype b1 --ef 64 --queries 100000 --modes 2 --waves 0 --steps 3 \
                --env "" USEARCH_AMD_HASH_CAP=16384 USEARCH_AMD_HASH_CAP=32768 USEARCH_AMD_HASH_CAP=65536 > "$OUT/hashcap_b1.log" 2>&1; grep -v "wave exits\|amdgpu.ids" "$OUT/hashcap_b1.log"
              timeout 600 python scripts/sweep.py --n 20000000 --dim 96 --dtype i8 --ef 80 --queries 100000 --modes 2 --waves 0 --steps 3 \
                --env "" USEARCH_AMD_HASH_CAP=16384 USEARCH_AMD_HASH_CAP=32768 > "$OUT/hashcap_i8.log" 2>&1; grep -v "wave exits\|amdgpu.ids" "$OUT/hashcap_i8.log" ;;
    c4)       timeout 900 python bench.py --vectors 100000000 --dim 96 --dtype i8 --queries 100000 --steps 10 --warmup 2 > "$OUT/c4.json" 2> "$OUT/c4.log"; grep -v "ef=" "$OUT/c4.log" | tail -8; cat "$OUT/c4.json" ;;
    rehearse) BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --vectors 500000 --dim 768 --dtype f16 --queries 2000 --steps 3 --warmup 1 \
                > "$OUT/rehearse_replicas.json" 2> "$OUT/rehearse_replicas.log"; tail -3 "$OUT/rehearse_replicas.log"; cat "$OUT/rehearse_replicas.json"
              BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --sharded --vectors 2000000 --dim 128 --dtype b1 --queries 20000 --steps 3 --warmup 1 \
                > "$OUT/rehearse_sharded.json" 2> "$OUT/rehearse_sharded.log"; tail -3 "$OUT/rehearse_sharded.log"; cat "$OUT/rehearse_sharded.json" ;;
    *) echo "unknown step $step" ;;
  esac
done
echo "=== done $(date +%T)"
