#!/bin/bash
# Collects, on the GPU box, the artifacts bench.py's roofline block refers to: the rocprofv3 kernel statistics of the bench
# command and the two PMC passes (FETCH_SIZE, WRITE_SIZE) folded into per-kernel HBM bytes.  Output: gpurun_out/prof_g/.
set -e
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/prof_g"
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --episodes 16384 --no-cpu-baseline --no-train-probe --no-aux"
# the kernel statistics cover the TIMED region only (--no-profile: no second, bracketed pass in the trace); the line carries the device-side
# row accounting of exactly that region (k_conv3_auto_accounting_timed_region)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o bench -- python3 "$R/bench.py" $ARGS --no-profile > "$O/bench_under_rocprof.json" 2> "$O/err1.log"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O" -o fetch -- python3 "$R/bench.py" $ARGS --no-profile > "$O/bench_fetch.json" 2> "$O/err2.log"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O" -o write -- python3 "$R/bench.py" $ARGS --no-profile > "$O/bench_write.json" 2> "$O/err3.log"
python3 "$R/tools/pmc_traffic.py" "$O/fetch_counter_collection.csv" "$O/write_counter_collection.csv" "$O/bench_fetch.json" "$O/pmc_traffic.json" "${AZ_GIT_SHA:-}"
# the raw traces are large: keep the summaries only
for f in bench_kernel_trace.csv fetch_kernel_trace.csv write_kernel_trace.csv fetch_counter_collection.csv write_counter_collection.csv; do
    [ -f "$O/$f" ] && rm -f -- "$O/$f"
done
ls "$O"
