#!/bin/bash
# One GPU-box pass that produces everything tools/refresh_profiles.py turns into profiles/: the bench line, the rocprofv3
# kernel trace of the same command (minus the CPU leg and the one-pair bitstream extra) and the two PMC passes.
# usage (from the repo root, through gpurun): bash tools/profile_round.sh
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
rm -rf "$R/gpurun_out/prof_r01" "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write"
cd "$R"
timeout -k 10 400 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r01" -o r01 -- python3 "$R/bench.py" --no-cpu-baseline --no-codec > "$R/gpurun_out/prof_bench.log" 2>&1
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --no-f32-compare --no-graph --no-codec"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -o f -- python3 "$R/bench.py" $PMC_ARGS > "$R/gpurun_out/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -o w -- python3 "$R/bench.py" $PMC_ARGS > "$R/gpurun_out/pmc_write.log" 2>&1
cd "$R"
# keep the merge small: the trace database and the counter CSVs are all refresh_profiles.py reads
find gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write -type f ! -name '*results.db' ! -name '*counter_collection.csv' -delete
ls -la gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -c 600 gpurun_out/bench_n1.json
