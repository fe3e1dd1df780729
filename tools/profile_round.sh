#!/bin/bash
# One GPU-box pass that produces everything tools/refresh_profiles.py turns into profiles/ (round tag as $1, default r02): the bench
# line (with its own PMC child passes), the rocprofv3 kernel trace of the same command (minus the CPU leg and the one-pair bitstream
# extra), one SQ-counter pass, and kernel traces of the training step / the CQE forward.
# usage (from the repo root, through gpurun): bash tools/profile_round.sh r02
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
rm -rf "$R/gpurun_out/prof_$TAG" "$R/gpurun_out/pmc_sq" "$R/gpurun_out/prof_train" "$R/gpurun_out/prof_cqe" "$R/gpurun_out/prof_cqetrain"
cd "$R"
timeout -k 10 700 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$TAG" -o $TAG -- python3 "$R/bench.py" --no-cpu-baseline --no-codec --no-pmc > "$R/gpurun_out/prof_bench.log" 2>&1 || echo "trace failed"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d "$R/gpurun_out/pmc_sq" -o s -- python3 "$R/bench.py" --pmc-child > "$R/gpurun_out/pmc_sq.log" 2>&1 || echo "sq failed"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_train" -o tr -- python3 "$R/tools/train_prof.py" bf16 > "$R/gpurun_out/prof_train.log" 2>&1 || echo "train trace failed"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_cqe" -o c -- python3 "$R/tools/cqe_prof.py" bf16 > "$R/gpurun_out/prof_cqe.log" 2>&1 || echo "cqe trace failed"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_cqetrain" -o ct -- python3 "$R/tools/cqe_prof.py" bf16 train > "$R/gpurun_out/prof_cqetrain.log" 2>&1 || echo "cqe train trace failed"
cd "$R"
# keep the merge small: the trace databases and the counter CSVs are all refresh_profiles.py reads
find gpurun_out/prof_$TAG gpurun_out/pmc_sq gpurun_out/prof_train gpurun_out/prof_cqe gpurun_out/prof_cqetrain -type f ! -name '*results.db' ! -name '*counter_collection.csv' -delete
ls -la gpurun_out/prof_$TAG gpurun_out/pmc_sq
tail -c 400 gpurun_out/bench_n1.json
