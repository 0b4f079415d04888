#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench run (concurrent stage graphs) and of --sequential
#   2. PMC passes (one counter group per run, never combined with tracing) around bench.py ITSELF (eager launches:
#      --trace-only --no-graph), so that the counters belong to the benchmarked process; scripts/pmc_summary.py picks the
#      dominant kernels out by name
# Everything lands in gpurun_out/prof_<tag>/ ; scripts/prof_summary.py and scripts/pmc_summary.py condense it.
set -e -o pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace_concurrent" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 3 --trace-only > "$out/trace_concurrent.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace_sequential" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 3 --trace-only --sequential > "$out/trace_sequential.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_bench_$name" -o pmc -- python3 "$root/bench.py" --steps 4 --warmup 2 --trace-only --no-graph > "$out/pmc_bench_$name.log" 2>&1
done
echo "profiles collected in $out"
