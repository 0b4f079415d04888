#!/bin/bash
# rocprofv3 evidence for the UpperNetwlocal path (run through gpurun from the repo root): kernel trace + stats of the training
# step's graph replays (bench.py --wlocal-only --trace-only), a one-step timeline, and PMC passes (HBM fetch / write bytes) of
# the same program run eagerly.  Lands in gpurun_out/prof_<tag>_wlocal/.
set -e -o pipefail
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_wlocal
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench -- python3 "$root/bench.py" --wlocal-only --trace-only > "$out/trace.log" 2>&1
db=$(find "$out/trace" -name "*.db" | head -1)
# 1 eager warm-up body + 1 capture (not executed) + 55 replays = 56 executed steps
python3 "$root/scripts/prof_summary.py" "$db" 56 --grids > "$out/kernel_stats.csv"
python3 "$root/scripts/trace_timeline.py" "$db" --marker adam_kernel > "$out/timeline.txt"
rm -rf "$out/trace"
if [ "$2" == "pmc" ]; then
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    MMEGO_WLOCAL_EAGER=1 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$grp" -o pmc -- python3 "$root/bench.py" --wlocal-only --trace-only > "$out/pmc_$grp.log" 2>&1
  done
fi
head -40 "$out/kernel_stats.csv"
tail -2 "$out/timeline.txt"
