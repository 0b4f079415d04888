#!/bin/bash
# Kernel-trace timeline of ONE sequential U+L step (Upper stage, then Lower stage): scripts/trace_timeline.py on a short trace.
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/tl_seq
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out/trace" -o bench -- python3 "$root/bench.py" --steps 6 --warmup 2 --trace-only --sequential > "$out/trace.log" 2>&1
db=$(find "$out/trace" -name "*.db" | head -1)
python3 "$root/scripts/trace_timeline.py" "$db" > "$out/timeline.txt"
rm -rf "$out/trace"
tail -3 "$out/timeline.txt"
