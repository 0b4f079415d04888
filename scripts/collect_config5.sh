#!/bin/bash
# rocprofv3 evidence for BASELINE config 5 (B=2048 T=16 N=256, bf16-operand IMU products), run through gpurun from the repo root:
#   1. the plain benchmark (kernels / IMU forward / whole forward, fp32 and bf16)
#   2. kernel trace + stats of the config-5 forward in bf16 mode (1 warm-up + 2 forwards)
#   3. PMC passes of the same run (one counter group per run, never combined with tracing)
# Everything lands in gpurun_out/prof_<tag>_c5/ ; scripts/config5_summary.py condenses it into profiles/<tag>_config5.json.
set -e -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_c5
mkdir -p "$out"
python3 "$root/scripts/bench_config5.py" > "$out/bench_config5.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o c5 -- python3 "$root/scripts/bench_config5.py" --trace > "$out/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -o pmc -- python3 "$root/scripts/bench_config5.py" --trace > "$out/pmc_$name.log" 2>&1
done
echo "config-5 profiles collected in $out"
