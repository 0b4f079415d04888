#!/bin/bash
# Round-6 evidence in one gpurun call (run from the repo root): kernel trace + stats of the default bench command and of the sequential
# run, the same two with both IMU_Net forwards in the split3 mode (MMEGO_IMU_PRECISION=split3), the launch-by-launch timelines of one
# sequential step (fp32 and split3), PMC passes (HBM fetch / write bytes, MFMA busy, LDS conflicts) around bench.py itself in both modes
# and around the UpperNetwlocal step -- condensed ON THE BOX into gpurun_out/prof_r06/ (the rocpd databases exceed the merge-back limit).
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_r06
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for prec in fp32 split3; do
  export MMEGO_IMU_PRECISION=$prec
  for mode in concurrent sequential; do
    flag=""; [ $mode == sequential ] && flag="--sequential"
    rocprofv3 --kernel-trace --stats -d "$out/trace_${prec}_$mode" -o bench -- python3 "$root/bench.py" --steps 20 --warmup 3 --trace-only $flag > "$out/trace_${prec}_$mode.log" 2>&1
    db=$(find "$out/trace_${prec}_$mode" -name "*.db" | head -1)
    python3 "$root/scripts/prof_summary.py" "$db" --grids > "$out/kernel_stats_${prec}_$mode.csv"
    # (a sequential U+L step ends with its second Adam launch)
    [ $mode == sequential ] && (python3 "$root/scripts/trace_timeline.py" "$db" --marker adam_kernel --per-step 2 > "$out/timeline_${prec}_sequential.txt" || true)
    rm -rf "$out/trace_${prec}_$mode"
    echo "trace $prec $mode done"
  done
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo "$grp" | cut -d' ' -f1)
    tag=bench; [ $prec == split3 ] && tag=split3
    rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_${tag}_$name" -o pmc -- python3 "$root/bench.py" --steps 4 --warmup 2 --trace-only --no-graph > "$out/pmc_${tag}_$name.log" 2>&1
    echo "pmc $prec $name done"
  done
done
unset MMEGO_IMU_PRECISION
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  MMEGO_WLOCAL_EAGER=1 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_wlocal_$name" -o pmc -- python3 "$root/bench.py" --wlocal-only --trace-only > "$out/pmc_wlocal_$name.log" 2>&1
  echo "pmc wlocal $name done"
done
python3 "$root/scripts/pmc_summary.py" "$out" > "$out/pmc_counters.json"
# keep the raw counter rows of the kernels the summary names (the full tables are tens of MB)
python3 - "$out" <<'PY'
import glob, os, re, shutil, sys
out = sys.argv[1]
pat = re.compile("gemm_tile_big|gemm_tile_persistent|lstm_step_dma_kernel|lstm_seq_xcd|tconv_seq|gcn_front|graph_dA_fused|mlp_bwd_layer|mlp_fwd_layer|local_group_l1|pool8|s3_gemm|s3_gemm_big|s3_step|s3_cvt|vox_")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if fs:
        with open(fs[0]) as f, open(d + ".csv", "w") as o:
            n = 0
            for i, line in enumerate(f):
                if i == 0 or (pat.search(line) and n < 3000):
                    o.write(line)
                    n += i > 0
    shutil.rmtree(d)
PY
bash "$root/scripts/collect_wlocal.sh" r06 > "$out/collect_wlocal.log" 2>&1 || true
ls -la "$out"
