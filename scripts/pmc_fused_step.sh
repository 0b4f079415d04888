#!/bin/bash
# PMC passes (HBM fetch / write bytes, MFMA busy) around scripts/bench_fused_step.py.
# usage (through gpurun): bash scripts/pmc_fused_step.sh <tag>
set -e -o pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_fused_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d "$out/$name" -o pmc -- python3 "$root/scripts/bench_fused_step.py" > "$out/$name.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "lstm_step_bf16" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["us:" + r["Counter_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    line = k + ": launches %d" % len(c.get("FETCH_SIZE", []))
    if "FETCH_SIZE" in m:
        line += "  fetch %.0f MB (x2 corrected)  write %.0f MB  us %.0f" % (2 * m["FETCH_SIZE"] / 1024, m.get("WRITE_SIZE", 0) / 1024, m["us:FETCH_SIZE"])
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += "  mfma_busy %.3f  clk %.2f GHz" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), cyc / m["us:GRBM_GUI_ACTIVE"] / 1e3)
    print(line)
PY
