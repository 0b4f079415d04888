#!/bin/bash
# Wave-state / instruction-mix / L2 counters of the dominant kernels (separate rocprofv3 --pmc passes, never combined with tracing):
#   usage (through gpurun): bash scripts/pmc_wave_states.sh <tag> -- <program and arguments>
# e.g.  bash scripts/pmc_wave_states.sh step -- python3 bench.py --steps 3 --warmup 1 --trace-only --no-graph
# Output: gpurun_out/pmc_ws_<tag>/summary.txt (per kernel: averages per launch and the fractions of wave cycles).
set -e -o pipefail
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_ws_$tag
rm -rf "$out"; mkdir -p "$out"
cmd=("$@")
[[ "${cmd[1]}" != /* ]] && cmd[1]="$root/${cmd[1]}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F SQ_INSTS_VALU_MFMA_MOPS_BF SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- "${cmd[@]}" > "$out/p$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:64]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "TCC_REQ_sum"):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
tot = {k: sum(dur[k]) for k in acc}
with open(root + "/summary.txt", "w") as fo:
    for k in sorted(acc, key=lambda k_: -tot.get(k_, 0))[:12]:
        m = {n: sum(v) / len(v) for n, v in acc[k].items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        line = ["%s  (%d launches, %.1f us avg in the counter pass)" % (k, len(acc[k].get("SQ_WAVE_CYCLES", [])), sum(dur[k]) / max(len(dur[k]), 1))]
        if wc:
            f_ = lambda n: m.get(n, 0.0) / wc
            line.append("  of wave cycles: wait_any %.3f  wait_inst_any %.3f (lds %.3f)  active_inst_any %.3f [valu %.3f lds %.3f vmem %.3f sca %.3f misc %.3f]"
                        % (f_("SQ_WAIT_ANY"), f_("SQ_WAIT_INST_ANY"), f_("SQ_WAIT_INST_LDS"), f_("SQ_ACTIVE_INST_ANY"), f_("SQ_ACTIVE_INST_VALU"),
                           f_("SQ_ACTIVE_INST_LDS"), f_("SQ_ACTIVE_INST_VMEM"), f_("SQ_ACTIVE_INST_SCA"), f_("SQ_ACTIVE_INST_MISC")))
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            line.append("  per launch: %.0f k GPU cycles; MFMA busy %.3f of SIMD cycles; insts: vmem_rd %.0f vmem_wr %.0f lds %.0f valu %.0f salu %.0f; mfma Mops f32 %.0f bf16 %.0f; "
                        "LDS data-fifo-full %.0f cmd-fifo-full %.0f; avg in flight: vmem %.1f lds %.1f per busy cycle"
                        % (cyc / 1e3, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc), m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_VMEM_WR", 0),
                           m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_VALU_MFMA_MOPS_F", 0),
                           m.get("SQ_INSTS_VALU_MFMA_MOPS_BF", 0), m.get("SQ_LDS_DATA_FIFO_FULL", 0), m.get("SQ_LDS_CMD_FIFO_FULL", 0),
                           m.get("SQ_INST_LEVEL_VMEM", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1), m.get("SQ_INST_LEVEL_LDS", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1)))
        if "TCC_REQ_sum" in m:
            line.append("  L2: %.0f requests, hit %.3f, EA read requests %.0f" % (m["TCC_REQ_sum"], m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1), m.get("TCC_EA0_RDREQ_sum", 0)))
        fo.write("\n".join(line) + "\n\n")
print(open(root + "/summary.txt").read())
PY
