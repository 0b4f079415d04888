"""Condense the PMC passes of scripts/collect_profiles.sh into per-launch counter averages for the two dominant kernels.
usage: python scripts/pmc_summary.py gpurun_out/prof_<tag> > profiles/<tag>_pmc_counters.json"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
# kernel-name prefix -> counter-pass directory stem.  From r03 the passes wrap bench.py itself ("bench"); the r02 layout (passes around
# the micro-benchmarks: "step" / "gemm") is still understood.
KERNELS = {"gemm_tile_big_kernel": ("bench",), "lstm_step_dma_kernel<32>": ("bench", "step"), "lstm_step_dma_kernel<16>": ("bench",), "lstm_step_small_kernel": ("bench",),
           "gemm_tile_persistent_kernel": ("bench", "gemm"), "lstm_step_dma_kernel": ("step",),
           # r04: the fused ST-GCN step's kernels and the train-mode PointNet backward layer (the dominant kernel of the UpperNetwlocal step)
           "tconv_seq_kernel": ("bench",), "gcn_front_kernel": ("bench",), "tconv_wgrad_kernel": ("bench",), "graph_dA_fused_kernel": ("bench",),
           "mlp_bwd_layer_kernel": ("bench", "wlocal"), "local_group_l1_kernel": ("wlocal",), "pool8_bwd_kernel": ("wlocal",),
           "pool8_bn_act_kernel": ("wlocal",),
           # r04, late: rnn_slow's recurrence as one persistent launch per layer (lstm_seq.hip)
           "lstm_seq_xcd_kernel": ("bench",),
           # r05: the split3 mode's kernels (passes with MMEGO_IMU_PRECISION=split3: "split3"), the PointNet forward layer, LocalVoxelNet's
           # kernels, the paired BiLSTM(64) launches
           "s3_gemm_kernel": ("split3",), "s3_gemm_big_kernel": ("split3",), "s3_step_kernel": ("split3",), "s3_cvt_kernel": ("split3",), "s3_fc_relu_kernel": ("split3",),
           "mlp_fwd_layer_kernel": ("bench", "wlocal"), "vox_l1_fwd_kernel": ("wlocal",), "vox_l1_bwd_kernel": ("wlocal",),
           "vox_dw_kernel": ("wlocal",), "vox_mid_fwd_kernel": ("wlocal",), "vox_mid_bwd_kernel": ("wlocal",),
           "lstm64_fwd_multi_kernel": ("wlocal",), "lstm64_bwd_multi_kernel": ("wlocal",)}
res = {}
for kname, stems in KERNELS.items():
    by_grid = collections.defaultdict(lambda: collections.defaultdict(list))
    files = []
    for which in stems:
        files = glob.glob(os.path.join(root, "pmc_%s_*" % which, "**", "*counter_collection.csv"), recursive=True)
        if files:
            break
    for f in files:
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                by_grid["all"][r["Counter_Name"]].append(float(r["Counter_Value"]))
                by_grid["dur"][r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    c = {k: sum(v) / len(v) for k, v in by_grid["all"].items()}
    if not c:
        continue
    entry = {"counters_avg_per_launch": c, "launches_sampled": {k: len(v) for k, v in by_grid["all"].items()}}
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 256 CUs x 4 SIMDs
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        entry["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        entry["lds_conflict_fraction_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md): x2
        entry["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        d = by_grid["dur"]["FETCH_SIZE"]
        us = sum(d) / len(d) / 1e3                       # kernel duration inside the (serialising) counter pass
        entry["avg_us_in_counter_pass"] = us
        entry["hbm_GBps"] = entry["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
        entry["hbm_fraction_of_8TBps"] = entry["hbm_GBps"] / 8000.0
    res[kname] = entry
res["note"] = ("separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ group) around `bench.py --trace-only --no-graph` itself "
               "(scripts/collect_profiles.sh); FETCH doubled per MI355X_MICROARCH.md; per-launch averages over all launches of the "
               "kernel in that process (gemm_tile_persistent_kernel: the K = 512 and K = 1024 projection products of both IMU_Net forwards)")
print(json.dumps(res, indent=1))
