#!/bin/bash
# r06: the fp32 recurrent step's time taken apart by in-kernel stamps (scripts/clock_probe.hip, built HERE into
# mmego_amd/lib/variants/clock_probe with -DMMEGO_STAMP) -- both launch forms (ndir = 2: one 256-workgroup launch of 64 x 32 x 4 tiles
# per timestep; ndir = 1: the two-chain form's single-direction launches of 64 x 16 x 4 tiles, two workgroups per CU) and the probe's
# elimination bits (1: loaders do not wait for their transfers, 2: no transfers, 4: no LDS reads).  Output: the table under profiles/.
root=${GRAFT_REPO_ROOT:-$(pwd)}
p=$root/mmego_amd/lib/variants/clock_probe
for ndir in 2 1; do
  for dbg in 0 2 6; do
    echo "== ndir $ndir, elimination bits $dbg"
    PROBE_NDIR=$ndir PROBE_STEP_DBG=$dbg PROBE_SKIP_GEMM=1 timeout -k 5 60 $p 1 2>&1 | grep "lstm_step"
  done
done
