#!/bin/bash
# A/B of HIP-runtime environment knobs on the default U+L step (bench.py --trace-only): graph queue mapping, kernarg placement, ...
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --trace-only --steps 300 --warmup 10 2>&1 | tail -1; }
run X=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GPU_STREAMOPS_CP_WAIT=0
run GPU_STREAMOPS_CP_WAIT=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
# (DEBUG_HIP_DYNAMIC_QUEUES=1 is NOT run: it aborts the process inside the HIP runtime at start-up -- r03)
run X=0
