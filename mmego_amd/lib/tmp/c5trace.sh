cd /tmp && export TMPDIR=/tmp
for i in 1; do
  rm -rf /tmp/c5t$i
  rocprofv3 --kernel-trace --stats -d /tmp/c5t$i -o c5 -- python3 $GRAFT_REPO_ROOT/scripts/bench_config5.py --trace > /tmp/c5t$i.log 2>&1
  db=$(find /tmp/c5t$i -name "*.db" | head -1)
  python3 - "$db" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, end-start from kernels order by start"))
agg = collections.defaultdict(list)
for n, d in rows: agg[n.split("(")[0].replace("void ", "")].append(d)
for k in ("attn_pool_frag_bf16_kernel<8>", "upper_front_eval_bf16_kernel", "cross_attn_fwd_mfma_kernel<true, true>", "gemm_tile_kernel<64, 64, 2, 2, true, true, 32>", "lstm_step_bf16_fused256_kernel<false>", "topk_rows_sort_kernel<4>", "group_sum2_kernel", "mlp3_eval_bf16_kernel"):
    v = agg.get(k, [])
    print(k, [round(x / 1e3, 1) for x in v])
PY
done
