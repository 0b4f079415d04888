set -e
for r in 1 2; do
  for v in 0 1; do
    echo "SPLIT_IMU=$v run $r"
    MMEGO_SPLIT_IMU=$v python bench.py --trace-only --steps 300 --warmup 10
  done
done
