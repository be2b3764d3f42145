set -eo pipefail
TAG=r2_mid; OUT=gpurun_out
python bench.py --graph products --d 128 --heads 8 --steps 3 --warmup 1 --cpu-sample-edges 1500000 --verbose > "$OUT/${TAG}_products_h8_d128_bench.json" 2> "$OUT/${TAG}_products.err"
echo "[rest] products done"
python bench.py --emulate-world 8 --graph papers100m --steps 5 --warmup 2 > "$OUT/${TAG}_emulate8_papers100m_bench.json" 2>/dev/null
echo "[rest] papers done"
python bench.py --emulate-world 8 --graph rmat25 --steps 3 --warmup 1 --verbose > "$OUT/${TAG}_emulate8_rmat25_bench.json" 2> "$OUT/${TAG}_rmat.err"
echo "[rest] rmat done"
