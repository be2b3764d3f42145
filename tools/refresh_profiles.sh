#!/bin/bash
# Re-create the measurements kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/refresh_profiles.sh <tag> [quick]     e.g. r2_final  -> gpurun_out/<tag>_*
# bench line, rocprofv3 kernel stats of the same command, separate --pmc passes (one counter group
# per pass: more than that exceeds the hardware's counter slots and rocprofv3 aborts), and the
# bench lines of the other workloads / input variants.  Copy the files you want judged into profiles/.
set -eo pipefail
TAG=${1:-r5_final}
QUICK=${2:-}          # "quick" = headline only; "variants" / "big" = only that part (a gpurun call is limited to 20 minutes)
PART=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
if [ "$PART" = "all" ] || [ "$PART" = "quick" ] || [ "$PART" = "headline" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o stats -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 1 \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
cp "$(find "$OUT/prof_$TAG" -name 'stats_kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
echo "[refresh] kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$TAG" -o $c -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fused --profile-steps 1 > "$OUT/pmc_$c.log" 2>&1
  echo "[refresh] pmc $c done"
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d "$OUT/pmc_$TAG" -o l2 -- \
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fused --profile-steps 1 > "$OUT/pmc_l2.log" 2>&1
echo "[refresh] pmc l2 done"
python tools/pmc_summary.py "$(dirname "$(find "$OUT/pmc_$TAG" -name 'l2_counter_collection.csv' | head -1)")" \
    FETCH_SIZE WRITE_SIZE l2 > "$OUT/${TAG}_pmc_summary.json"
python tools/make_traffic_json.py "$OUT/${TAG}_pmc_summary.json" reddit_h1_d64 "$OUT/pmc_WRITE_SIZE.log" > "$OUT/${TAG}_pmc_traffic.json"
# the headline line AFTER the counter passes: bench.py quotes roofline.traffic from profiles/pmc_traffic.json when
# that file was measured on the same kernel sources (copy the new one into profiles/ afterwards)
cp "$OUT/${TAG}_pmc_traffic.json" profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 3 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "[refresh] bench done"
fi
[ "$PART" = "quick" ] || [ "$PART" = "headline" ] && { echo "[refresh] headline: done"; exit 0; }
if [ "$PART" = "all" ] || [ "$PART" = "variants" ]; then
# input variants of the headline workload: worst-case locality, signed values
python bench.py --alpha 0 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_reddit_alpha0_bench.json" 2>/dev/null
python bench.py --values normal --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_reddit_normal_bench.json" 2>/dev/null
# the same graph at other row widths / head counts (not BASELINE configs: how far the headline tuning carries)
for cfg in "128 1" "256 1" "16 4" "32 2" "32 8"; do
  set -- $cfg
  python bench.py --graph reddit --d $1 --heads $2 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_reddit_h$2_d$1_bench.json" 2>/dev/null
done
echo "[refresh] variants done"
python bench.py --graph harness --d 1024 --heads 1 --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_harness_d1024_bench.json" 2>/dev/null
python bench.py --graph harness --d 64 --heads 8 --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_harness_8x64_bench.json" 2>/dev/null
python bench.py --graph cora --steps 50 --warmup 5 > "$OUT/${TAG}_cora_bench.json" 2>/dev/null
python bench.py --graph cora --steps 50 --warmup 5 --no-cpu-baseline --hip-graph > "$OUT/${TAG}_cora_hipgraph_bench.json" 2>/dev/null
echo "[refresh] small shapes done"
# node labelings (round 5): ids sorted by degree, communities of consecutive ids
for lab in degree clustered; do
  python bench.py --labeling $lab --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_reddit_${lab}_bench.json" 2>/dev/null
  python bench.py --labeling $lab --d 256 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_reddit_${lab}_d256_bench.json" 2>/dev/null
done
echo "[refresh] labelings done"
fi
[ "$PART" = "variants" ] && { echo "[refresh] variants: done"; exit 0; }
python tools/tune_sweep.py --fused "" > "$OUT/${TAG}_fused_passes.txt" 2>/dev/null
python bench.py --graph products --d 16 --heads 8 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_products_h8_d16_bench.json" 2>/dev/null
python bench.py --graph products --d 128 --heads 8 --steps 3 --warmup 1 --cpu-sample-edges 200000 > "$OUT/${TAG}_products_h8_d128_bench.json" 2>/dev/null
echo "[refresh] products done"
# one-GPU rehearsal of rank 0's shard of the 8-way multi-GPU configs (timing only: exchanges = local copies)
python bench.py --emulate-world 8 --graph papers100m --steps 5 --warmup 2 > "$OUT/${TAG}_emulate8_papers100m_bench.json" 2>/dev/null
python bench.py --emulate-world 8 --graph rmat25 --steps 3 --warmup 1 > "$OUT/${TAG}_emulate8_rmat25_bench.json" 2>/dev/null
# the REAL collective path on this one GPU (round 4): one-rank nccl (= RCCL) process group, self-halo, async all-to-alls
python bench.py --gpus 1 --graph papers100m --rccl-self --steps 5 --warmup 2 > "$OUT/${TAG}_rccl_self_papers100m_bench.json" 2> "$OUT/${TAG}_rccl_self.err"
python tools/time_fp64.py > "$OUT/${TAG}_fp64_timing.txt" 2>/dev/null
echo "[refresh] all done"
