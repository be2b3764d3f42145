set -e
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --emulate-world 8 --graph papers100m --steps 5 --warmup 2 > gpurun_out/r5_final_emulate8_papers100m_run$i.json 2>/dev/null
  timeout -k 10 300 python bench.py --gpus 1 --graph papers100m --rccl-self --steps 5 --warmup 2 > gpurun_out/r5_final_rccl_self_papers100m_run$i.json 2>/dev/null
done
timeout -k 10 300 python bench.py --emulate-world 8 --graph rmat25 --steps 3 --warmup 1 > gpurun_out/r5_final_emulate8_rmat25_bench.json 2>/dev/null
python - <<'PY'
import json,glob
for k in ("emulate8","rccl_self"):
    v=[]
    for f in sorted(glob.glob("gpurun_out/r5_final_%s_papers100m_run*.json"%k)):
        l=json.loads([x for x in open(f).read().splitlines() if x.startswith("{")][-1])
        v.append((round(l["ms_per_step"],2), round(l["exposed_exchange_ms"],2), l["config"]["schedule"]["kv_packed"], l["config"]["schedule"]["columns_fused"]))
    print(k, v)
PY
