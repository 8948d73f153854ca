cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -5
timeout 900 python bench.py --dtype bf16 --no-cpu-baseline --dump-shapes gpurun_out/shapes_bf16a.txt > gpurun_out/bench_bf16a.json 2> gpurun_out/bench_bf16a.err; echo rc=$?
tail -3 gpurun_out/bench_bf16a.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/bench_bf16a.json") if l.startswith("{")][-1])
print("fps", j["value"], "ms/step", j["ms_per_step"], j.get("batch_consistency"))
print(json.dumps(j.get("roofline"), indent=0)[:600])
for k, v in sorted(j.get("kernels", {}).items(), key=lambda kv: -kv[1].get("ms_per_step", 0)): print(k, v)
PY
head -30 gpurun_out/shapes_bf16a.txt
