R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04q
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err; python - <<PY
import json
j=json.load(open("$O/bench.json"))
print({k:j[k] for k in ("value","ms_per_step","ms_per_step_eager","p50_clip_latency_ms") if k in j})
print("roofline", {k:j["roofline"].get(k) for k in ("kernel","avg_ms","achieved","frac","algorithmic_tflops","algorithmic_speedup")})
print("step", j.get("roofline_step"))
for o in j.get("other_configs",[]): print(o.get("model"), o.get("ms_per_step"), o.get("roofline_step",{}).get("frac"))
print(j.get("with_collectives",{}).get("ms_per_step"))
PY
