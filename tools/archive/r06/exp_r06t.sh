set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06t; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for r in 1 2 3; do
for v in base x_occ0; do
if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_$v.so; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives --profile-json $O/prof_${v}_$r.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v round $r: ms/step %.3f  p50 %.3f  eager p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['p50_clip_latency_eager_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
done
done
unset FV_LIB_PATH
python - <<PY
import json
for v in ('base','x_occ0'):
    for r in (1,2,3):
        t=json.load(open('$O/prof_%s_%d.json'%(v,r)))
        print(v, r, [round(x['avg_ms']*1e3,1) for x in t if 'convT' in x['kernel']])
PY
