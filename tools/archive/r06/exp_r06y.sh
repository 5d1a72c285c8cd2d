set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
for r in 1 2 3; do
for v in 0 1; do
export FV_X_UPS_TILE256=$v
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives --profile-json $O/prof_${v}_$r.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('tile256=$v round $r: ms/step %.3f  p50 %.3f  eager p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['p50_clip_latency_eager_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
done
done
python - <<PY
import json
for v in (0,1):
    for r in (1,2,3):
        t=json.load(open('$O/prof_%d_%d.json'%(v,r)))
        print('tile256=%d'%v, r, [(x['kernel'].split('tile=')[1].split('>')[0], round(x['avg_ms']*1e3,1)) for x in t if 'convT' in x['kernel']])
PY
FV_X_UPS_TILE256=1 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "hifigan" 2>&1 | tail -2
