set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06aa; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for r in 1 2 3; do
for v in y2 noy2; do
if [ $v = noy2 ]; then export FV_X_NO_Y2=1; else unset FV_X_NO_Y2; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives --profile-json $O/prof_${v}_$r.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v round $r: ms/step %.3f  eager %.3f  p50 %.3f  dominant %.1f us %.3f  finite %s' % (j['ms_per_step'], j['ms_per_step_eager'], j['p50_clip_latency_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac'], j['output_finite']))"
done
done
unset FV_X_NO_Y2
python - <<PY
import json
for v in ('y2','noy2'):
    t=json.load(open('$O/prof_%s_1.json'%v))
    tot=sum(r['total_ms'] for r in t)/3
    w=sum(r['total_ms'] for r in t if 'conv_wino44' in r['kernel'])/3
    print(v, 'serialized %.3f ms, conv_wino44 %.3f ms' % (tot, w))
    for r in t:
        if 'conv_wino44' in r['kernel'] and 'cin=128' in r['kernel']: print('   ', r['kernel'], r['launches'], round(r['avg_ms']*1e3,1))
PY
