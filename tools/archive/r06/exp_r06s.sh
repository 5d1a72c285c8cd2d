set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for r in 1 2 3; do
for v in walk nowalk; do
if [ $v = nowalk ]; then export FV_X_SUM3_NO_WALK=1; else unset FV_X_SUM3_NO_WALK; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives --profile-json $O/prof_${v}_$r.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v round $r: ms/step %.3f  p50 %.3f  eager p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['p50_clip_latency_eager_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
done
done
unset FV_X_SUM3_NO_WALK
python - <<PY
import json
for v in ('walk','nowalk'):
    t=json.load(open('$O/prof_%s_1.json'%v))
    print(v)
    for r in t:
        if 'post' in r['kernel'] or 'convT' in r['kernel']: print('  ',r['kernel'], r['launches'], round(r['avg_ms']*1e3,1))
PY
