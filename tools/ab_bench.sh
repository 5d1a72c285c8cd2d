# A/B of an experimental library build against the shipped one on the headline step, interleaved rounds:
#   bash tools/ab_bench.sh vocoder_amd/csrc/libfishvoc_x1.so [rounds]
X=$GRAFT_REPO_ROOT/$1
R=${2:-3}
for r in $(seq $R); do
  for v in base exp; do
    if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$X; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v round $r: ms/step %.3f  p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
  done
done
