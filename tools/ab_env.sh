# A/B of one environment knob on the headline step, interleaved rounds (the chip's clock follows its recent load):
#   bash tools/ab_env.sh FV_CONV_STAGGER "0 4 8" [rounds] [extra bench flags]
VAR=$1; VALS=$2; R=${3:-3}; shift 3
for r in $(seq $R); do
  for v in $VALS; do
    if [ "$v" = "0" ] && [ "$VAR" = "FV_X_W44_NO_QR" ]; then unset $VAR; E=""; else E="$VAR=$v"; fi; env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$VAR=$v round $r: ms/step %.3f  p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
  done
done
