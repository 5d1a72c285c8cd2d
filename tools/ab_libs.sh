#!/bin/bash
# Interleaved A/B of several library builds on the headline bench line (ms/step, p50 single-clip latency):
#   bash tools/ab_libs.sh "base x_ring11 x_ring15" [rounds]      (base = the shipped library; x_NAME = vocoder_amd/csrc/libfishvoc_x_NAME.so)
LIBS=$1; R=${2:-3}
for r in $(seq $R); do
  for v in $LIBS; do
    if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_$v.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v round $r: ms/step %.3f  p50 %.3f  eager p50 %.3f  dominant %.1f us %.3f' % (j['ms_per_step'], j['p50_clip_latency_ms'], j['p50_clip_latency_eager_ms'], j['roofline']['avg_ms']*1e3, j['roofline']['frac']))"
  done
done
