#!/bin/bash
# Socket power and shader clock while the headline step replays: is the step power-capped?
#   bash tools/power_trace.sh [seconds] -> gpurun_out/<tag>/power_trace.txt   (rocm-smi sampled at ~10 Hz beside a replay loop)
R=${GRAFT_REPO_ROOT:-$(pwd)}
SECS=${1:-6}
OUT=${2:-$R/gpurun_out/power_trace.txt}
cd $R
rocm-smi --showmaxpower --showpower --showclocks > $OUT.idle 2>&1
python - <<PY &
import sys, time, torch
sys.path.insert(0, "$R")
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0))
mel = torch.from_numpy(syn.synthetic_mel(32, 80, 86, 1234)).cuda()
out = torch.empty((32, 1, eng.output_length(86)), device="cuda")
for _ in range(5): eng(mel, out)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < $SECS:
    for _ in range(20): eng(mel, out)
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
print("replayed %d steps in %.2f s: %.3f ms/step" % (n, dt, dt / n * 1e3), flush=True)
PY
PID=$!
sleep 1.5
: > $OUT.samples
for i in $(seq 1 $((SECS * 10 - 20))); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | tr '\n' ' ' >> $OUT.samples
  echo >> $OUT.samples
  sleep 0.05
done
wait $PID
python - <<PY > $OUT
import re
idle = open("$OUT.idle").read()
print("idle / limits:"); print("\n".join(l for l in idle.splitlines() if re.search(r"Power|sclk|Max", l)))
pw, ck = [], []
for l in open("$OUT.samples"):
    m = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m: pw.append(float(m.group(1)))
    if c: ck.append(int(c.group(1)))
if pw: print("under the replayed B = 32 step: %d samples, socket power min / mean / max %.0f / %.0f / %.0f W" % (len(pw), min(pw), sum(pw) / len(pw), max(pw)))
if ck: print("sclk min / mean / max %d / %d / %d MHz" % (min(ck), sum(ck) / len(ck), max(ck)))
PY
cat $OUT
