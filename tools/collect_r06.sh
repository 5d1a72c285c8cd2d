# round-6 evidence: bash tools/collect_r06.sh <tag> [tests]   (run through gpurun from the repo root; ~15 min with tests)
set -x
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -5 $O/pytest_gpu.log
fi
cd /tmp && export TMPDIR=/tmp
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they cannot share one), kernel trace only
for m in hifigan bigvgan vocos; do
  B=$([ $m = hifigan ] && echo 32 || ([ $m = bigvgan ] && echo 64 || echo 128))
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${m}_$c -- python $R/tools/probe_model.py $m $B 2 > $O/pmc_${m}_$c.log 2>&1 || true
  done
  python $R/tools/pmc_summary.py $O/pmc_${m}_FETCH_SIZE $O/pmc_${m}_WRITE_SIZE 2 $O/${m}_hbm_traffic.json > $O/${m}_hbm_traffic.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE $O/traffic.json --build $TAG
# ... and becomes profiles/traffic.json of this copy, so that the bench line below finds its dominant kernel in it
# (written FRESH from this build's passes, not merged into the previous file: a key that no longer matches must show up as a missing entry, not as an old one — LOG R6.15)
cp $O/traffic.json $R/profiles/traffic.json; cp $O/traffic.json $O/traffic_merged.json
cd $R
timeout 1200 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
# how `roofline.traffic` was measured, with the by-kernel tables of the three configs
python tools/driver_repro.py $O $TAG $O/bench.json > $O/driver_repro.md 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
# the roofline section of bench.py itself under the profiler (single stream, every launch comparable): its hipEvent averages and rocprofv3's agree
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -- python $R/bench.py --roofline-only > $O/bench_roofline_only.json 2> /dev/null
cp $(ls -t $(find $O/prof_ro -name "*kernel_stats.csv") | head -1) $O/bench_roofline_only_kernel_stats.csv
python $R/tools/trace_stats_by_grid.py $O/prof_ro $O/bench_roofline_only_kernel_stats_by_grid.csv; rm -rf $O/prof_ro
for m in hifigan bigvgan vocos; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python $R/tools/probe_model.py $m > $O/prof_$m.log 2>&1
  cp $(find $O/prof_$m -name "*kernel_stats.csv" | head -1) $O/${m}_kernel_stats_serialized.csv
  rm -rf $O/prof_$m
done
# the replayed step as shipped (branch streams): per stage wall / kernel-sum / concurrency; a single clip's replayed forward by queue
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_step -- python $R/tools/probe_step.py hifigan > /dev/null 2>&1
python $R/tools/step_timeline.py $O/trace_step > $O/step_timeline.txt 2>&1; rm -rf $O/trace_step
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_b1 -- python $R/tools/probe_latency.py > /dev/null 2>&1
python $R/tools/latency_timeline.py $O/trace_b1 76 > $O/b1_timeline.txt 2>&1; rm -rf $O/trace_b1
cd $R
TOP=80 python tools/probe_latency.py > $O/latency_b1.txt 2>&1
bash tools/pmc_clock.sh 32 > $O/clock_per_kernel.txt 2>&1
bash tools/power_trace.sh 6 $O/power_trace.txt > /dev/null 2>&1; rm -f $O/power_trace.txt.idle $O/power_trace.txt.samples
# matrix-pipe counters of the dominant conv: this round's kernel, and with the row-split epilogue off
bash tools/pmc_conv.sh 128 5504 11 1 > $O/conv_pmc.txt 2>&1
{ echo "== the same layer without the row-split 16-byte-store epilogue (FV_X_W44_NO_QR=1)"; FV_X_W44_NO_QR=1 bash tools/pmc_conv.sh 128 5504 11 1; } >> $O/conv_pmc.txt 2>&1
# the fused narrow pairs: F(4,4) (round 5) next to F(2,3) (FV_PAIR_WINO44=0), SQ counters
for s in "16 44032 3 1" "16 44032 11 1" "16 44032 7 1" "32 22016 11 1" "32 22016 7 1" "64 11008 3 1" "128 5504 3 1"; do
  echo "== C T k d = $s (default form)" >> $O/pair_pmc.txt; bash tools/pmc_pair.sh $s >> $O/pair_pmc.txt 2>&1
  case "$s" in *" 11 1"|*" 7 1") echo "== C T k d = $s (F(2,3) form, FV_PAIR_WINO44=0)" >> $O/pair_pmc.txt; FV_PAIR_WINO44=0 bash tools/pmc_pair.sh $s >> $O/pair_pmc.txt 2>&1;; esac
done
bash tools/pmc_kernel.sh bigvgan aa_snake 64 > $O/aa_snake_pmc.txt 2>&1
python tools/probe_pair_wino.py 128 64 32 16 > $O/pair_wino_vs_direct.txt 2>&1
python tools/probe_pair44.py shipped 32 16 > $O/pair44_standalone.txt 2>&1
{ echo "== F(2,3) pairs everywhere (FV_PAIR_WINO44=0)"; FV_PAIR_WINO44=0 python tools/probe_pair_wino.py 32 16; } >> $O/pair_wino_vs_direct.txt 2>&1
# seeded differential fuzzers against the CPU oracle on this build (~3 min)
timeout 900 python tools/fuzz_all.py > $O/fuzz_all.txt 2>&1; grep -v amdgpu.ids $O/fuzz_all.txt | tail -12
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE gpurun_out/pmc_conv_a gpurun_out/pmc_pair_? gpurun_out/pmck_?
ls $O
