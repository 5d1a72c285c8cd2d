set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06z; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_models.py -m gpu -q -k "fused_amp_convs or engine_built" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hifigan_$c -- python $R/tools/probe_model.py hifigan 32 2 > $O/pmc_hifigan_$c.log 2>&1 || true
done
python $R/tools/pmc_summary.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE 2 $O/hifigan_hbm_traffic.json > $O/hifigan_hbm_traffic.txt 2>&1
python $R/tools/pmc_traffic.py $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE $O/traffic.json --build r06z
cp $O/traffic.json $R/profiles/traffic.json; cp $O/traffic.json $O/traffic_merged.json
cd $R
timeout 1200 python bench.py --profile-json $O/bench_kernels_hipevents.json > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python tools/driver_repro.py $O r06z $O/bench.json > $O/driver_repro.md 2>> $O/bench.err
rm -rf $O/pmc_hifigan_FETCH_SIZE $O/pmc_hifigan_WRITE_SIZE
python -c "
import json
j=json.load(open('$O/bench.json')); r=j['roofline']
print(j['ms_per_step'], r['avg_ms'], r['frac'], r['traffic'], r['traffic_source'])
t=json.load(open('$O/traffic.json')); print(len(t), [k for k in t if '1376' in k])
"
