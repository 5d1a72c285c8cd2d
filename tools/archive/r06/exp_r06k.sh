set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06k; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python tools/probe_w44_ablation.py base >> $O/w44_standalone.txt 2>&1
grep -v amdgpu.ids $O/w44_standalone.txt
python bench.py > $O/bench.json 2> $O/bench.err; head -c 1500 $O/bench.json
