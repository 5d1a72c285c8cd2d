set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for r in 1 2 3; do
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_pqphase.so python tools/probe_pair44.py phase 32 >> $O/pair44_standalone.txt 2>&1
  python tools/probe_pair44.py inloop 32 >> $O/pair44_standalone.txt 2>&1
done
grep -v amdgpu.ids $O/pair44_standalone.txt
bash tools/ab_libs.sh "x_pqphase base" 3 > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
python bench.py > $O/bench.json 2> $O/bench.err; head -c 2500 $O/bench.json
