# bash tools/flake_loop.sh <n> [lib]: repeat the BigVGAN/Vocos batch-vs-single property test n times, count failures
N=${1:-20}
[ -n "$2" ] && export FV_LIB_PATH=$GRAFT_REPO_ROOT/$2
f=0
for i in $(seq $N); do
  python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "benchmark_sizes_properties and f16x3" 2>&1 | tail -1 | grep -q passed || { f=$((f+1)); echo "fail at $i"; }
done
echo "failures $f / $N (lib=${FV_LIB_PATH:-shipped})"
