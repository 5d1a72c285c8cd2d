mkdir -p gpurun_out/lat44
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/lat44/gpu_suite.txt
python tools/fuzz_all.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lat44/fuzz_all.txt
