mkdir -p gpurun_out/lat44
bash tools/ab_env.sh FV_X_LAT44_ROWS "0 32" 4 --steps 5 2>&1 | tee gpurun_out/lat44/ab_rows_bench.txt
