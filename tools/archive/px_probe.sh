for px in 1 2 4 8; do echo "== PX=$px"; FV_PW_PX=$px FV_PROBE_ROUNDS=3 python tools/probe_pointwise.py 0,1 2>&1 | grep -v amdgpu | head -12 | cut -c1-100; done
