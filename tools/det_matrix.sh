echo "== shipped (tile2 on)"; python tools/probe_determinism.py bigvgan 1 300 f16x3 0 | tail -1
echo "== nop variant (tile2 on)"; FV_LIB_PATH=$GRAFT_REPO_ROOT/vocoder_amd/csrc/libfishvoc_xn.so python tools/probe_determinism.py bigvgan 1 300 f16x3 0 | tail -1
