R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in main $VARIANTS main $VARIANTS; do
  echo "== $v"
  if [ $v = main ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so; fi
  python tools/probe_wino_time.py 2>&1 | grep "C="
done
