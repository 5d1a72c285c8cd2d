for v in base x_da3 x_da6 x_occ2 x_occ4; do
  if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_$v.so; fi
  echo "== $v"; python tools/probe_pair_wino.py 2>/dev/null | grep -v "k=3" | awk '{print $1,$2,$3, $10, $11}' | tr '\n' ';'; echo
done
unset FV_LIB_PATH
bash tools/ab_libs.sh "base x_da3 x_da6 x_occ2 x_occ4" 2
