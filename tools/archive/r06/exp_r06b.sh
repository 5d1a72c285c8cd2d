# round-6 experiment batch b: what bounds conv_wino44 (VALU share), are co-resident workgroups in lock-step, does a start-up stagger pay
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
bash tools/pmc_conv_full.sh 128 5504 11 1 > $O/conv_pmc_full.txt 2>&1
for v in ts ts_stag2; do
  echo "== $v" >> $O/w44_timeline.txt
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so FV_WINO=2 python tools/probe_conv_timeline.py 128 11 5504 1 32 res >> $O/w44_timeline.txt 2>&1
  FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so FV_WINO=2 python tools/probe_conv_timeline.py 128 7 5504 1 32 res >> $O/w44_timeline.txt 2>&1
done
for r in 1 2; do
  python tools/probe_w44_ablation.py shipped >> $O/w44_stagger_standalone.txt 2>&1
  for v in stag1 stag2; do FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_x_$v.so python tools/probe_w44_ablation.py $v >> $O/w44_stagger_standalone.txt 2>&1; done
done
bash tools/ab_libs.sh "base x_stag1 x_stag2" 2 > $O/ab_stagger_step.txt 2>&1
cat $O/conv_pmc_full.txt $O/w44_timeline.txt $O/w44_stagger_standalone.txt $O/ab_stagger_step.txt
