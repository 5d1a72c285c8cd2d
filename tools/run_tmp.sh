python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "latency" 2>&1 | tail -2
for r in 1 2 3; do
for v in base x_prev; do
if [ $v = base ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$PWD/vocoder_amd/csrc/libfishvoc_$v.so; fi
TOP=80 python tools/probe_latency.py 2>/dev/null | python -c "
import sys,re
tot=0; n=0
for l in sys.stdin:
    if 'p50' in l: p=l.split()[-1]
    if 'conv_wino_lat44' in l:
        m=re.search(r'([0-9.]+) ms x',l); tot+=float(m.group(1)); n+=1
print('$v p50', p, 'lat44 kernels serialized ms %.4f over %d rows'%(tot,n))"
done; done
