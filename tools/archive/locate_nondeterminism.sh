# Locates a run-to-run difference of the BigVGAN engine by stage / dilation pair (run on the GPU box; FV_LIB_PATH may point at
# a variant build): FV_DEBUG_STOP codes are stage * 100 + pair * 10 + half, see tools/probe_f16_locate.py
for code in ${CODES:-21 121 221 321 999}; do FV_DEBUG_STOP=$code python tools/probe_f16_locate.py ${N:-300} ${PREC:-f16x3} 2>&1 | grep -v -e amdgpu.ids -e "^FV_DEBUG_STOP: " | tail -8; done
