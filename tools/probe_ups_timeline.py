#!/usr/bin/env python
"""Per-workgroup phase time stamps of the LAST tiled-conv launch of a HiFiGAN-V1 forward in profiling form (single stream, tree form: the stage-4 upsampler
ConvTranspose1d(32 -> 16) reading its three branch inputs — hifigan.py:230-231 behind the stack-mean of :132-133), library built with -DFV_X_CONV_TS:
    FV_LIB_PATH=.../libfishvoc_x_ts.so python tools/probe_ups_timeline.py [FV_DEBUG_STOP value to end the forward earlier]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    os.environ["FV_DEBUG_STOP"] = sys.argv[1]
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config

cfg = dict(syn.HIFIGAN_V1_44K)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=syn.hifigan_state_dict(cfg, 0))
mel = torch.from_numpy(syn.synthetic_mel(32, 80, 86, 1)).cuda()
for _ in range(2):
    eng(mel)
torch.cuda.synchronize()
ts = torch.zeros(16384 * 16, dtype=torch.int64, device="cuda:0")
L = _lib.lib()
L.fv_debug_set_splitk_timestamps.argtypes = [ctypes.c_void_p]
L.fv_debug_set_splitk_timestamps(ts.data_ptr())
tab = eng.profile(mel, repeats=1)
torch.cuda.synchronize()
L.fv_debug_set_splitk_timestamps(None)
for r in tab:
    if "convT" in r["kernel"]:
        print(f"  {r['kernel']}: {r['avg_ms'] * 1e3:.1f} us")
a = ts.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
hw = a[:, 15]
t = a[:, :15].astype(np.float64) / 100.0
t0 = t[:, 0].min()
nch = int(((t[0, 1:13] != 0).sum()))
start, first, loop_end, end = t[:, 0] - t0, t[:, 1] - t0, t[:, 13] - t0, t[:, 14] - t0
print(f"last stamped launch: {len(a)} workgroups, {nch} chunks, span {end.max():.1f} us")
pct = lambda v: f"p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}"
print("  start                        ", pct(start))
print("  start -> chunk 0 staged      ", pct(first - start))
if nch > 1:
    print("  chunk period                 ", pct(np.diff(t[:, 1:1 + nch], axis=1).ravel()))
print("  last chunk staged -> loop end", pct(loop_end - (t[:, nch] - t0)))
print("  epilogue (incl. drain)       ", pct(end - loop_end))
print("  workgroup life               ", pct(end - start))
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((hw >> 32) << 8)
ucu = np.unique(cu)
cnt = np.array([np.sum(cu == c) for c in ucu])
print(f"  CUs seen {len(ucu)}; workgroups per CU min {cnt.min()} mean {cnt.mean():.2f} max {cnt.max()}")
conc = [max(int(sum((start[cu == c] <= tt) & (end[cu == c] > tt))) for tt in np.linspace(0, end.max(), 80)) for c in ucu[:16]]
print("  max concurrent workgroups on a CU (16 CUs sampled):", max(conc), " mean over time on CU 0: %.2f" % np.mean([sum((start[cu == ucu[0]] <= tt) & (end[cu == ucu[0]] > tt)) for tt in np.linspace(0, end.max(), 200)]))
