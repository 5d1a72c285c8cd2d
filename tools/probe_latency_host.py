#!/usr/bin/env python
"""Where a single clip's p50 goes between the host and the GPU: the bench's loop (default stream -> the engine's side stream), the same on a caller-owned
stream, the host time of the call alone, and the GPU span between two events.   python tools/probe_latency_host.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vocoder_amd import _lib, synthetic as syn
from vocoder_amd.engine import Engine, upsampler_config
cfg = dict(syn.HIFIGAN_V1_44K); sd = syn.hifigan_state_dict(cfg, seed=0)
eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd)
mel = torch.from_numpy(syn.synthetic_mel(1, 80, 86, seed=1)).cuda()
out = torch.empty((1, 1, eng.output_length(86)), device="cuda")


def p50(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    lat = []
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); lat.append((time.perf_counter() - t) * 1e3)
    return float(np.percentile(lat, 50))


def host_only(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    lat = []
    for _ in range(n):
        t = time.perf_counter(); fn(); lat.append((time.perf_counter() - t) * 1e3); torch.cuda.synchronize()
    return float(np.percentile(lat, 50))


def gpu_span(fn, stream=None, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    sp = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize(); sp.append(e0.elapsed_time(e1))
    return float(np.percentile(sp, 50))


for rnd in range(3):
    print(f"round {rnd}: default stream   p50 {p50(lambda: eng(mel)):.3f} ms  (out= given: {p50(lambda: eng(mel, out)):.3f})  host call {host_only(lambda: eng(mel, out)):.3f}  GPU span {gpu_span(lambda: eng(mel, out)):.3f}")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        print(f"round {rnd}: caller's stream  p50 {p50(lambda: eng(mel)):.3f} ms  (out= given: {p50(lambda: eng(mel, out)):.3f})  host call {host_only(lambda: eng(mel, out)):.3f}  GPU span {gpu_span(lambda: eng(mel, out), s):.3f}")
