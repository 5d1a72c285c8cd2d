#!/usr/bin/env python
"""Headline benchmark: HiFiGAN-V1 44.1 kHz synthesis throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one generator forward (mel -> waveform) over one batch of 32 synthetic one-second clips per GPU
(BASELINE config[1]: hifigan V1 44.1 kHz, 80-bin mel, T_mel = 86 -> 44 032 samples per clip), inputs already resident in
HBM, random-but-fixed weights of the real architecture (no network for checkpoints).  Multi-GPU = utterance sharding,
weak scaling: every rank runs its own 32 clips (config[4] = 256 clips over 8 GPUs); weights are fanned out from rank 0 by
one RCCL broadcast before the timed region; there is no collective on the data path.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      — the dominant kernel of the forward (by total time), timed live with hipEvents on the launch stream by
                  the engine's per-launch profiler (fv_profile_*), algorithmic flops per launch / avg duration vs the
                  fp32-MFMA peak (or bytes vs HBM peak when that is the binding roof);
  cpu_baseline  — the CPU oracle (oracle/, a C port of the reference forward) on a bounded sample of the same workload,
                  all host cores, rank 0 / N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from vocoder_amd import _lib, synthetic as syn  # noqa: E402
from vocoder_amd.engine import Engine, upsampler_config  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix peak (dense)
PEAK_MFMA_F16_TFLOPS = 2500.0  # same guide: dense fp16/bf16 matrix peak; the f16x3 mode spends 3 fp16 products per MAC,
PEAK_F16X3_TFLOPS = PEAK_MFMA_F16_TFLOPS / 3.0   # so its algorithmic-flop ceiling is a third of that
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak
SAMPLE_RATE = 44100
BATCH_PER_GPU = 32
T_MEL = 86


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="clips per GPU (default 32 = BASELINE config)")
    ap.add_argument("--frames", type=int, default=T_MEL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=96, help="clips in the bounded CPU-oracle sample")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"],
                    help="arithmetic of the MFMA-bound convs: f32 = exact fp32 MFMA (the reference's arithmetic, headline); "
                         "f16x3 = opt-in split-fp16 MFMA with fp32-class accuracy")
    ap.add_argument("--no-alt-precision", action="store_true",
                    help="skip the extra f16x3 measurement that a default (f32) single-GPU run appends as 'alt_precision'")
    ap.add_argument("--profile-json", default=None, help="also dump the per-kernel hipEvent table to this file")
    return ap.parse_args()


def cpu_baseline(cfg, sd, clips: int, frames: int) -> dict:
    """Times the oracle (kind 'port': oracle/fv_oracle.c restates the reference forward; the reference itself is
    Python and does not travel to the GPU box) on `clips` one-second clips with all host cores."""
    from oracle import oracle as orc
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    mel = syn.synthetic_mel(clips, cfg["num_mels"], frames, seed=1234)
    orc.hifigan_forward(sd, cfg, mel[:1, :, :8])  # page in / build
    # the OpenMP port parallelises over (clip, out-channel) rows; on a many-core host the best thread count is well
    # below the core count (barrier cost per conv), so pick it on a short probe and report the count actually used
    best, best_dt = 1, float("inf")
    for n in (8, 16, 32, 64, 128, 256):
        if n > avail:
            break
        orc.set_num_threads(n)
        t0 = time.perf_counter()
        orc.hifigan_forward(sd, cfg, mel[:1, :, :16])
        dt = time.perf_counter() - t0
        if dt < best_dt:
            best, best_dt = n, dt
    orc.set_num_threads(best)
    t0 = time.perf_counter()
    y = orc.hifigan_forward(sd, cfg, mel)
    dt = time.perf_counter() - t0
    return {"value": y.shape[0] * y.shape[-1] / dt, "unit": "samples/s", "cores": orc.num_threads(), "kind": "port",
            "x_realtime": y.shape[0] * y.shape[-1] / dt / SAMPLE_RATE,
            "sample": f"{clips} x 1 s clips (T_mel={frames}) of the same HiFiGAN-V1-44k workload, {dt:.2f} s of CPU work"}


def roofline_from_profile(table: list[dict], repeats: int) -> dict:
    """Dominant kernel = largest total time in the forward.  `achieved` = algorithmic flops (or bytes) per launch divided
    by its average hipEvent duration."""
    top = max(table, key=lambda r: r["total_ms"])
    t_s = top["avg_ms"] * 1e-3
    tf = top["flops_per_launch"] / t_s / 1e12
    gbs = top["bytes_per_launch"] / t_s / 1e9
    peak_tf = PEAK_F16X3_TFLOPS if top["kernel"].startswith(("conv_f16x3", "pair_f16x3")) else PEAK_MFMA_F32_TFLOPS
    t_mfma = top["flops_per_launch"] / (peak_tf * 1e12)
    t_hbm = top["bytes_per_launch"] / (PEAK_HBM_GBS * 1e9)
    # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.py);
    # PMC counters cannot be read from inside this process, so this is looked up, not measured live
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            from pmc_traffic import bench_key
            ent = json.load(open(tpath)).get(bench_key(top["kernel"]))
            traffic = ent["hbm_bytes_per_launch"] if ent else None
        except Exception:
            traffic = None
    if t_mfma >= t_hbm:
        out = {"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf}
        if peak_tf != PEAK_MFMA_F32_TFLOPS:
            out["peak_note"] = "dense fp16 MFMA peak 2500 TFLOP/s / 3 products per MAC (f16x3 split)"
    else:
        out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    out.update({"traffic": traffic, "kernel": top["kernel"], "avg_ms": top["avg_ms"],
                "launches_per_step": top["launches"] // repeats,
                "flops_per_launch": top["flops_per_launch"], "bytes_per_launch": top["bytes_per_launch"],
                "share_of_step": top["total_ms"] / sum(r["total_ms"] for r in table)})
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # launched by torch.distributed.run: one process per GPU
        import torch.distributed as dist  # noqa: PLC0415
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=0) if rank == 0 else None
    if dist is not None:
        from vocoder_amd.sharding import broadcast_state_dict
        sd_t = broadcast_state_dict(sd, src=0, device=dev)    # one-time weight fan-out (56 MB) over RCCL
        sd_eng = sd_t
    else:
        sd_eng = sd
    eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd_eng, precision=a.precision)

    B, T = a.batch, a.frames
    mel = torch.from_numpy(syn.synthetic_mel(B, cfg["num_mels"], T, seed=1234 + rank)).to(dev)
    out = torch.empty((B, 1, eng.output_length(T)), dtype=torch.float32, device=dev)
    samples_per_step = B * eng.output_length(T)

    for _ in range(a.warmup):
        eng(mel, out)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng(mel, out)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = bool(torch.isfinite(out).all().item()) and float(out.abs().max().item()) <= 1.0

    result = None
    if rank == 0:
        # B=1 clip latency (p50) — the second half of BASELINE's metric
        mel1 = mel[:1].contiguous()
        for _ in range(5):
            eng(mel1)
        torch.cuda.synchronize(dev)
        lat = []
        for _ in range(30):
            t1 = time.perf_counter()
            eng(mel1)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t1) * 1e3)
        repeats = 3
        table = eng.profile(mel, repeats=repeats)
        if a.profile_json:
            os.makedirs(os.path.dirname(os.path.abspath(a.profile_json)), exist_ok=True)
            json.dump(table, open(a.profile_json, "w"), indent=1)
        value = world * samples_per_step * a.steps / elapsed
        result = {
            "metric": "audio_samples_per_sec", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "f32" else "f32 via f16x3 split (fp16 operand planes, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "hifigan V1 44.1 kHz, 80-bin mel, batch=32x1 s synthetic mel per MI355X (BASELINE config[1]; "
                                   "config[4] at 8 GPUs)",
                       "clips_per_gpu": B, "t_mel": T, "samples_per_clip": eng.output_length(T),
                       "global_batch": B * world, "parallelism": f"utterance-shard x{world}"},
            "x_realtime": value / SAMPLE_RATE, "x_realtime_per_gpu": value / SAMPLE_RATE / world,
            "p50_clip_latency_ms": float(np.percentile(lat, 50)), "p90_clip_latency_ms": float(np.percentile(lat, 90)),
            "output_finite": ok,
            "roofline": roofline_from_profile(table, repeats),
        }
        # whole-step view next to the dominant-kernel one: algorithmic flops of the forward (55.97 GFLOP per
        # one-second clip at T_mel = 86; summed here from the profiler's per-launch figures) over the measured step time
        step_flops = sum(r["flops_per_launch"] * (r["launches"] // repeats) for r in table)
        peak_tf = PEAK_F16X3_TFLOPS if a.precision == "f16x3" else PEAK_MFMA_F32_TFLOPS
        ach = step_flops / (elapsed / a.steps) / 1e12
        result["roofline_step"] = {"flops_per_step": step_flops, "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s",
                                   "frac": ach / peak_tf,
                                   "note": "all kernels of one forward, measured step time (branch streams + graph replay)"}
        if world == 1 and a.precision == "f32" and not a.no_alt_precision:
            # the opt-in f16x3 mode on the same batch: throughput and its deviation from the exact-fp32 output above.
            # Auxiliary: a failure here must not cost the headline line.
            try:
                ref_out = out.clone()
                eng2 = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd_eng, precision="f16x3")
                out2 = torch.empty_like(out)
                for _ in range(a.warmup):
                    eng2(mel, out2)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    eng2(mel, out2)
                torch.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t1
                v2 = samples_per_step * a.steps / dt2
                result["alt_precision"] = {
                    "precision": "f16x3", "value": v2, "unit": "samples/s", "ms_per_step": dt2 / a.steps * 1e3,
                    "x_realtime": v2 / SAMPLE_RATE, "max_abs_diff_vs_f32_output": float((out2 - ref_out).abs().max().item()),
                    "note": "opt-in (Engine(precision='f16x3')); not the headline value"}
                eng2.close()
            except Exception as exc:  # noqa: BLE001
                result["alt_precision"] = {"precision": "f16x3", "error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, sd, a.cpu_clips, T)
            except Exception as exc:  # noqa: BLE001 - the oracle library may be missing on a box: report, keep the GPU line
                result["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        else:
            result["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
