#!/usr/bin/env python
"""Headline benchmark: HiFiGAN-V1 44.1 kHz synthesis throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: re-executes itself as N ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one generator forward (mel -> waveform) over one batch of 32 synthetic one-second clips per GPU
(BASELINE config[1]: hifigan V1 44.1 kHz, 80-bin mel, T_mel = 86 -> 44 032 samples per clip), inputs already resident in
HBM, random-but-fixed weights of the real architecture (no network for checkpoints).  Multi-GPU = utterance sharding,
weak scaling: every rank runs its own 32 clips (config[4] = 256 clips over 8 GPUs); weights are fanned out from rank 0 by
one RCCL broadcast before the timed region; there is no collective on the data path (the reference's analogue is
Lightning's one-process-per-device launch, configs/trainer/default.yaml:6-9).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline        — the dominant kernel of the forward (by total time), timed live with hipEvents on the launch stream by
                    the engine's per-launch profiler (fv_profile_*): `achieved` / `frac` = matrix flops the kernel ISSUES per launch /
                    avg duration vs the fp32-MFMA peak (a hardware fraction, <= 1; bytes vs the HBM peak when that roof binds);
                    a Winograd kernel issues fewer products than the layer's direct sum — the algorithmic rate (what the reference
                    computes, over the same time) is carried as `algorithmic_tflops`, the ratio as `algorithmic_speedup`;
  with_collectives— the same K steps with config[4]'s "result collection over RCCL" inside the timed region: rank 0 owns the
                    global batch, every step = broadcast mels -> forward on the rank's shard -> all_gather waveforms; not the
                    headline value;
  other_configs   — BASELINE config[2] (BigVGAN-24k B=64) and config[3] (Vocos-24k B=128): ms/step and step-level roofline;
  cpu_baseline    — the CPU oracle (oracle/, a C port of the reference forward) on a bounded sample of the same workload,
                    rank 0 / N=1 only; host core count, CPU model and the thread count used are stated.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from vocoder_amd import _lib, synthetic as syn  # noqa: E402

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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BigVGAN-24k B=64 / Vocos-24k B=128 measurements appended as 'other_configs' (N=1 only)")
    ap.add_argument("--no-collectives", action="store_true", help="skip the scatter -> forward -> gather figure")
    ap.add_argument("--profile-json", default=None, help="also dump the per-kernel hipEvent table to this file")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the roofline section: three single-stream forwards with per-launch hipEvents (what `roofline` is computed "
                         "from) and a short line — the command to put under `rocprofv3 --kernel-trace --stats` so that its per-kernel "
                         "averages are taken under the same conditions as bench.py's own")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group / scatter-gather plumbing only, no engine and no timing claims "
                         "(backend gloo when there is no GPU: the CPU test of the self-launch path)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# launch: `python bench.py --gpus N` with N > 1 and no torchrun environment re-executes itself as N ranks
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_self_launch(a) -> None:
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def init_dist(dev):
    """One process per GPU over RCCL (backend "nccl"); gloo when there is no GPU (dry run on CPU).  A plain
    `python bench.py` (no torchrun environment) becomes a 1-rank group, so that the N = 1 line runs the same
    broadcast -> forward -> all_gather code on RCCL as the N = 8 one."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "WORLD_SIZE" not in os.environ:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dev.type == "cuda":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    return dist


def dry_run(a, world, rank, local_rank, real_stdout) -> None:
    from vocoder_amd.sharding import gather_batch, scatter_batch, shard_sizes
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    dist = init_dist(dev) if (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ) else None
    ranks = dist.get_world_size() if dist is not None else 1
    ok = True
    if dist is not None:
        gb = a.batch * ranks + 1   # ragged on purpose
        full = torch.arange(gb * 4 * 3, dtype=torch.float32).reshape(gb, 4, 3) if rank == 0 else None
        mine = scatter_batch(full, gb, (4, 3), src=0, device=dev)
        ok = mine.shape[0] == shard_sizes(gb, ranks)[rank]
        back = gather_batch(mine * 2.0, gb, dst=0)
        if rank == 0:
            ok = ok and bool(torch.equal(back.cpu(), full * 2.0))
        # the two bulk collectives of the timed with_collectives figure: broadcast in, all_gather out (equal shards)
        eq = torch.arange(ranks * 2 * 5, dtype=torch.float32, device=dev).reshape(ranks * 2, 5) if rank == 0 else torch.zeros((ranks * 2, 5), device=dev)
        dist.broadcast(eq, src=0)
        mine2 = eq[rank * 2:(rank + 1) * 2] + 1.0
        allw = torch.empty_like(eq)
        dist.all_gather_into_tensor(allw, mine2.contiguous())
        ok = ok and bool(torch.equal(allw.cpu(), torch.arange(ranks * 2 * 5, dtype=torch.float32).reshape(ranks * 2, 5) + 1.0))
        oks = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(oks, op=dist.ReduceOp.MIN)
        ok = bool(oks.item() == 1.0)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "backend": "nccl" if use_gpu else "gloo",
                          "scatter_gather_ok": bool(ok), "note": "plumbing only: no engine, nothing measured"}), file=real_stdout, flush=True)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------
def _cpu_quota() -> float | None:
    """CPUs' worth of time the container's cgroup allows (cpu.max / cfs quota), None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2: "<quota|max> <period>"
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class _ClipWorkers:
    """Pool of oracle/cpu_clips.py processes (one whole clip at a time each, serial convs inside a clip)."""

    def __init__(self, n: int):
        env = dict(os.environ, OMP_NUM_THREADS="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        self.procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "oracle", "cpu_clips.py")], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, text=True, env=env, cwd=REPO) for _ in range(n)]
        for p in self.procs:
            if p.stdout.readline().strip() != "ready":
                self.close()
                raise RuntimeError("cpu_clips worker did not start")

    def run(self, workers: int, clips_each: int, frames: int, seed: int) -> tuple[float, int, int]:
        """`workers` processes synthesise `clips_each` clips each from a common start time; returns (wall seconds from that
        start to the last worker's end, clips, samples)."""
        start_at = time.time() + 0.5
        for i, p in enumerate(self.procs[:workers]):
            p.stdin.write(f"run {clips_each} {frames} {seed + i} {start_at}\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in self.procs[:workers]]
        return max(r["t1"] for r in res) - start_at, sum(r["clips"] for r in res), sum(r["samples"] for r in res)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.write("quit\n")
                p.stdin.flush()
            except OSError:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()


def cpu_baseline(cfg, sd, clips: int, frames: int) -> dict:
    """Times the oracle (kind 'port': oracle/fv_oracle.c restates the reference forward; the reference itself is
    Python and does not travel to the GPU box) on one-second clips, clip-parallel over the host's cores: one worker process
    per core in use, whole clips per worker, the convs serial inside a clip (oracle/cpu_clips.py)."""
    from oracle import oracle as orc
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cpu_quota()
    # cores the process can actually run on at once: the affinity mask, capped by the container's CPU quota (the GPU box shows
    # 256 hardware threads to a container that is allowed 16 CPUs' worth of time: more workers than that only time-share them)
    avail = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    cands = sorted({n for n in (max(1, avail // 2), avail, min(affinity, 2 * avail)) if n >= 1})
    pool = _ClipWorkers(max(cands))
    try:
        # worker count: one clip per worker at half, all and twice the usable cores; best clips/s
        tried, best, best_rate = {}, 1, 0.0
        for n in cands:
            dt, c, _ = pool.run(n, 1, frames, seed=7000)
            tried[n] = round(dt, 3)
            if c / dt > best_rate:
                best, best_rate = n, c / dt
        # bounded sample: about 15 s of CPU work at the probed rate, whole clips per worker, at least `clips` clips
        each = max(1, int(round(15.0 * best_rate / best)), (clips + best - 1) // best)
        dt, n_clips, n_samples = pool.run(best, each, frames, seed=1234)
    finally:
        pool.close()
    # B = 1 (BASELINE config[0], the reference's own CPU-runnable case): one 1 s clip alone on the otherwise idle host, the convs
    # parallel over output channels inside the clip (16 OpenMP threads: the port's per-conv barriers stop paying beyond that
    # — round-2 probe on the 256-thread box: 8 / 16 / 32 / 64 threads -> 1.9 / 1.2 / 1.8 / 2.7 s per 8 clips); median of 5
    b1_threads = min(16, avail)
    orc.set_num_threads(b1_threads)
    mel1 = syn.synthetic_mel(1, cfg["num_mels"], frames, seed=1234)
    orc.hifigan_forward(sd, cfg, mel1[:, :, :8])
    lat = []
    for _ in range(5):
        t1 = time.perf_counter()
        y = orc.hifigan_forward(sd, cfg, mel1)
        lat.append(time.perf_counter() - t1)
    n_samp = y.shape[-1]
    return {"value": n_samples / dt, "unit": "samples/s", "cores": min(best, avail), "workers": best, "kind": "port",
            "host_cores": os.cpu_count(), "host_cores_usable": avail, "cgroup_cpu_quota": quota, "cpu_model": _cpu_model(),
            "worker_probe_s": tried, "parallelism": "one worker process per usable core, whole clips per process, serial convs inside a clip",
            "x_realtime": n_samples / dt / SAMPLE_RATE,
            "b1_clip_latency_ms": float(np.median(lat) * 1e3), "b1_x_realtime": n_samp / float(np.median(lat)) / SAMPLE_RATE,
            "b1_threads": b1_threads,
            "sample": f"{n_clips} x 1 s clips (T_mel={frames}) of the same HiFiGAN-V1-44k workload, {each} per worker, {dt:.2f} s wall "
                      f"on {best} clip-parallel worker processes (best clips/s of a one-clip-per-worker probe at {cands} workers) on "
                      f"{avail} usable cores ({affinity} hardware threads visible, cgroup CPU quota {quota}); b1_* = one 1 s clip alone with {b1_threads} OpenMP threads inside its convs, median of 5"}


# ------------------------------------------------------------------------------------------------------------------
# roofline
# ------------------------------------------------------------------------------------------------------------------
_WINO_PRODUCTS = {3: 4 / 6, 7: 10 / 14, 11: 16 / 22}   # matrix products per output pair: Winograd F(2,3) tap groups / direct sum
_WINO4_PRODUCTS = {3: 6 / 12, 7: 16 / 28, 11: 26 / 44}  # ... per four outputs: F(4,3) tap groups (conv_wino4_impl.h) / direct sum
_WINO44_PRODUCTS = {7: 13 / 28, 11: 20 / 44}            # ... F(4,4) tap groups (conv_wino44_impl.h)


def executed_flops(rec: dict) -> float:
    """MFMA flops a launch really issues: the algorithmic (direct-sum) count, except for the Winograd kernels (conv_wino_impl.h,
    pair_wino_impl.h), which compute the same outputs with 4 / 10 / 16 products per output pair instead of 6 / 14 / 22, and the F(4,3) ones
    (conv_wino4_impl.h): 6 / 16 / 26 per four outputs instead of 12 / 28 / 44; F(4,4) (conv_wino44_impl.h): 13 / 20 instead of 28 / 44."""
    k = rec["kernel"]
    for pre in ("conv_wino44<k=", "pair_wino44<k=", "conv_wino_lat44<k="):   # (round 5: the fused narrow pairs and the latency kernel at k = 7 / 11)
        if k.startswith(pre):
            return rec["flops_per_launch"] * _WINO44_PRODUCTS[int(k[len(pre):].split()[0])]
    if k.startswith("conv_wino4<k="):
        return rec["flops_per_launch"] * _WINO4_PRODUCTS[int(k[len("conv_wino4<k="):].split()[0])]
    for pre in ("conv_wino<k=", "pair_wino<k=", "conv_wino_lat<k="):
        if k.startswith(pre):
            return rec["flops_per_launch"] * _WINO_PRODUCTS[int(k[len(pre):].split()[0])]
    return rec["flops_per_launch"]


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
    traffic, traffic_src = None, None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            from pmc_traffic import bench_key, kernel_source_sha16
            tj = json.load(open(tpath))
            ent = tj.get(bench_key(top["kernel"]))
            traffic = ent["hbm_bytes_per_launch"] if ent else None
            if ent:
                # looked up, so it can go stale: the collection records a hash of the kernel sources, compared with this tree's
                same = tj.get("_src_sha16") == kernel_source_sha16()
                traffic_src = ("profiles/traffic.json (rocprofv3 PMC passes of " + str(tj.get("_build", "round-1 build r01f")) +
                               "; looked up by kernel name, not measured in this run; kernel sources " +
                               ("unchanged since that collection)" if same else "CHANGED since that collection: figure may be stale)"))
        except Exception:
            traffic = None
    if t_mfma >= t_hbm:
        # `achieved` / `frac` = flops the matrix pipe ISSUES per second against its dense peak (a hardware fraction, <= 1).  For a Winograd
        # kernel that is fewer than the layer's algorithmic flops: the direct sum the reference computes is carried next to it as
        # algorithmic_tflops, and algorithmic_speedup = algorithmic / issued products (22/16, 14/10, 6/4)
        ex = executed_flops(top)
        ex_tf = ex / t_s / 1e12
        out = {"bound": "mfma", "achieved": ex_tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": ex_tf / peak_tf,
               "issued_flops_per_launch": ex, "algorithmic_tflops": tf, "algorithmic_speedup": top["flops_per_launch"] / ex}
        if ex != top["flops_per_launch"]:
            out["note"] = ("achieved / frac count the matrix products the kernel issues (Winograd F(2,3) / F(4,3) / F(4,4) tap groups: issued_flops_per_launch); "
                           "algorithmic_tflops counts the layer's direct sum 2 C_in C_out k T B (flops_per_launch) over the same time and "
                           "may exceed the peak")
        if peak_tf != PEAK_MFMA_F32_TFLOPS:
            out["peak_note"] = "dense fp16 MFMA peak 2500 TFLOP/s / 3 products per MAC (f16x3 split)"
    else:
        out = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS}
    out.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": top["kernel"], "avg_ms": top["avg_ms"],
                "launches_per_step": top["launches"] // repeats,
                "flops_per_launch": top["flops_per_launch"], "bytes_per_launch": top["bytes_per_launch"],
                "share_of_step": top["total_ms"] / sum(r["total_ms"] for r in table)})
    return out


def step_roofline(table, repeats, ms_per_step, peak_tf=PEAK_MFMA_F32_TFLOPS) -> dict:
    """Whole step against the MFMA roof: `frac` = issued matrix flops / time / peak (<= 1); the algorithmic (direct-sum) flops of the same
    step are carried as algorithmic_tflops (they exceed the issued ones where Winograd kernels run)."""
    flops = sum(r["flops_per_launch"] * (r["launches"] // repeats) for r in table)
    ex = sum(executed_flops(r) * (r["launches"] // repeats) for r in table)
    ach = ex / (ms_per_step * 1e-3) / 1e12
    return {"flops_per_step": flops, "issued_flops_per_step": ex, "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf,
            "algorithmic_tflops": flops / (ms_per_step * 1e-3) / 1e12, "algorithmic_speedup": flops / ex}


def time_engine(eng, mel, out, steps, warmup, dev) -> float:
    for _ in range(warmup):
        eng(mel, out)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng(mel, out)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


MIXED_SHAPES = ((1, 86), (4, 200), (32, 86), (2, 431), (8, 47), (16, 120), (3, 301), (24, 64))   # (clips, mel frames)


def mixed_shapes(eng, cfg, dev, forwards: int = 200) -> dict:
    """`forwards` calls cycling eight (batch, frames) pairs, a fresh output tensor per call (none passed in), graph replay
    left at its default: throughput of a caller whose requests never repeat back to back."""
    mels = [torch.from_numpy(syn.synthetic_mel(b, cfg["num_mels"], t, seed=500 + i)).to(dev) for i, (b, t) in enumerate(MIXED_SHAPES)]
    for m in mels:   # first touch of every shape (workspace growth, lazy module loads) stays outside the timed region
        eng(m)
    torch.cuda.synchronize(dev)
    samples, ok = 0, True
    t0 = time.perf_counter()
    for i in range(forwards):
        y = eng(mels[i % len(mels)])
        samples += y.numel()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(y).all().item())
    return {"forwards": forwards, "shapes_clips_x_frames": [list(x) for x in MIXED_SHAPES], "ms_per_forward": dt / forwards * 1e3,
            "value": samples / dt, "unit": "samples/s", "x_realtime": samples / dt / SAMPLE_RATE, "output_finite": ok,
            "note": "fresh output tensor per call, shapes change every call (no hipGraph replay possible)"}


def other_configs(dev, steps, warmup) -> list[dict]:
    """BASELINE config[2] and config[3] on one MI355X: same timing method as the headline, step-level roofline."""
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    res = []

    def run(name, eng, mel, sr, workload):
        out = torch.empty((mel.shape[0], 1, eng.output_length(mel.shape[2])), dtype=torch.float32, device=dev)
        dt = time_engine(eng, mel, out, steps, warmup, dev)
        tab = eng.profile(mel, repeats=2)
        top = max(tab, key=lambda r: r["total_ms"])
        n = out.numel()
        r = {"config": workload, "model": name, "batch": int(mel.shape[0]), "t_mel": int(mel.shape[2]),
             "ms_per_step": dt * 1e3, "value": n / dt, "unit": "samples/s", "x_realtime": n / dt / sr,
             "output_finite": bool(torch.isfinite(out).all().item()),
             "roofline_step": step_roofline(tab, 2, dt * 1e3),
             "dominant_kernel": {"kernel": top["kernel"], "avg_ms": top["avg_ms"], "launches_per_step": top["launches"] // 2,
                                 "tflops": top["flops_per_launch"] / (top["avg_ms"] * 1e-3) / 1e12,
                                 "gbs": top["bytes_per_launch"] / (top["avg_ms"] * 1e-3) / 1e9,
                                 "share_of_step": top["total_ms"] / sum(x["total_ms"] for x in tab)},
             "serialized_kernel_ms": sum(x["total_ms"] for x in tab) / 2}
        res.append(r)
        eng.close()

    try:
        cfg = dict(syn.BIGVGAN_24K)
        eng = Engine(_lib.FV_MODEL_BIGVGAN, ups=upsampler_config(**cfg), state_dict=syn.bigvgan_state_dict(cfg, 0))
        run("bigvgan-24k", eng, torch.from_numpy(syn.synthetic_mel(64, 80, 94, 1)).to(dev), 24000,
            "BASELINE config[2]: bigvgan (snake + anti-alias), 24 kHz, batch=64 x 1 s, 1 MI355X")
    except Exception as exc:  # noqa: BLE001 - auxiliary figure: never cost the headline line
        res.append({"model": "bigvgan-24k", "error": f"{type(exc).__name__}: {exc}"})
    try:
        cfg = dict(syn.VOCOS_24K)
        eng = Engine(_lib.FV_MODEL_VOCOS, backbone=convnext_config(**cfg["backbone"]), head=istft_head_config(**cfg["head"]),
                     state_dict=syn.vocos_state_dict(cfg, 0))
        run("vocos-24k", eng, torch.from_numpy(syn.synthetic_mel(128, 80, 94, 2)).to(dev), 24000,
            "BASELINE config[3]: vocos ConvNeXt [3,3,27,3] + ISTFT head, 24 kHz, batch=128 x 1 s, 1 MI355X")
    except Exception as exc:  # noqa: BLE001
        res.append({"model": "vocos-24k", "error": f"{type(exc).__name__}: {exc}"})
    return res


def one_shot(dev) -> list[dict]:
    """What the reference's own caller pays (test.py:31-38,88-90: build the model, load a checkpoint, ONE forward per file — nothing is ever
    repeated, so hipGraph replay never engages): per model, through the drop-in module exactly as test.py drives it — ctor + load_state_dict +
    .to(device) (`module_ms`), engine creation from the module's device state dict (`engine_create_ms`: one flat D2H copy, weight-norm fold and
    fragment packing on all host cores, uploads), the first forward of a one-second clip (`first_forward_ms`: workspace allocation, every kernel
    cold), the second (the capture call) and the third (replayed).  The first model also pays the process's first touch of the code objects."""
    from vocoder_amd.modules.generators import BigVGANGenerator, HiFiGANGenerator, UnifyGenerator
    from vocoder_amd.modules.encoders import ConvNeXtEncoder
    from vocoder_amd.modules.generators.vocos import ISTFTHead

    def hifigan():
        cfg = dict(syn.HIFIGAN_V1_44K)
        return HiFiGANGenerator(**cfg), syn.hifigan_state_dict(cfg, 0), (1, 80, 86), True

    def bigvgan():
        cfg = dict(syn.BIGVGAN_24K)
        return BigVGANGenerator(**cfg), syn.bigvgan_state_dict(cfg, 0), (1, 80, 94), False   # (filter buffers are derived: not in the dict)

    def vocos():
        cfg = dict(syn.VOCOS_24K)
        return UnifyGenerator(ConvNeXtEncoder(**cfg["backbone"]), ISTFTHead(**cfg["head"])), syn.vocos_state_dict(cfg, 0), (1, 80, 94), True

    res = []
    for name, make in (("hifigan-v1-44k", hifigan), ("bigvgan-24k", bigvgan), ("vocos-24k", vocos)):
        try:
            t0 = time.perf_counter()
            gen, sd, shape, strict = make()
            gen.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=strict)
            gen = gen.eval().to(dev)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            parts = [gen] if hasattr(gen, "engine") else [gen.backbone, gen.head]   # (UnifyGenerator chains two engine modules)
            for m in parts:
                m.engine(dev)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            mel = torch.from_numpy(syn.synthetic_mel(*shape, seed=77)).to(dev)
            torch.cuda.synchronize(dev)
            fw = []
            for _ in range(3):
                t = time.perf_counter()
                y = gen(mel)
                torch.cuda.synchronize(dev)
                fw.append((time.perf_counter() - t) * 1e3)
            res.append({"model": name, "tensors": len(sd), "weight_mb": sum(int(np.asarray(v).size) for v in sd.values()) * 4 / 1e6,
                        "module_ms": (t1 - t0) * 1e3, "engine_create_ms": (t2 - t1) * 1e3, "first_forward_ms": fw[0],
                        "second_forward_capture_ms": fw[1], "third_forward_replay_ms": fw[2],
                        "create_plus_first_forward_ms": (t2 - t1) * 1e3 + fw[0], "output_finite": bool(torch.isfinite(y).all().item())})
            for m in parts:
                m.invalidate_engine()
            del gen, parts, y
        except Exception as exc:  # noqa: BLE001 - auxiliary figure: never cost the headline line
            res.append({"model": name, "error": f"{type(exc).__name__}: {exc}"})
    return res


def main():
    a = parse()
    maybe_self_launch(a)
    # stdout carries exactly ONE line, the JSON: native libraries write banners to fd 1 through their own stdio buffers (RCCL prints
    # its version block when the process group comes up, flushed at exit, i.e. AFTER the line) — everything else goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    try:
        _main(a, real_stdout)
    finally:
        real_stdout.flush()


def _main(a, real_stdout):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run:
        dry_run(a, world, rank, local_rank, real_stdout)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    from vocoder_amd.engine import Engine, upsampler_config
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, rccl_error = None, None
    t_pg = time.perf_counter()
    try:
        dist = init_dist(dev)                               # RCCL over xGMI; a 1-rank group at N = 1
    except Exception as exc:  # noqa: BLE001
        if world > 1:
            raise
        rccl_error = f"{type(exc).__name__}: {exc}"         # N = 1 keeps its headline line without the process group
    ranks = dist.get_world_size() if dist is not None else 1
    t_pg = time.perf_counter() - t_pg

    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=0) if rank == 0 else None
    t_bc = None
    if dist is not None:
        from vocoder_amd.sharding import broadcast_state_dict
        dist.barrier()                                           # (rank 0 alone builds the state dict: its time is not the broadcast's)
        t_bc = time.perf_counter()
        sd_eng = broadcast_state_dict(sd, src=0, device=dev)    # one-time weight fan-out (56 MB) over RCCL
        torch.cuda.synchronize(dev)
        t_bc = time.perf_counter() - t_bc
    else:
        sd_eng = sd
    eng = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd_eng, precision=a.precision)

    B, T = a.batch, a.frames
    mel = torch.from_numpy(syn.synthetic_mel(B, cfg["num_mels"], T, seed=1234 + rank)).to(dev)
    out = torch.empty((B, 1, eng.output_length(T)), dtype=torch.float32, device=dev)
    samples_per_step = B * eng.output_length(T)
    if a.roofline_only:
        eng.profile(mel, repeats=1)   # warm-up in the same mode (single stream, eager): every launch of the trace is comparable
        table = eng.profile(mel, repeats=3)
        if a.profile_json:
            json.dump(table, open(a.profile_json, "w"), indent=1)
        if rank == 0:
            print(json.dumps({"roofline_only": True, "roofline": roofline_from_profile(table, 3),
                              "serialized_kernel_ms": sum(r["total_ms"] for r in table) / 3}), file=real_stdout, flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(a.warmup):
        eng(mel, out)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        eng(mel, out)
    ev1.record()
    fence()
    local = time.perf_counter() - t0
    elapsed = max_over_ranks(local)
    # Self-diagnosis of the N > 1 line: every rank's own step time by its own GPU clock (hipEvents around its K steps, before
    # the closing barrier) and by the host clock, the one-time weight fan-out, the process-group start-up and the environment the transport depends
    # on — so that the first 8-GPU curve the driver measures explains itself (a slow rank, a slow link, a missing IPC mode).  No efficiency figure.
    mine = torch.tensor([ev0.elapsed_time(ev1) / a.steps, local / a.steps * 1e3, (t_bc or 0.0) * 1e3, t_pg * 1e3], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(ranks)] if dist is not None else [mine]
    if dist is not None:
        dist.all_gather(per_rank, mine)
    per_rank = [[float(v) for v in t.tolist()] for t in per_rank]

    ok = bool(torch.isfinite(out).all().item()) and float(out.abs().max().item()) <= 1.0

    # ---- config[4]'s "result collection over RCCL", timed: rank 0 owns the global batch -----------------------------
    # Two bulk collectives per step (the forms RCCL is exercised with everywhere: no point-to-point ordering to get wrong):
    # the mels fan out with one broadcast from rank 0 (every rank slices its shard), the waveforms come back with one
    # all_gather.  sharding.scatter_batch / gather_batch (ragged shards, point-to-point) are the library form of the same.
    coll = None
    if not a.no_collectives:
        # Everything that can fail on one rank alone (allocations, the first forward on the shard) happens BEFORE the loop of
        # collectives, and the ranks agree on one success flag before any of them enters it: a rank that left the loop on an
        # exception would leave the others blocked inside RCCL until the watchdog fires.
        from vocoder_amd.sharding import shard_slice
        gb = B * ranks
        L = eng.output_length(T)
        full = whole = None
        err = None
        try:
            full = torch.empty((gb, cfg["num_mels"], T), dtype=torch.float32, device=dev)
            if rank == 0:
                full.copy_(torch.cat([torch.from_numpy(syn.synthetic_mel(B, cfg["num_mels"], T, seed=1234 + r)) for r in range(ranks)]))
            whole = torch.empty((gb, 1, L), dtype=torch.float32, device=dev)
            eng(full[shard_slice(gb, ranks, rank)], out)
            torch.cuda.synchronize(dev)
        except Exception as exc:  # noqa: BLE001 - auxiliary figure: never cost the headline line
            err = f"{type(exc).__name__}: {exc}"
        all_ok = err is None
        if dist is not None:
            flag = torch.tensor([1.0 if all_ok else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            all_ok = bool(flag.item() == 1.0)
        if all_ok:
            def coll_step():
                if dist is None:
                    eng(full, whole)
                    return
                dist.broadcast(full, src=0)                      # 1-rank group at N = 1: the same RCCL calls
                eng(full[shard_slice(gb, ranks, rank)], out)
                dist.all_gather_into_tensor(whole, out)

            for _ in range(max(1, a.warmup)):
                coll_step()
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                coll_step()
            fence()
            dtc = max_over_ranks(time.perf_counter() - t0)
            if rank == 0:
                # the collected batch against the plain forward of rank 0's own shard (same engine, same clips): bit for bit
                eng(mel, out)
                same = bool(torch.equal(whole[:B], out))
                coll = {"ms_per_step": dtc / a.steps * 1e3, "value": gb * L * a.steps / dtc, "unit": "samples/s",
                        "x_realtime": gb * L * a.steps / dtc / SAMPLE_RATE, "global_batch": gb,
                        "backend": "rccl" if dist is not None else "none (process group unavailable)",
                        "broadcast_bytes_per_step": gb * cfg["num_mels"] * T * 4 if dist is not None else 0,
                        "all_gather_bytes_per_step": gb * L * 4 if dist is not None else 0,
                        "collected_shape": list(whole.shape), "collected_finite": bool(torch.isfinite(whole).all().item()),
                        "rank0_shard_equals_direct_forward": same,
                        "note": "each step: rank 0 broadcasts the global mel batch over RCCL, every rank runs its shard, one "
                                "all_gather returns all waveforms (at N=1 a 1-rank RCCL group: the collectives run, on one GPU)"}
        elif rank == 0:
            coll = {"error": err or "another rank failed before the collective loop"}
        del full, whole

    result = None
    if rank == 0:
        # B=1 clip latency (p50) — the second half of BASELINE's metric: hipGraph replay (same buffers every call) and
        # eager (replay off: what a caller with ever-changing shapes / buffers gets)
        mel1 = mel[:1].contiguous()

        def p_lat(n=30):
            for _ in range(5):
                eng(mel1)
            torch.cuda.synchronize(dev)
            lat = []
            for _ in range(n):
                t1 = time.perf_counter()
                eng(mel1)
                torch.cuda.synchronize(dev)
                lat.append((time.perf_counter() - t1) * 1e3)
            return lat
        lat = p_lat()
        eng.set_batch_invariant(True)    # what the bit-reproducible mode costs a single clip
        lat_inv = p_lat()
        eng.set_batch_invariant(False)
        p_lat(5)
        eng.set_graph_replay(False)
        lat_eager = p_lat()
        # what a server sees: the B = 32 step with graph replay off, and a stream of forwards whose shapes keep changing and
        # whose outputs are fresh tensors every call (the engine keys its captured graph on pointers and shapes, so nothing
        # below can be replayed; workspace re-use across shapes is the engine's own)
        eager_step = time_engine(eng, mel, out, max(5, a.steps // 2), 2, dev)
        eng.set_graph_replay(True)
        mixed = mixed_shapes(eng, cfg, dev)
        repeats = 3
        table = eng.profile(mel, repeats=repeats)
        if a.profile_json:
            os.makedirs(os.path.dirname(os.path.abspath(a.profile_json)), exist_ok=True)
            json.dump(table, open(a.profile_json, "w"), indent=1)
        value = world * samples_per_step * a.steps / elapsed
        result = {
            "metric": "audio_samples_per_sec", "value": value, "unit": "samples/s",
            "n_gpus": world, "rccl_ranks": ranks if dist is not None else 0, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "f32" else "f32 via f16x3 split (fp16 operand planes, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "hifigan V1 44.1 kHz, 80-bin mel, batch=32x1 s synthetic mel per MI355X (BASELINE config[1]; "
                                   "config[4] at 8 GPUs)",
                       "clips_per_gpu": B, "t_mel": T, "samples_per_clip": eng.output_length(T),
                       "global_batch": B * world, "parallelism": f"utterance-shard x{world}"},
            "x_realtime": value / SAMPLE_RATE, "x_realtime_per_gpu": value / SAMPLE_RATE / world,
            "p50_clip_latency_ms": float(np.percentile(lat, 50)), "p90_clip_latency_ms": float(np.percentile(lat, 90)),
            "p50_clip_latency_batch_invariant_ms": float(np.percentile(lat_inv, 50)),
            "p50_clip_latency_eager_ms": float(np.percentile(lat_eager, 50)),
            "p90_clip_latency_eager_ms": float(np.percentile(lat_eager, 90)),
            "ms_per_step_eager": eager_step * 1e3,
            "mixed_shapes": mixed,
            "output_finite": ok,
            "roofline": roofline_from_profile(table, repeats),
        }
        gpu_ms = [r[0] for r in per_rank]
        result["multi_gpu"] = {
            "ranks": ranks, "per_rank_ms_per_step_gpu_clock": gpu_ms, "per_rank_ms_per_step_host_clock": [r[1] for r in per_rank],
            "ms_per_step_min": min(gpu_ms), "ms_per_step_max": max(gpu_ms), "slowest_rank": int(np.argmax(gpu_ms)),
            "weight_broadcast_ms_per_rank": [r[2] for r in per_rank] if dist is not None else None,
            "weight_broadcast_bytes": int(sum(int(np.asarray(v).size) for v in sd.values()) * 4),
            "process_group_init_ms_per_rank": [r[3] for r in per_rank],
            "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "NCCL_P2P_DISABLE", "RCCL_MSCCL_ENABLE",
                                                   "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "OMP_NUM_THREADS")},
            "device": torch.cuda.get_device_name(dev), "visible_devices": torch.cuda.device_count(),
            "note": "per-rank figures are diagnostics for reading the scaling curve; `value` / `ms_per_step` above are the max-over-ranks contract"}
        if rccl_error:
            result["rccl_error"] = rccl_error
        # whole-step view next to the dominant-kernel one: algorithmic flops of the forward (55.97 GFLOP per
        # one-second clip at T_mel = 86; summed here from the profiler's per-launch figures) over the measured step time
        peak_tf = PEAK_F16X3_TFLOPS if a.precision == "f16x3" else PEAK_MFMA_F32_TFLOPS
        result["roofline_step"] = step_roofline(table, repeats, elapsed / a.steps * 1e3, peak_tf)
        result["roofline_step"]["note"] = "all kernels of one forward, measured step time (branch streams + graph replay)"
        result["with_collectives"] = coll
        if world == 1 and a.precision == "f32" and not a.no_alt_precision:
            # the opt-in f16x3 mode on the same batch: throughput and its deviation from the exact-fp32 output above.
            # Auxiliary: a failure here must not cost the headline line.
            try:
                eng(mel, out)
                ref_out = out.clone()
                eng2 = Engine(_lib.FV_MODEL_HIFIGAN, ups=upsampler_config(**cfg), state_dict=sd_eng, precision="f16x3")
                out2 = torch.empty_like(out)
                dt2 = time_engine(eng2, mel, out2, a.steps, a.warmup, dev)
                v2 = samples_per_step / dt2
                result["alt_precision"] = {
                    "precision": "f16x3", "value": v2, "unit": "samples/s", "ms_per_step": dt2 * 1e3,
                    "x_realtime": v2 / SAMPLE_RATE, "max_abs_diff_vs_f32_output": float((out2 - ref_out).abs().max().item()),
                    "note": "opt-in (Engine(precision='f16x3')); not the headline value"}
                eng2.close()
                del ref_out, out2
            except Exception as exc:  # noqa: BLE001
                result["alt_precision"] = {"precision": "f16x3", "error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and a.precision == "f32" and not a.no_other_configs:
            eng.close()
            del out
            result["other_configs"] = other_configs(dev, max(5, a.steps // 2), a.warmup)
        if world == 1 and a.precision == "f32" and not a.no_other_configs:
            result["one_shot"] = one_shot(dev)
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
        print(json.dumps(result), file=real_stdout, flush=True)


if __name__ == "__main__":
    main()
