"""CPU: host-side logic — C-ABI exports, config resolver, drop-in module key parity, sharding (gloo, world_size 2)."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from vocoder_amd import _lib, config, synthetic as syn
from vocoder_amd.sharding import shard_sizes, shard_slice

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "fishvoc.h")).read()
    declared = set(re.findall(r"FV_API\s+[\w\s\*]+?\b(fv_\w+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()      # loads without a GPU (no compute calls here)
    for name in declared:
        assert hasattr(L, name), name
    assert L.fv_abi_version() == _lib.FV_ABI_VERSION


def test_loading_the_library_brings_torch_in_first():
    """One HIP runtime per process: torch's wheel carries its own libamdhip64, and libfishvoc_hip.so must bind to that copy — `_lib.lib()` imports torch
    before it opens the library (loaded the other way round, the system runtime arrives as a second one and its first hipMalloc finds no device once torch
    has initialised it: `build()` followed by `smoke()` in one process).  Checked in a fresh interpreter."""
    import subprocess
    code = ("import sys; from vocoder_amd import _lib; assert 'torch' not in sys.modules; _lib.lib(); assert 'torch' in sys.modules; "
            "m = open('/proc/self/maps').read(); n = len({l.split()[-1] for l in m.splitlines() if 'libamdhip64' in l}); assert n == 1, n; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_config_struct_layout_matches_header():
    # int32 fields only, so sizeof is a good canary for drift between fishvoc.h and the ctypes mirror
    assert ctypes.sizeof(_lib.UpsamplerConfig) == 4 * (2 + 8 + 8 + 1 + 8 + 24 + 5 + 2)
    assert ctypes.sizeof(_lib.ConvNeXtConfig) == 4 * (2 + 8 + 8 + 1)
    assert ctypes.sizeof(_lib.IstftHeadConfig) == 20
    assert ctypes.sizeof(_lib.ConvDesc) == 40
    assert ctypes.sizeof(_lib.LogMelConfig) == 28
    assert ctypes.sizeof(_lib.RefineGANConfig) == 4 * (2 + 8 + 8 + 3)
    assert ctypes.sizeof(_lib.Config) == (8 + ctypes.sizeof(_lib.UpsamplerConfig) + ctypes.sizeof(_lib.ConvNeXtConfig) + 20 + 28 +
                                          ctypes.sizeof(_lib.RefineGANConfig))


def test_fv_create_validation_without_gpu():
    """Config validation (the reference's ctor asserts) happens before any device work."""
    from vocoder_amd.engine import upsampler_config
    L = _lib.lib()
    cfg = _lib.Config()
    cfg.abi_version = _lib.FV_ABI_VERSION
    cfg.model = _lib.FV_MODEL_HIFIGAN
    cfg.ups = upsampler_config(**dict(syn.HIFIGAN_V1_44K, hop_length=511))
    h = ctypes.c_void_p()
    assert L.fv_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"hop_length must be 512" in L.fv_last_error()
    cfg.ups = upsampler_config(**dict(syn.HIFIGAN_V1_44K, use_template=True, upsample_rates=[8, 8, 2, 2, 2, 1],
                                      upsample_kernel_sizes=[16, 16, 8, 2, 2, 1]))
    assert L.fv_create(ctypes.byref(cfg), ctypes.byref(h)) == -2   # stride_f0 = 1 before the last stage: length mismatch upstream
    cfg.abi_version = 99
    assert L.fv_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_config_compose_and_instantiate():
    gen, cfg = config.build_generator("hifigan", overrides={"num_mels": 80})
    g = cfg["model"]["generator"]
    assert g["hop_length"] == 512 and g["num_mels"] == 80 and g["use_template"] is False
    assert type(gen).__name__ == "HiFiGANGenerator"
    assert config._resolve_str("${eval: '${model.hop_length} * 32'}", {"model": {"hop_length": 512}}) == 16384
    v, _ = config.build_generator("vocos", "24000_256_1024")
    assert type(v.backbone).__name__ == "ConvNeXtEncoder" and v.head.n_fft == 1024
    with pytest.raises(AssertionError, match="hop_length must be 512"):
        config.build_generator("hifigan", overrides={"hop_length": 256})
    # every generator YAML of the reference that builds a network on this path: in-group ``defaults`` (vocos-huge = vocos + own keys)
    small, _ = config.build_generator("vocos-small", "24000_256_1024")
    assert len(small.backbone.stages) == 1 and len(small.backbone.stages[0]) == 8 and small.head.n_fft == 1024
    huge_cfg = config.compose_model("vocos-huge", "24000_256_1024")["model"]["generator"]
    assert huge_cfg["backbone"]["dims"] == [352, 704, 1408, 2816] and huge_cfg["head"]["dim"] == 2816
    assert huge_cfg["head"]["padding"] == "same" and huge_cfg["backbone"]["input_channels"] == 100 and "defaults" not in huge_cfg
    ff, _ = config.build_generator("firefly-gan-base", "44100_512_2048")
    assert type(ff.head).__name__ == "HiFiGANGenerator" and type(ff.backbone).__name__ == "ConvNeXtEncoder"


def test_reference_target_strings_resolve_to_dropins():
    cls = config.locate("fish_vocoder.modules.generators.hifigan.HiFiGANGenerator")
    assert cls.__module__ == "vocoder_amd.modules.generators.hifigan"
    assert config.locate("fish_vocoder.modules.encoders.convnext.ConvNeXtEncoder").__module__.startswith("vocoder_amd")


def test_dropin_state_dict_keys_and_strict_load():
    from vocoder_amd.modules.generators import BigVGANGenerator, HiFiGANGenerator
    cfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=20,
               upsample_initial_channel=64, use_template=False)
    gen = HiFiGANGenerator(**cfg)
    sd = syn.hifigan_state_dict(cfg, 3)
    assert list(gen.state_dict().keys()) == list(sd.keys())      # reference order, reference names
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    with pytest.raises(RuntimeError):
        gen.load_state_dict({k: torch.from_numpy(v) for k, v in list(sd.items())[1:]}, strict=True)
    tg = HiFiGANGenerator(**dict(cfg, use_template=True))      # the ctor default upstream: noise_convs.{i}.{weight,bias}
    assert list(tg.state_dict().keys()) == list(syn.hifigan_state_dict(dict(cfg, use_template=True), 3).keys())
    with pytest.raises(TypeError):
        tg.eval()(torch.zeros(1, 20, 4))
    with pytest.raises(AssertionError):
        HiFiGANGenerator(**dict(cfg, hop_length=15))
    # no CPU fallback: a CPU tensor must fail loudly, never compute
    with pytest.raises(RuntimeError, match="no CPU path"):
        gen.eval()(torch.zeros(1, 20, 4))
    assert gen.remove_parametrizations() is None
    big = BigVGANGenerator(**cfg)
    keys = set(big.state_dict().keys())
    bsd = syn.bigvgan_state_dict(cfg, 1)
    assert set(bsd) <= keys and all(k.endswith("filter") for k in keys - set(bsd))


def test_lightning_checkpoint_prefix_and_mel_preparation(tmp_path):
    from vocoder_amd.inference import InferenceModel, load_generator_state_dict, prepare_mel, write_wav
    sd = {"generator.conv_pre.bias": torch.zeros(4), "discriminators.x": torch.zeros(1)}
    assert list(load_generator_state_dict({"state_dict": sd})) == ["conv_pre.bias"]
    assert prepare_mel(torch.zeros(50, 80), 80).shape == (1, 80, 50)
    assert prepare_mel(torch.zeros(2, 80, 50), 80).shape == (2, 80, 50)
    write_wav(tmp_path / "a.wav", np.zeros((100, 1), np.float32), 44100)
    assert (tmp_path / "a.wav").stat().st_size == 44 + 200
    from vocoder_amd.inference import read_wav
    sig = np.stack([np.sin(np.arange(300) * 0.05), np.cos(np.arange(300) * 0.03)], 1).astype(np.float32) * 0.7
    write_wav(tmp_path / "s.wav", sig, 24000)
    back, sr = read_wav(tmp_path / "s.wav")
    assert sr == 24000 and back.shape == (2, 300) and np.abs(back.T - sig).max() <= 2.0 / 32767   # write scales by 32767, read by 1 / 32768 (librosa)
    with pytest.raises(NotImplementedError):
        InferenceModel(torch.nn.Identity())(torch.zeros(1, 1, 8))


class _HParams:   # a module-level non-tensor object, like the DictConfig / callbacks a Lightning .ckpt carries
    def __init__(self):
        self.lr = 1e-4


def test_lightning_checkpoint_with_non_tensor_objects_needs_explicit_trust(tmp_path):
    """The reference loads checkpoints with a plain torch.load (test.py:32); ours is weights-only by default and says how to
    opt in when a checkpoint holds pickled objects."""
    from vocoder_amd.inference import load_generator_state_dict
    sd = {"generator.conv_pre.bias": torch.arange(4.0), "discriminators.x": torch.zeros(1)}
    p = tmp_path / "lightning.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": _HParams(), "epoch": 3}, p)
    with pytest.raises(RuntimeError, match="trust-checkpoint"):
        load_generator_state_dict(p)
    got = load_generator_state_dict(p, trust_checkpoint=True)
    assert list(got) == ["conv_pre.bias"] and torch.equal(got["conv_pre.bias"], torch.arange(4.0))
    q = tmp_path / "weights.pt"            # tensors only: loads without the opt-in
    torch.save({"state_dict": sd}, q)
    assert list(load_generator_state_dict(q)) == ["conv_pre.bias"]
    # a missing or corrupt file is not a trust question: the original error comes through, with or without the opt-in
    for trust in (False, True):
        with pytest.raises(FileNotFoundError):
            load_generator_state_dict(tmp_path / "absent.ckpt", trust_checkpoint=trust)
    bad = tmp_path / "corrupt.ckpt"
    bad.write_bytes(p.read_bytes()[:200])
    with pytest.raises(Exception) as ei:
        load_generator_state_dict(bad, trust_checkpoint=True)
    assert "trust-checkpoint" not in str(ei.value)


def test_shard_slices_cover_batch_exactly():
    for batch, world in [(256, 8), (5, 8), (0, 2), (33, 4), (1, 1)]:
        sizes = shard_sizes(batch, world)
        assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1
        covered = [i for r in range(world) for i in range(batch)[shard_slice(batch, world, r)]]
        assert covered == list(range(batch))
    with pytest.raises(ValueError):
        shard_slice(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from vocoder_amd.sharding import broadcast_state_dict, gather_batch, scatter_batch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = dict(hop_length=4, upsample_rates=[2, 2], upsample_kernel_sizes=[4, 4], resblock_kernel_sizes=[3],
                   resblock_dilation_sizes=[[1, 3, 5]], num_mels=6, upsample_initial_channel=8, use_template=False)
        sd = syn.hifigan_state_dict(cfg, 5) if rank == 0 else None
        got = broadcast_state_dict(sd, src=0)
        ref = syn.hifigan_state_dict(cfg, 5)
        same = list(got) == list(ref) and all(np.array_equal(got[k].numpy(), ref[k]) for k in ref)
        batch = 5   # ragged: ranks get 3 and 2 clips
        full = torch.arange(batch * 6 * 7, dtype=torch.float32).reshape(batch, 6, 7) if rank == 0 else None
        mine = scatter_batch(full, batch, (6, 7), src=0)
        out = gather_batch(mine * 2.0, batch, dst=0)   # stand-in for the per-rank forward
        ok = same and mine.shape[0] == (3 if rank == 0 else 2)
        if rank == 0:
            ok = ok and torch.equal(out, full * 2.0)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_scatter_gather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _clipwise(x):
    """Stand-in for a batch-invariant forward on CPU (the engine needs a GPU): every output row is a function of its own clip alone,
    (B, C, T) -> (B, 1, 4 T) like a hop-4 generator."""
    y = torch.tanh(x.double().cumsum(2).mean(1, keepdim=True) * 0.37).float()
    return y.repeat_interleave(4, dim=2) + torch.arange(4 * x.shape[2], dtype=torch.float32)[None, None] * 1e-3


def _ragged_worker(rank, world, port, q):
    import torch.distributed as dist
    from vocoder_amd.sharding import gather_batch, scatter_batch, shard_sizes
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for batch in (37, 1, 2):           # 19 / 18 clips; one clip: rank 1's shard is EMPTY; one clip each
            C, T = 80, 11
            g = torch.Generator().manual_seed(100 + batch)
            full = torch.randn(batch, C, T, generator=g) if rank == 0 else None
            tmpl = torch.randn(batch, 1, 4 * T, generator=g) if rank == 0 else None      # a second tensor per clip (pitch template)
            mine = scatter_batch(full, batch, (C, T), src=0)
            mine_t = scatter_batch(tmpl, batch, (1, 4 * T), src=0)
            sizes = shard_sizes(batch, world)
            ok = ok and mine.shape == (sizes[rank], C, T) and mine_t.shape == (sizes[rank], 1, 4 * T)
            local = _clipwise(mine) + mine_t if sizes[rank] else torch.empty((0, 1, 4 * T))
            out = gather_batch(local, batch, dst=0)
            if rank == 0:
                ok = ok and out.shape == (batch, 1, 4 * T) and torch.equal(out, _clipwise(full) + tmpl)   # bit-identical to the global batch
            else:
                ok = ok and out is None
        q.put((rank, bool(ok), shard_sizes(37, world)))
    finally:
        dist.destroy_process_group()


def test_ragged_batch_invariant_scatter_gather_world2_gloo():
    """the ragged path of BASELINE config[4]'s plumbing — 37 clips over two ranks (19 / 18), a second per-clip tensor, an empty
    shard (1 clip over 2 ranks) — through scatter_batch / gather_batch with a clip-wise (batch-invariant) forward: the collected batch equals the
    global batch bit for bit.  The engine-side half (a ragged split bit-identical under fv_set_batch_invariant) is the GPU test
    test_ragged_shards_against_the_global_batch_and_the_batch_invariant_switch."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, [19, 18]), (1, True, [19, 18])]


def test_precision_option_is_validated_without_a_gpu():
    """The opt-in arithmetic selector of the drop-in modules / the binding (include/fishvoc.h fv_precision)."""
    from vocoder_amd import _lib
    from vocoder_amd.modules.generators.hifigan import HiFiGANGenerator
    from vocoder_amd import synthetic
    assert _lib.PRECISIONS == {"f32": 0, "f16x3": 1}
    gen = HiFiGANGenerator(**synthetic.HIFIGAN_V1_44K)
    assert gen.precision == "f32"
    gen.precision = "f16x3"
    assert gen.precision == "f16x3"
    with pytest.raises(ValueError):
        gen.precision = "bf16"
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "fishvoc.h")).read()
    assert "FV_PRECISION_F32 = 0" in hdr and "FV_PRECISION_F16X3 = 1" in hdr


def test_refinegan_dropin_has_the_reference_state_dict_keys():
    from vocoder_amd.modules.generators.refinegan import RefineGANGenerator
    cfg = dict(sampling_rate=16000, hop_length=16, downsample_rates=(2, 2, 2, 2), upsample_rates=(2, 2, 2, 2),
               leaky_relu_slope=0.2, num_mels=12, start_channels=4)
    gen = RefineGANGenerator(**cfg)
    sd = syn.refinegan_state_dict(cfg, seed=1)
    own = gen.state_dict()
    assert list(own.keys()) == list(sd.keys())          # same names in the same (reference) order
    for k, v in sd.items():
        assert tuple(own[k].shape) == v.shape, k
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert gen.noise_elems(2, 7) == sum(2 * c * t for c, t in syn.refinegan_stage_shapes(cfg, 7)) * 6
    with pytest.raises(AssertionError):
        RefineGANGenerator(**dict(cfg, hop_length=32))   # refinegan.py:202


def test_bench_self_launches_n_ranks_without_torchrun():
    """`python bench.py --gpus 2` (no torchrun environment) must re-execute itself as 2 ranks through torch.distributed.run
    and report n_gpus == the process group's world size (a plain `--gpus 8` silently ran one rank).  --dry-run
    exercises only the launcher, the process group (gloo here, RCCL on a GPU box) and the scatter/gather plumbing."""
    import json
    import subprocess
    import sys
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""    # the CPU path of the dry run, also on a GPU box
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "3"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints exactly one line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["scatter_gather_ok"] is True and j["dry_run"] is True
    # and a mismatch between --gpus and an existing torchrun environment is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=repo)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_cpu_clip_worker_protocol():
    """bench.py's cpu_baseline leg: oracle/cpu_clips.py workers (one process per core in use, whole clips each) answer the
    run / quit protocol and agree with the in-process oracle on the clips they were given."""
    import bench
    from oracle import oracle as orc
    pool = bench._ClipWorkers(2)
    try:
        dt, clips, samples = pool.run(2, 2, 3, seed=40)
        assert clips == 4 and samples == 4 * 3 * 512 and dt > 0
        dt, clips, samples = pool.run(1, 1, 2, seed=41)      # a subset of the pool, another shape
        assert clips == 1 and samples == 2 * 512
    finally:
        pool.close()
    assert all(p.poll() is not None for p in pool.procs)
