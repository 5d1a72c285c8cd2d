"""GPU: the drop-in nn.Module classes (what a fish_vocoder user touches) end to end: Hydra-style config ->
instantiate -> load_state_dict(strict) -> .eval().to('cuda') -> forward(mel) -> waveform, against the reference goldens."""
import numpy as np
import pytest

from conftest import load_golden
from vocoder_amd import config, synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TOL = 1e-4


def _t(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_hifigan_module_from_config_matches_reference_golden():
    g = load_golden("hifigan_v1_t12.npz")
    gen, cfg = config.build_generator("hifigan", overrides={"num_mels": 80})
    gen.load_state_dict(_t(syn.hifigan_state_dict(g["cfg"], g["seed"])), strict=True)
    gen = gen.eval().to("cuda:0")
    y = gen(torch.from_numpy(g["mel"]).cuda())
    assert y.shape == (1, 1, 12 * 512) and y.device.type == "cuda"
    assert np.abs(y.cpu().numpy() - g["out"]).max() <= TOL
    # GANModel.forward contract (gan.py:282-288)
    from vocoder_amd.inference import InferenceModel
    out, zero = InferenceModel(gen)(None, None, input_spec=torch.from_numpy(g["mel"]).cuda())
    assert zero == 0 and torch.equal(out, y)
    # reloading other weights rebuilds the engine
    gen.load_state_dict(_t(syn.hifigan_state_dict(g["cfg"], g["seed"] + 1)), strict=True)
    y2 = gen(torch.from_numpy(g["mel"]).cuda())
    assert not torch.allclose(y, y2)
    with pytest.raises(RuntimeError, match="inference-only"):
        gen.train()(torch.from_numpy(g["mel"]).cuda())


def test_bigvgan_and_vocos_and_firefly_modules():
    from vocoder_amd.modules.encoders import ConvNeXtEncoder
    from vocoder_amd.modules.generators import BigVGANGenerator, HiFiGANGenerator, ISTFTHead, UnifyGenerator
    g = load_golden("bigvgan_tiny.npz")
    m = BigVGANGenerator(**g["cfg"])
    inc = m.load_state_dict(_t(syn.bigvgan_state_dict(g["cfg"], g["seed"])), strict=False)
    assert not inc.unexpected_keys and all(k.endswith("filter") for k in inc.missing_keys)
    y = m.eval().cuda()(torch.from_numpy(g["mel"]).cuda())
    assert np.abs(y.cpu().numpy() - g["out"]).max() <= TOL

    g = load_golden("vocos_tiny.npz")
    u = UnifyGenerator(ConvNeXtEncoder(**g["cfg"]["backbone"]), ISTFTHead(**g["cfg"]["head"]))
    u.load_state_dict(_t(syn.vocos_state_dict(g["cfg"], g["seed"])), strict=True)
    y = u.eval().cuda()(torch.from_numpy(g["mel"]).cuda())
    assert y.shape == g["out"].shape
    assert np.abs(y.cpu().numpy() - g["out"]).max() <= TOL * float(np.abs(g["out"]).max())   # relative to the capture's peak (0.06)

    # Firefly-GAN composition (f3): ConvNeXt backbone -> HiFiGAN head with k=13 pre/post convs, vs the oracle
    from oracle import oracle as orc
    cfg = dict(backbone=dict(input_channels=20, depths=[1, 1], dims=[32, 64], kernel_size=7),
               head=dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
                         resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=64,
                         upsample_initial_channel=64, use_template=False, pre_conv_kernel_size=13,
                         post_conv_kernel_size=13))
    sd = syn.firefly_state_dict(cfg, 4)
    f = UnifyGenerator(ConvNeXtEncoder(**cfg["backbone"]), HiFiGANGenerator(**cfg["head"]))
    f.load_state_dict(_t(sd), strict=True)
    mel = syn.synthetic_mel(2, 20, 9, 3)
    y = f.eval().cuda()(torch.from_numpy(mel).cuda()).cpu().numpy()
    assert np.abs(y - orc.firefly_forward(sd, cfg, mel)).max() <= TOL


def test_fused_firefly_and_vocos_engine_kinds_match_module_chain():
    """FV_MODEL_VOCOS / FV_MODEL_FIREFLY run backbone+head inside one fv_forward; must equal the two-engine chain."""
    from vocoder_amd import _lib
    from vocoder_amd.engine import Engine, convnext_config, istft_head_config, upsampler_config
    cfg = dict(backbone=dict(input_channels=20, depths=[1, 1], dims=[32, 64], kernel_size=7),
               head=dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
                         resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=64,
                         upsample_initial_channel=64, use_template=False, pre_conv_kernel_size=13,
                         post_conv_kernel_size=13))
    sd = syn.firefly_state_dict(cfg, 4)
    from oracle import oracle as orc
    mel = syn.synthetic_mel(2, 20, 9, 3)
    eng = Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**cfg["backbone"]), ups=upsampler_config(**cfg["head"]),
                 state_dict=sd)
    y = eng(torch.from_numpy(mel).cuda()).cpu().numpy()
    assert np.abs(y - orc.firefly_forward(sd, cfg, mel)).max() <= TOL


def test_logmel_frontend_module_and_wav_to_wav():
    """f1: wave -> log-mel on the GPU (reference LogMelSpectrogram golden), then mel -> HiFiGAN without leaving the device."""
    import json
    from vocoder_amd.data.transforms import LogMelSpectrogram
    from oracle import oracle as orc
    z = load_golden("logmel.npz")
    for tag in "ab":
        cfg = json.loads(bytes(z[f"{tag}_cfg"]).decode())
        m = LogMelSpectrogram(**cfg).eval().cuda()
        assert set(m.state_dict()) == {"spectrogram.window", "mel_scale.fb"}
        y = m(torch.from_numpy(z[f"{tag}_wave"]).cuda())
        assert y.shape == z[f"{tag}_logmel"].shape
        # |d| on log-mel: the floor log(1e-5) region amplifies relative error of tiny mel energies; values here are O(1)
        assert np.abs(y.cpu().numpy() - z[f"{tag}_logmel"]).max() <= 2e-4
    # 44.1 kHz config, longer signal, vs the oracle
    cfg = dict(sample_rate=44100, n_fft=2048, win_length=2048, hop_length=512, n_mels=80, f_min=0.0, f_max=22050)
    rng = np.random.default_rng(0)
    wave = (0.2 * rng.normal(size=(3, 512 * 40))).astype(np.float32)
    m = LogMelSpectrogram(**cfg).eval().cuda()
    mel = m(torch.from_numpy(wave).cuda()[:, None, :])
    ref = orc.logmel_forward(wave, cfg)
    assert mel.shape == ref.shape == (3, 80, 40)
    assert np.abs(mel.cpu().numpy() - ref).max() <= 2e-4
    gen, _ = config.build_generator("hifigan", overrides={"num_mels": 80})
    out = gen.eval().cuda()(mel)
    assert out.shape == (3, 1, 40 * 512) and torch.isfinite(out).all()
    with pytest.raises(Exception):
        m(torch.zeros(1, 1, 100, device="cuda"))     # shorter than the reflect padding


def test_firefly_gan_base_full_size_runs_and_f16x3_agrees():
    """The shipped Firefly composition (firefly-gan-base.yaml: ConvNeXt 128..512 -> HiFiGAN head, num_mels 512, k = 13
    pre / post convs) at full size: too big for the CPU oracle in a test, so check the size-independent properties —
    finite, bounded by the tanh, batch items independent — and that the opt-in f16x3 engine lands on the same waveform."""
    from vocoder_amd import _lib, synthetic as syn
    from vocoder_amd.engine import Engine, convnext_config, upsampler_config
    cfg = dict(syn.FIREFLY_BASE_44K)
    sd = syn.firefly_state_dict(cfg, 0)
    mel = torch.from_numpy(syn.synthetic_mel(3, 128, 40, 7)).cuda()
    outs = {}
    for prec in ("f32", "f16x3"):
        eng = Engine(_lib.FV_MODEL_FIREFLY, backbone=convnext_config(**cfg["backbone"]), ups=upsampler_config(**cfg["head"]),
                     state_dict=sd, precision=prec)
        y = eng(mel)
        torch.cuda.synchronize()
        assert tuple(y.shape) == (3, 1, 40 * 512) and bool(torch.isfinite(y).all()) and float(y.abs().max()) <= 1.0
        y1 = eng(mel[1:2].contiguous())
        torch.cuda.synchronize()
        assert float((y1 - y[1:2]).abs().max()) <= 2e-6     # same clip alone: tile choice may differ, values may not
        outs[prec] = y.clone()
        eng.close()
    assert float((outs["f32"] - outs["f16x3"]).abs().max()) <= 1e-4


@pytest.mark.parametrize("sr,n_fft,hop,n_mels", [(44100, 2048, 512, 128), (24000, 1024, 256, 100), (24000, 3072, 2048, 100)])
def test_logmel_frontend_at_every_shipped_resolution_vs_oracle(sr, n_fft, hop, n_mels):
    """configs/model/resolution/*.yaml through the log-mel front-end (f1), including hop 2048 with n_fft 3072."""
    from vocoder_amd.data.transforms import LogMelSpectrogram
    from oracle import oracle as orc
    cfg = dict(sample_rate=sr, n_fft=n_fft, win_length=n_fft, hop_length=hop, n_mels=n_mels, f_min=0.0, f_max=sr // 2)
    wave = (0.1 * np.random.default_rng(n_fft).normal(size=(2, hop * 7))).astype(np.float32)
    m = LogMelSpectrogram(**cfg).eval().cuda()
    mel = m(torch.from_numpy(wave).cuda()[:, None, :])
    ref = orc.logmel_forward(wave, cfg)
    assert mel.shape == ref.shape == (2, n_mels, 7)
    assert np.abs(mel.cpu().numpy() - ref).max() <= 2e-4


def test_cli_wav_branch_matches_the_direct_pipeline(tmp_path):
    """inference.main on a stereo .wav (test.py:50-71: channels -> batch items, log-mel on the GPU, generator) writes what
    LogMelSpectrogram -> generator produce directly; a wrong sampling rate is refused (no resampler here)."""
    from vocoder_amd import inference
    gen, cfg = config.build_generator("hifigan", overrides={"num_mels": 80})
    sd = syn.hifigan_state_dict(dict(syn.HIFIGAN_V1_44K), 3)
    torch.save({"state_dict": {f"generator.{k}": torch.from_numpy(v) for k, v in sd.items()}}, tmp_path / "g.ckpt")
    rng = np.random.default_rng(1)
    wav = (0.3 * rng.normal(size=(512 * 9, 2))).astype(np.float32).clip(-1, 1)
    (tmp_path / "in").mkdir()
    inference.write_wav(tmp_path / "in" / "a.wav", wav, 44100)
    inference.main(["--generator", "hifigan", "--num-mels", "80", "--ckpt-path", str(tmp_path / "g.ckpt"),
                    "--input-path", str(tmp_path / "in"), "--output-path", str(tmp_path / "out")])
    got, sr = inference.read_wav(tmp_path / "out" / "a.wav")
    assert sr == 44100 and got.shape == (2, 512 * 9)
    model = inference.build_model("hifigan", overrides={"num_mels": 80}, ckpt_path=tmp_path / "g.ckpt")
    y, _ = inference.read_wav(tmp_path / "in" / "a.wav")
    want = model(torch.from_numpy(y)[:, None].cuda())[0].squeeze(1).cpu().numpy()
    assert np.abs(got - np.clip(want, -1, 1)).max() <= 2.0 / 32767
    inference.write_wav(tmp_path / "in" / "b.wav", wav, 22050)
    with pytest.raises(ValueError, match="does not resample"):
        inference.main(["--generator", "hifigan", "--num-mels", "80", "--ckpt-path", str(tmp_path / "g.ckpt"),
                        "--input-path", str(tmp_path / "in" / "b.wav"), "--output-path", str(tmp_path / "out")])


def test_bigvgan_with_the_one_parameter_snake_activation_vs_oracle():
    """BigVGANGenerator(activation=Snake) (bigvgan.py:18-71,266): only activation_post becomes Snake (no beta); the AMPBlocks
    are built without the argument and keep SnakeBeta with their beta keys (bigvgan.py:330,335-337) — so a reference checkpoint
    of such a model loads strictly (apart from the derived filter buffers)."""
    from vocoder_amd.modules.generators.bigvgan import BigVGANGenerator, Snake
    from oracle import oracle as orc
    cfg = dict(hop_length=16, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], resblock_kernel_sizes=[3, 7],
               resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], num_mels=20, upsample_initial_channel=64, use_template=False)
    sd = syn.bigvgan_state_dict(cfg, 6, post_beta=False)
    gen = BigVGANGenerator(**cfg, activation=Snake).eval()
    own = set(gen.state_dict())
    assert "activation_post.act.beta" not in own and "resblocks.0.activations.0.act.beta" in own
    missing, unexpected = gen.load_state_dict(_t(sd), strict=False)   # (the alias-free filter buffers are not in the synthetic dict)
    assert not unexpected and all(k.endswith("filter") for k in missing), (missing, unexpected)
    mel = syn.synthetic_mel(2, 20, 11, 4)
    y = gen.cuda()(torch.from_numpy(mel).cuda()).cpu().numpy()
    ref = orc.bigvgan_forward(sd, cfg, mel)
    assert np.abs(y - ref).max() <= TOL
    with pytest.raises(NotImplementedError):
        BigVGANGenerator(**cfg, activation=torch.nn.ReLU)


def test_linear_spectrogram_standalone_vs_reference_golden_and_oracle():
    """LinearSpectrogram.forward on its own (spectrogram.py:25-56; the VAE encoders' input): the log-mel engine with the
    filterbank stage switched off (fv_logmel_config.n_mels = 0)."""
    import json
    from vocoder_amd.data.transforms.spectrogram import LinearSpectrogram
    from oracle import oracle as orc
    z = load_golden("logmel.npz")
    for tag in "ab":
        cfg = json.loads(bytes(z[f"{tag}_cfg"]).decode())
        m = LinearSpectrogram(cfg["n_fft"], cfg["win_length"], cfg["hop_length"]).eval().cuda()
        assert set(m.state_dict()) == {"window"}
        y = m(torch.from_numpy(z[f"{tag}_wave"]).cuda()).cpu().numpy()
        if f"{tag}_linear" in z:
            assert y.shape == z[f"{tag}_linear"].shape and np.abs(y - z[f"{tag}_linear"]).max() <= 1e-4 * max(1.0, np.abs(z[f"{tag}_linear"]).max())
        ref = orc.linear_spectrogram(z[f"{tag}_wave"], cfg["n_fft"], cfg["win_length"], cfg["hop_length"])
        assert y.shape == ref.shape and np.abs(y - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    m = LinearSpectrogram(3072, 3072, 2048).eval().cuda()      # 1537 bins: the vocos-small-vae encoder's input at 24000_2048_3072
    wave = (0.1 * np.random.default_rng(0).normal(size=(2, 2048 * 5))).astype(np.float32)
    y = m(torch.from_numpy(wave).cuda()[:, None, :]).cpu().numpy()
    ref = orc.linear_spectrogram(wave, 3072, 3072, 2048)
    assert y.shape == ref.shape == (2, 1537, 5) and np.abs(y - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_module_level_batch_invariance_switch():
    """`gen.batch_invariant = True` / `gen.conv_algorithm` on a drop-in module (include/fishvoc.h fv_set_batch_invariant, fv_set_conv_algorithm):
    a clip forwarded alone equals the same clip inside a batch bit for bit; the default stays within 2e-5."""
    from vocoder_amd.modules.generators.hifigan import HiFiGANGenerator
    cfg = dict(hop_length=64, upsample_rates=[4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
               resblock_dilation_sizes=[[1, 3, 5]] * 3, num_mels=80, upsample_initial_channel=256, use_template=False,
               pre_conv_kernel_size=7, post_conv_kernel_size=7)
    sd = syn.hifigan_state_dict(cfg, seed=4)
    gen = HiFiGANGenerator(**cfg)
    gen.load_state_dict(_t(sd), strict=True)
    gen = gen.eval().cuda()
    mel = torch.from_numpy(syn.synthetic_mel(24, 80, 60, seed=6)).cuda()
    y = gen(mel).clone()
    y1 = gen(mel[7:8]).clone()
    assert float((y1 - y[7:8]).abs().max()) <= 2e-5
    with pytest.raises(ValueError):
        gen.conv_algorithm = "fft"
    for algo in ("auto", "direct", "winograd"):
        gen.conv_algorithm = algo
        gen.batch_invariant = True
        yi = gen(mel).clone()
        assert torch.equal(gen(mel[7:8]), yi[7:8]) and torch.equal(gen(mel[20:24]), yi[20:24]), algo
        assert float((yi - y).abs().max()) <= 2e-5
        gen.batch_invariant = False
