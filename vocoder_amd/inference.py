"""Inference wrapper with the contract of the reference's task model + test CLI.

* ``InferenceModel.forward(audio, mask=None, input_spec=None) -> (fake_audio, 0)`` is what
  ``GANModel.forward`` promises ``test.py:89`` (reference models/gan.py:282-288).  With a ``mel_transform`` attached
  (the GPU log-mel front-end) the ``audio`` branch works too: ``input_spec = mel_transform(audio.squeeze(1))``
  (gan.py:284).
* ``load_generator_state_dict`` accepts a raw generator state dict or a Lightning checkpoint
  (``{"state_dict": {"generator.<key>": ...}}``, test.py:32-37).
* ``main`` mirrors test.py:50-99: ``.pt``/``.pth`` mels (add a batch dim to 2-D mels, transpose when the last dim is
  ``num_mels``) and ``.wav`` audio (channels become batch items, log-mel on the GPU), write ``(B, 1, T)`` -> wav.
  PCM wav only and no resampling (librosa / soundfile are not available here): other inputs are refused, not guessed.
"""
from __future__ import annotations

import argparse
import time
import wave
from pathlib import Path

import numpy as np
import torch
from torch import nn

from . import config as fvconfig


class InferenceModel(nn.Module):
    def __init__(self, generator: nn.Module, sampling_rate: int = 44100, num_mels: int = 128, hop_length: int = 512,
                 mel_transform: nn.Module | None = None):
        super().__init__()
        self.generator = generator
        self.mel_transform = mel_transform   # mel_transforms.input of the reference's task model (gan.yaml:4,31)
        self.sampling_rate, self.num_mels, self.hop_length = sampling_rate, num_mels, hop_length

    @property
    def device(self):
        return next(self.generator.parameters()).device

    def forward(self, audio, mask=None, input_spec=None):
        if input_spec is None:
            if self.mel_transform is None or audio is None:
                raise NotImplementedError(
                    "wave -> mel (mel_transforms.input, gan.py:284) needs a mel_transform (build_model attaches one); "
                    "pass input_spec= otherwise")
            input_spec = self.mel_transform(audio.squeeze(1) if audio.dim() == 3 else audio)
        return self.generator(input_spec), 0


def load_generator_state_dict(ckpt, trust_checkpoint: bool = False) -> dict:
    """Raw state dict, or Lightning checkpoint with the ``generator.`` prefix (other prefixes — discriminators,
    mel_transforms — are dropped, they are not part of the generator).

    Files are read with ``torch.load(weights_only=True)``.  The reference (test.py:32) uses a plain ``torch.load``, and real
    Lightning ``.ckpt`` files usually carry non-tensor objects (hyper_parameters as a DictConfig, callbacks, optimizer
    state) that the weights-only unpickler rejects: pass ``trust_checkpoint=True`` (CLI ``--trust-checkpoint``) to fall
    back to a full unpickle for a file you trust, or export just the tensors once with
    ``torch.save({"state_dict": ckpt["state_dict"]}, "weights.pt")``."""
    if isinstance(ckpt, (str, Path)):
        path = ckpt
        import pickle
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=True)
        except OSError:
            raise          # missing / unreadable file: not a trust question, and no second (unsafe) attempt
        except (pickle.UnpicklingError, RuntimeError) as exc:   # the refusal itself, or a RuntimeError wrapping it (by torch version)
            refused = isinstance(exc, pickle.UnpicklingError) or any(
                m in str(exc) for m in ("weights_only", "Weights only", "Unsupported global", "unsupported global", "UnpicklingError"))
            if not refused:
                raise      # e.g. a corrupt zip archive: the full unpickler would fail the same way
            if not trust_checkpoint:
                raise RuntimeError(
                    f"{path}: the weights-only loader refused this checkpoint ({type(exc).__name__}: {str(exc)[:200]}). "
                    "Lightning checkpoints often hold non-tensor objects (hyper_parameters, callbacks). If you trust the "
                    "file, pass trust_checkpoint=True / --trust-checkpoint, or re-save only its 'state_dict'.") from exc
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" in ckpt:
        ckpt = ckpt["state_dict"]
    if any(k.startswith("generator.") for k in ckpt):
        ckpt = {k[len("generator."):]: v for k, v in ckpt.items() if k.startswith("generator.")}
    return dict(ckpt)


def prepare_mel(mel: torch.Tensor, num_mels: int) -> torch.Tensor:
    """test.py:78-82: (T, M) or (M, T) or batched; returns (B, num_mels, T) float32."""
    mel = mel.to(torch.float32)
    if mel.dim() == 2:
        mel = mel[None]
    if mel.shape[-1] == num_mels:
        mel = mel.transpose(1, 2)
    return mel.contiguous()


def diffsinger_mel_to_ln(mel: torch.Tensor) -> torch.Tensor:
    """DiffSinger log10 mels -> the natural-log convention of the generators (scripts/convert_diffsinger_mel.py:9-17)."""
    return mel / 0.434294


def write_wav(path, audio: np.ndarray, sampling_rate: int) -> None:
    """audio: (T,) or (T, C) float in [-1, 1] -> 16-bit PCM (soundfile is not available here)."""
    a = np.asarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[:, None]
    pcm = (np.clip(a, -1.0, 1.0) * 32767.0).round().astype("<i2")
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(int(sampling_rate))
        w.writeframes(pcm.tobytes())


def read_wav(path) -> tuple[np.ndarray, int]:
    """PCM wav (8 / 16 / 24 / 32 bit) -> ((channels, T) float32 in [-1, 1), sampling rate), like
    ``librosa.load(path, sr=None, mono=False)`` for such files (test.py:51)."""
    with wave.open(str(path), "rb") as w:
        nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        a = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        a = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        a = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return np.ascontiguousarray(a.reshape(-1, nch).T), int(sr)


def build_model(generator="hifigan", resolution="44100_512_2048", overrides=None, config_root=None,
                ckpt_path=None, device="cuda", trust_checkpoint=False) -> InferenceModel:
    from .data.transforms import LogMelSpectrogram
    gen, cfg = fvconfig.build_generator(generator, resolution, overrides, config_root)
    if ckpt_path is not None:
        gen.load_state_dict(load_generator_state_dict(ckpt_path, trust_checkpoint), strict=True)
    m = cfg["model"]
    # configs/model/spectrogram/mel.yaml:1-8 (mel_transforms.input)
    mel = LogMelSpectrogram(sample_rate=m["sampling_rate"], n_fft=m["n_fft"], win_length=m["win_length"],
                            hop_length=m["hop_length"], n_mels=m["num_mels"], f_min=0, f_max=m["sampling_rate"] // 2)
    model = InferenceModel(gen, m["sampling_rate"], m["num_mels"], m["hop_length"], mel_transform=mel)
    return model.eval().to(device)


@torch.no_grad()
def main(argv=None):
    ap = argparse.ArgumentParser(description="mel (.pt) or audio (.wav) -> wav on the MI355X engine (mirrors fish_vocoder/test.py)")
    ap.add_argument("--generator", default="hifigan")
    ap.add_argument("--resolution", default="44100_512_2048")
    ap.add_argument("--config-root", default=None, help="use another configs/ tree (e.g. the reference's)")
    ap.add_argument("--ckpt-path", required=True)
    ap.add_argument("--input-path", required=True)
    ap.add_argument("--output-path", required=True)
    ap.add_argument("--num-mels", type=int, default=None)
    ap.add_argument("--diffsinger", action="store_true", help="inputs are log10 mels")
    ap.add_argument("--trust-checkpoint", action="store_true",
                    help="fall back to a full unpickle (like the reference's plain torch.load) when the weights-only loader "
                         "refuses a Lightning checkpoint that holds non-tensor objects")
    a = ap.parse_args(argv)
    model = build_model(a.generator, a.resolution, {"num_mels": a.num_mels} if a.num_mels else None, a.config_root,
                        a.ckpt_path, trust_checkpoint=a.trust_checkpoint)
    inp = Path(a.input_path)
    files = [inp] if inp.is_file() else sorted(p for p in inp.rglob("*") if p.suffix in (".pt", ".pth", ".wav"))
    base = inp.parent if inp.is_file() else inp
    for f in files:
        if f.suffix == ".wav":   # test.py:50-71: channels -> batch items, zero padding, mel_transforms.input
            y, sr = read_wav(f)
            if sr != model.sampling_rate:
                raise ValueError(f"{f}: {sr} Hz, the model runs at {model.sampling_rate} Hz and this tool does not resample")
            y = torch.from_numpy(y)[:, None]
            y = torch.nn.functional.pad(y, (0, model.hop_length - (model.hop_length % y.shape[-1])))   # as written upstream
            mel = model.mel_transform(y.to(model.device).squeeze(1))
        else:
            mel = torch.load(f, map_location="cpu", weights_only=True)
            if a.diffsinger:
                mel = diffsinger_mel_to_ln(mel)
            mel = prepare_mel(mel, model.num_mels).to(model.device)
        t0 = time.time()
        fake = model(None, None, input_spec=mel)[0]
        torch.cuda.synchronize()
        print(f"{f}: {fake.shape[-1] / model.sampling_rate:.2f}s of audio in {time.time() - t0:.4f}s")
        out = Path(a.output_path) / f.relative_to(base).with_suffix(".wav")
        write_wav(out, fake.squeeze(1).cpu().numpy().T, model.sampling_rate)


if __name__ == "__main__":
    main()
