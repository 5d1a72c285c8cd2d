from .spectrogram import LinearSpectrogram, LogMelSpectrogram  # noqa: F401
