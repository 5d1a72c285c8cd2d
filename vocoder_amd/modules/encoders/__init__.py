from .convnext import ConvNeXtEncoder  # noqa: F401
