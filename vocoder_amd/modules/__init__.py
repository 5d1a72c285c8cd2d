"""Host-side mirror of the reference's module tree for the generator hot path
(fish_vocoder/modules/{generators,encoders}): same class names, ctor kwargs, state-dict keys and
``forward`` contract; the arithmetic runs in libfishvoc_hip.so."""
