"""vocoder_amd — MI355X-native (gfx950) inference engine behind the fish_vocoder generator API.

Public surface (mirrors /root/reference/fish_vocoder for the generator hot path only):

* ``vocoder_amd.modules.generators.{HiFiGANGenerator, BigVGANGenerator, ISTFTHead, UnifyGenerator}``,
  ``vocoder_amd.modules.encoders.ConvNeXtEncoder`` — drop-in ``nn.Module`` classes (same ctor kwargs, state-dict keys,
  ``forward(mel) -> waveform``);
* ``vocoder_amd.config`` — loads the reference's ``configs/model/generator/*.yaml`` keys and instantiates ``_target_``;
* ``vocoder_amd.engine`` — thin handle on the C ABI (``include/fishvoc.h``, ``csrc/libfishvoc_hip.so``);
* ``vocoder_amd.sharding`` — utterance-batch sharding helpers for one process per GPU.

The compute path is hand-written HIP only; importing works on a CPU-only box (for config/host logic), running does not.
"""
__version__ = "0.1.0"
