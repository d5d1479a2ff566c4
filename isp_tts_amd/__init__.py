"""MI355X-native acoustic-model forward path of isp-tts.

Host-side mirror of the reference's operator surface (`tts.modules.transformer`, `tts.modules.aligner`,
`tts.models.acoustic`) over hand-written gfx950 kernels reached through the C ABI in include/ispk.h.
"""
__version__ = "0.1.0"
