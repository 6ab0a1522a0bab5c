"""iris (MI355X build) -- the HiFiGAN vocoder path of ZECTBynmo/iris-tts, rebuilt for gfx950.

Only the vocoder hot path exists here: ``iris.vocoder`` and
``iris.hifigan_pretrained`` are drop-ins for the reference modules of the same names; the
arithmetic runs in hand-written HIP behind the C-ABI of ``include/iris_hifigan.h``.
Put this directory's parent (``iris-tts_amd/``) on ``PYTHONPATH`` in place of the reference's
``src/``.
"""

__version__ = "0.1.0"
