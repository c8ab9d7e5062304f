"""eraxvif5tts_amd -- MI355X-native (gfx950, hand-written HIP) F5-TTS flow-matching inference path.

Drop-in for the hot path of ``src/f5_tts`` of hungkq-1724/EraXviF5TTS:

    from eraxvif5tts_amd.infer.f5tts_wrapper import F5TTSWrapper      # f5_tts/infer/f5tts_wrapper.py
    from eraxvif5tts_amd.model import CFM, DiT                          # f5_tts/model/{cfm.py, backbones/dit.py}

All dense math runs in ``lib/libf5hip.so`` (C ABI in ``include/f5hip.h``); there is no CPU or eager-PyTorch
fallback for it: using the model without the built library or without an MI355X raises.
"""
__version__ = "0.1.0"
