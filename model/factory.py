from transformertts_amd.model.factory import tts_custom, tts_ljspeech  # noqa: F401
