from transformertts_amd.data.audio import Audio, MelGAN, WaveRNN  # noqa: F401
