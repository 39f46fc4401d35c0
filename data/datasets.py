"""Import-path shim: `from data.datasets import TTSDataset, TTSPreprocessor` (train_tts.py:10)."""
from transformertts_amd.data.datasets import Dataset, TTSDataset, TTSPreprocessor  # noqa: F401
