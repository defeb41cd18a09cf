"""``dataset`` as reference utils/vocoder/inference.py:24 imports it (bare import with utils/vocoder on sys.path)."""
from parrot_tts_amd.data import CodeDataset, mel_spectrogram, parse_manifest, parse_speaker  # noqa: F401
from parrot_tts_amd.vocoder import MAX_WAV_VALUE  # noqa: F401
