from parrot_tts_amd.data import parse_manifest, parse_speaker  # noqa: F401
from parrot_tts_amd.vocoder import MAX_WAV_VALUE  # noqa: F401
