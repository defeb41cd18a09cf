from parrot_tts_amd.tte import Parrot  # noqa: F401
