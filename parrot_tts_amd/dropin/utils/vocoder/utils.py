from parrot_tts_amd.checkpoint import scan_checkpoint  # noqa: F401
from parrot_tts_amd.vocoder import AttrDict, get_padding  # noqa: F401
