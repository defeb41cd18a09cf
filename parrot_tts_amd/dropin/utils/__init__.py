"""``utils`` as the reference drivers import it.  With ``parrot_tts_amd/dropin`` on sys.path both spellings
work: ``from utils import AttrDict`` and ``from utils.vocoder.models import CodeGenerator`` (the latter fails
in the reference itself when run from the repo root)."""
from parrot_tts_amd.checkpoint import scan_checkpoint  # noqa: F401
from parrot_tts_amd.vocoder import AttrDict, get_padding  # noqa: F401
