"""Import-name drop-in for the reference's ``modules`` package (reference modules/__init__.py:2-4):
put ``parrot_tts_amd/dropin`` on sys.path and ``from modules import ParrotDataset, Parrot`` resolves here."""
from parrot_tts_amd.data import ParrotDataset  # noqa: F401
from parrot_tts_amd.tte import Parrot  # noqa: F401
from .loss import ModelLoss  # noqa: F401
