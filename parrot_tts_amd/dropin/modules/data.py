from parrot_tts_amd.data import (DFATokenizer, ParrotDataset, get_mask_from_batch, get_mask_from_lengths)  # noqa: F401
