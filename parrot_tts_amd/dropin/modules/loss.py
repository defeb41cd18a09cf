class ModelLoss:
    """Training loss of the reference (modules/loss.py) -- training is out of scope for this build."""

    def __init__(self, *a, **k):
        raise NotImplementedError("parrot_tts_amd covers the inference path only; ModelLoss is a training component")
