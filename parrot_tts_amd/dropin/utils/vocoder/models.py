from parrot_tts_amd.vocoder import CodeGenerator  # noqa: F401
