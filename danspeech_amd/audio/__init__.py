from .resources import load_audio, load_audio_wavPCM  # noqa: F401
