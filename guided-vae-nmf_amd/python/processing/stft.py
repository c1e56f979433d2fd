"""Drop-in for the reference's python/processing/stft.py:16-24, 66-73 (GPU STFT/iSTFT)."""
from vaenmf.stft import stft, istft  # noqa: F401
