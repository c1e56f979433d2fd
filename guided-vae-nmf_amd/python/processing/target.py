"""Drop-in for the reference's python/processing/target.py:7-116 (labels on the GPU)."""
from vaenmf.target import (clean_speech_IBM, clean_speech_VAD, noise_robust_clean_speech_VAD,  # noqa: F401
                           noise_robust_clean_speech_IBM, ideal_wiener_mask)
