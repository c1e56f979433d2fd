"""Drop-in for the reference's python/models/spp_estimation.py:163-235 (GPU)."""
from vaenmf.spp_estimation import timo_mask_estimation, timo_vad_estimation, timo_noise_estimation  # noqa: F401
