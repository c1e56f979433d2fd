"""Drop-in for the reference's python/metrics.py (SI-SDR/SI-SIR/SI-SAR + statistics)."""
from vaenmf.metrics import energy_ratios, mean_confidence_interval, compute_stats  # noqa: F401
