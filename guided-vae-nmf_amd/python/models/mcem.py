"""Drop-in for the reference's python/models/mcem.py: same import path, same classes.
`from python.models.mcem import MCEM_M1, MCEM_M2` (scripts/evaluate_M1.py:13,
scripts/evaluate_M2_vad.py) resolves here when guided-vae-nmf_amd/ is on sys.path."""
from vaenmf.mcem import MCEM_M1, MCEM_M2, EM_noNMF, MCEM_M2_noNMF  # noqa: F401
