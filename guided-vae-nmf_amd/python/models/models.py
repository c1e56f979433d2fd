"""Drop-in for the reference's python/models/models.py (constructor signatures and
state_dict key layout of models.py:41-62, 90-133, 184-197)."""
from vaenmf.models import (Classifier, Classifier2Classes, Decoder, DeepGenerativeModel, Encoder,  # noqa: F401
                           GaussianSample, VariationalAutoencoder)
