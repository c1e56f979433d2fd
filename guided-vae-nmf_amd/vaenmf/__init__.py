"""vaenmf: MI355X-native engine for the VAE-NMF reconstruct path."""
from . import _lib
from .models import VariationalAutoencoder, DeepGenerativeModel, Classifier, Classifier2Classes, Encoder, Decoder
from .mcem import MCEM_M1, MCEM_M2, EM_noNMF, MCEM_M2_noNMF
from .engine import BatchEngine
