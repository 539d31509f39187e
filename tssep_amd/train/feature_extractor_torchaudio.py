"""Import-path parity with tssep/train/feature_extractor_torchaudio.py."""
from .feature_extractor import TorchMFCC  # noqa: F401
