"""MI355X-native quadruplet-loss fine-tuning path.

Host-side mirror of the reference interface for the hot path only:
`SentenceTransformer` (encode/fit/__call__), `models.losses.{QuadrupletLoss,
GammaQuadrupletLoss}`; everything numeric goes through the C-ABI in
csrc/ (libqst.so) -- there is no CPU or torch fallback.
"""
from .config import EncoderConfig, PRESETS, build_layout  # noqa: F401

__all__ = ["EncoderConfig", "PRESETS", "build_layout"]
