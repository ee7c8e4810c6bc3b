"""Drop-in for the reference's `models.losses` package (models/losses/__init__.py:1 re-exports GammaQuadrupletLoss)."""
from .losses import GammaQuadrupletLoss  # noqa: F401
