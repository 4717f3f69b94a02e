"""recpack_amd - MI355X-native ItemKNN / EASE behind recpack's fit()/predict()."""
from recpack_amd.algorithms import EASE, EASE_XY, EASEGram, ItemKNN, get_context  # noqa: F401

__all__ = ["ItemKNN", "EASE", "EASE_XY", "EASEGram", "get_context"]
