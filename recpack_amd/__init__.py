"""recpack_amd - MI355X-native ItemKNN / EASE behind recpack's fit()/predict()."""
from recpack_amd.algorithms import EASE, EASE_XY, EASEGram, ItemKNN, get_context  # noqa: F401

from recpack_amd.time_aware import (TARSItemKNN, TARSItemKNNCoocDistance, TARSItemKNNDing, TARSItemKNNHermann, TARSItemKNNLee,  # noqa: F401,E402
                                    TARSItemKNNLiu, TARSItemKNNLiu2012, TARSItemKNNVaz, TARSItemKNNXia, TimedInteractions)

__all__ = ["ItemKNN", "EASE", "EASE_XY", "EASEGram", "get_context", "TARSItemKNN", "TARSItemKNNDing", "TARSItemKNNLiu",
           "TARSItemKNNVaz", "TARSItemKNNLiu2012", "TARSItemKNNLee", "TARSItemKNNCoocDistance", "TARSItemKNNXia", "TARSItemKNNHermann",
           "TimedInteractions"]
