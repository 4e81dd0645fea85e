"""MI355X-native engine for the WDSR-B Conv3D hot path of mmbajo/PROBA-V.

Host side mirrors the reference's Python surface (models/modelsTF.py, models/loss.py,
models/trainClass.py, models/testClass.py, utils/parseConfig.py); every FLOP of the hot path runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/probav_hip.h (csrc/).  There is no
CPU or PyTorch fallback: compute entry points raise if the HIP library or a GPU is missing.
"""
from .arch import layer_table, reducer_plan            # noqa: F401
from .parseConfig import parseConfig                    # noqa: F401


def __getattr__(name):
    # heavy modules (torch, the HIP library) are imported on first use
    import importlib
    lazy = {"WDSRConv3D": "modelsTF", "Losses": "loss", "ModelTrainer": "trainClass",
            "Enhancer": "testClass"}
    if name in lazy:
        return getattr(importlib.import_module("." + lazy[name], __name__), name)
    raise AttributeError(name)
