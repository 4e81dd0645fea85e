"""Alias of probav_amd.modelsTF (reference path models/modelsTF.py)."""
from probav_amd.modelsTF import *  # noqa: F401,F403
from probav_amd import modelsTF as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
