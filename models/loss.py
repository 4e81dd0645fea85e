"""Alias of probav_amd.loss (reference path models/loss.py)."""
from probav_amd.loss import *  # noqa: F401,F403
from probav_amd import loss as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
