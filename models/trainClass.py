"""Alias of probav_amd.trainClass (reference path models/trainClass.py)."""
from probav_amd.trainClass import *  # noqa: F401,F403
from probav_amd import trainClass as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
