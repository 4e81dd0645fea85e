"""Alias of probav_amd.testClass (reference path models/testClass.py)."""
from probav_amd.testClass import *  # noqa: F401,F403
from probav_amd import testClass as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
