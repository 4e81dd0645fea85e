"""Alias of probav_amd.parseConfig (reference path utils/parseConfig.py)."""
from probav_amd.parseConfig import parseConfig  # noqa: F401
