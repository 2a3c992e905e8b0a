"""Radar configuration metadata (TI mmWave .cfg -> derived scalars)."""
from . import cfgManager as _m

ConfigManager = _m.ConfigManager
ConfigNotLoaded = _m.ConfigNotLoaded
InvalidConfiguration = _m.InvalidConfiguration
__all__ = ["ConfigManager", "ConfigNotLoaded", "InvalidConfiguration"]
