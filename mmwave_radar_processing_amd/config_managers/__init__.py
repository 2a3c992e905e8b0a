from .cfgManager import ConfigManager, ConfigNotLoaded, InvalidConfiguration

__all__ = ["ConfigManager", "ConfigNotLoaded", "InvalidConfiguration"]
