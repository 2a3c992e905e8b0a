"""Stdout logger with the reference's record format (reference: mmwave_radar_processing/logging/logger.py:11-65)."""
import logging
import sys

_FORMAT = "%(asctime)s | %(levelname)-8s | %(name)s | %(message)s"
_ROOT = "mmwave_radar_processing_amd"


def setup_logger(level=logging.INFO, name: str = _ROOT) -> logging.Logger:
    log = logging.getLogger(name)
    log.setLevel(level)
    if not any(getattr(h, "_mmw", False) for h in log.handlers):
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter(_FORMAT))
        h._mmw = True
        log.addHandler(h)
        log.propagate = False
    return log


def get_logger(name: str = _ROOT) -> logging.Logger:
    if not logging.getLogger(_ROOT).handlers:
        setup_logger(logging.WARNING)
    return logging.getLogger(name if name.startswith(_ROOT) else f"{_ROOT}.{name}")
