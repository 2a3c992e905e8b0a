from .logger import get_logger, setup_logger

__all__ = ["get_logger", "setup_logger"]
