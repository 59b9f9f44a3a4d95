"""Logger factory (mirror of lmcache/logging.py:11-14; same call signature)."""
import logging

_FMT = "%(levelname)s lmcache_amd: %(message)s [%(asctime)s.%(msecs)03d]"
if not logging.getLogger().handlers:
    logging.basicConfig(format=_FMT, level=logging.WARNING)


def init_logger(name: str) -> logging.Logger:
    return logging.getLogger(name)
