"""Logger 'CDR' -> stdout + ./log/<Model>/<dataset>/<timestamp>.log (reference utils/logger.py:12-45)."""
import logging
import os
import re

_ANSI = re.compile(r"\x1B(?:[@-Z\\-_]|\[[0-?]*[ -/]*[@-~])")


class _StripAnsi(logging.Filter):
    def filter(self, record):
        record.msg = _ANSI.sub("", str(record.msg))
        return True


def get_logger(file_path: str = None) -> logging.Logger:
    logger = logging.getLogger("CDR")
    for h in list(logger.handlers):
        logger.removeHandler(h)
        h.close()
    fmt = logging.Formatter("[%(asctime)s] %(levelname)s %(message)s", "%Y-%m-%d %H:%M:%S")
    logger.setLevel(logging.INFO)
    sh = logging.StreamHandler()
    sh.setLevel(logging.INFO)
    sh.setFormatter(fmt)
    logger.addHandler(sh)
    if file_path is not None:
        path = os.path.join("./log", file_path)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        fh = logging.FileHandler(path)
        fh.setLevel(logging.INFO)
        fh.setFormatter(fmt)
        fh.addFilter(_StripAnsi())
        logger.addHandler(fh)
    return logger
