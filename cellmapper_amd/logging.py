"""Package logger (level from the ``LOGLEVEL`` environment variable, like the reference's logging.py:7-25)."""
import logging
import os


def _make_logger() -> logging.Logger:
    log = logging.getLogger("cellmapper_amd")
    log.setLevel(os.environ.get("LOGLEVEL", "INFO"))
    if not log.handlers:
        try:
            from rich.console import Console
            from rich.logging import RichHandler

            handler = RichHandler(show_path=False, console=Console(stderr=True), show_time=False)
        except Exception:  # rich is optional
            handler = logging.StreamHandler()
            handler.setFormatter(logging.Formatter("%(levelname)s %(message)s"))
        log.addHandler(handler)
    log.propagate = False
    return log


logger = _make_logger()
