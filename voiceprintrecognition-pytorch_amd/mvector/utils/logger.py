"""``logger`` with the handful of loguru methods the package uses; loguru itself is used when installed."""
import logging

try:  # pragma: no cover - loguru is optional
    from loguru import logger  # noqa: F401
except ImportError:
    class _Logger:
        def __init__(self):
            self._log = logging.getLogger('mvector')
            if not self._log.handlers:
                h = logging.StreamHandler()
                h.setFormatter(logging.Formatter('%(asctime)s | %(levelname)-7s | %(message)s'))
                self._log.addHandler(h)
                self._log.setLevel(logging.INFO)

        def info(self, msg, *a):
            self._log.info(msg, *a)

        def warning(self, msg, *a):
            self._log.warning(msg, *a)

        def error(self, msg, *a):
            self._log.error(msg, *a)

        def debug(self, msg, *a):
            self._log.debug(msg, *a)

        def exception(self, msg, *a):
            self._log.exception(msg, *a)

    logger = _Logger()
