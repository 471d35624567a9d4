"""Logging facade with the call shape the stitchers use everywhere: `log.info(a, b, c, sep=" ")` joins its
positional arguments like print() and forwards one string to a standard `logging.Logger` (the reference's
utils/logger.py has the same surface).  The six level methods are generated, not written out."""
import logging

_LEVELS = ("debug", "info", "warning", "error", "critical", "exception")


class LogWrapper:
    def __init__(self, logger):
        self.logger = logger
        for level in _LEVELS:
            setattr(self, level, self._forwarder(getattr(logger, level)))

    @staticmethod
    def _forwarder(emit):
        def send(*parts, sep=" "):
            emit(sep.join(map(str, parts)))
        return send

    def off(self):
        """silence everything below CRITICAL (log_level=0)"""
        self.logger.setLevel(logging.CRITICAL)
