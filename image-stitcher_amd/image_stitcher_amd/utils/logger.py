"""Thin logging facade with the call shape of the reference's utils/logger.py (positional args joined
by `sep`)."""
import logging


class LogWrapper:
    def __init__(self, logger):
        self.logger = logger

    @staticmethod
    def _join(args, sep):
        return sep.join(str(a) for a in args)

    def debug(self, *args, sep=" "):
        self.logger.debug(self._join(args, sep))

    def info(self, *args, sep=" "):
        self.logger.info(self._join(args, sep))

    def warning(self, *args, sep=" "):
        self.logger.warning(self._join(args, sep))

    def error(self, *args, sep=" "):
        self.logger.error(self._join(args, sep))

    def critical(self, *args, sep=" "):
        self.logger.critical(self._join(args, sep))

    def exception(self, *args, sep=" "):
        self.logger.exception(self._join(args, sep))

    def off(self):
        self.logger.setLevel(logging.CRITICAL)
