"""Small host helpers: `timer` (logs a method's wall time at DEBUG through the owning object's logger, as the
reference's decorator of the same name does) and the two progress-bar factories ArbitraryStitcher chooses
between (`with_tqdm`)."""
import functools
import time

BAR_FORMAT = "{desc:<5.5}{percentage:3.0f}%|{bar:20}{r_bar}"


def tqdm(**options):
    """tqdm with the reference's bar layout; imported lazily so that the package loads without tqdm"""
    import tqdm as _tqdm_module
    return _tqdm_module.tqdm(**dict(options, bar_format=BAR_FORMAT))


def no_tqdm(iterable=None, **_ignored):
    """stand-in used when with_tqdm is false: hands the iterable back untouched"""
    return iterable


def timer(method):
    @functools.wraps(method)
    def timed(self, *args, **kwargs):
        began = time.perf_counter()
        try:
            return method(self, *args, **kwargs)
        finally:
            self.logger.debug(f"{method.__name__} took {time.perf_counter() - began} seconds")
    return timed
