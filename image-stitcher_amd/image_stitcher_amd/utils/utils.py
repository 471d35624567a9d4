"""@timer (per-call wall time at DEBUG, reference utils/utils.py:13-20) and the tqdm shims."""
import functools
import time


def tqdm(**kwargs):
    from tqdm import tqdm as _tqdm
    kwargs["bar_format"] = "{desc:<5.5}{percentage:3.0f}%|{bar:20}{r_bar}"
    return _tqdm(**kwargs)


def no_tqdm(**kwargs):
    return kwargs["iterable"]


def timer(func):
    @functools.wraps(func)
    def timed(self, *args, **kwargs):
        t0 = time.time()
        out = func(self, *args, **kwargs)
        self.logger.debug(f"{func.__name__} took {time.time() - t0} seconds")
        return out
    return timed
