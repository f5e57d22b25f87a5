"""Counterpart of the reference's ``predictive_coding/utils.py`` (/root/reference/predictive_coding/utils.py:4-16): the two helpers
that module exports, under the same names."""
import warnings


def _is_positive_int(x):
    return isinstance(x, int) and x > 0


def slow_down_warning(base, prop, solution):
    """Same message and category as the reference's (predictive_coding/utils.py:8-16)."""
    warnings.warn(
        "In {}, you have {} enabled, this will slow down training. Set to {} to disable it. ".format(base, prop, solution),
        category=RuntimeWarning)
