"""Accuracy metric of the end-to-end tests (role of /root/reference/python/eva/metric.py:6-19)."""
import numpy as np


def valuation_mse(first, second):
    """Mean, over the named vectors two valuations share, of each vector's mean squared
    difference.  Both valuations must name exactly the same vectors."""
    names = sorted(first)
    if names != sorted(second):
        raise ValueError("valuation_mse: the two valuations name different vectors")
    if not names:
        return 0.0
    per_vector = [float(np.square(np.asarray(first[n], dtype=float) - np.asarray(second[n], dtype=float)).mean())
                  for n in names]
    return sum(per_vector) / len(per_vector)
