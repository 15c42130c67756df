"""reference: /root/reference/python/eva/metric.py:6-19"""
import numpy as _np


def valuation_mse(a, b):
    """ Total mean squared error between two valuations (dict name -> list of numbers) """
    if set(a.keys()) != set(b.keys()):
        raise ValueError("Valuations must have the same keys")
    mse = 0
    for k in a.keys():
        mse += _np.mean((_np.array(a[k]) - _np.array(b[k])) ** 2)
    return mse / len(a)
