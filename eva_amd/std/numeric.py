"""reference: /root/reference/python/eva/std/numeric.py:5-21"""
from .. import py_to_eva


def horizontal_sum(x):
    """ Sum all elements of a vector; the result is replicated in every element. """
    x = py_to_eva(x)
    i = 1
    while i < x.program.vec_size:
        y = x << i
        x = x + y
        i <<= 1
    return x
