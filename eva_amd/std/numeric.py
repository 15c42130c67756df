"""Vector reductions built from rotations (role of /root/reference/python/eva/std/numeric.py:5-21)."""
from .. import py_to_eva


def horizontal_sum(vector):
    """Every slot of the result holds the sum of all slots of `vector`: log2(vec_size) rotate-and-add
    steps, the rotation distance doubling each time (vec_size is a power of two)."""
    total = py_to_eva(vector)
    distance = 1
    size = total.program.vec_size
    while distance < size:
        total = total + (total << distance)
        distance *= 2
    return total
