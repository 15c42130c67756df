"""CKKS compiler (reference: /root/reference/python/eva/ckks/__init__.py)."""
from .._eva._ckks import *  # noqa: F401,F403
