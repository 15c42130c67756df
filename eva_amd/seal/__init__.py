"""Execution backend (reference: /root/reference/python/eva/seal/__init__.py).  The names are the
reference's; the evaluator behind execute() is the MI355X library libeva_hip.so."""
from .._eva._seal import *  # noqa: F401,F403
