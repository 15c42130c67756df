"""eva_amd — MI355X-native execution backend behind EVA's Python surface.

Same names as the reference's `eva` package (/root/reference/python/eva/__init__.py:57-163):
EvaProgram, Expr, Input, Output, py_to_eva, evaluate, set_num_threads; submodules `ckks`
(CKKSCompiler), `seal` (generate_keys -> public/secret contexts; execute() runs on the GPU through
libeva_hip.so), `metric`, `std.numeric`.  `eva_amd.backend` is the raw ctypes view of the C-ABI.
Importing this package does not import torch.
"""
import numbers

from . import _hipruntime  # noqa: F401  (one HIP runtime per process; must precede the native modules)
from ._eva import *  # noqa: F401,F403  (Program, Term, Op, Type, evaluate, save, load, set_num_threads)
from . import _eva

__version__ = "0.1.0"

_current_program = None


def _curr():
    """ Returns the EvaProgram that is currently in context """
    if _current_program is None:
        raise RuntimeError("No Program in context")
    return _current_program


def _py_to_term(x, program):
    """ Maps supported types into native terms """
    if isinstance(x, Expr):
        return x.term
    elif isinstance(x, list):
        return program._make_dense_constant(x)
    elif isinstance(x, numbers.Number):
        return program._make_uniform_constant(x)
    elif isinstance(x, _eva.Term):
        return x
    raise TypeError("No conversion to Term available for " + str(x))


def py_to_eva(x, program=None):
    """ Maps supported types (Expr, Term, list, number) into Expr; constants are created in
    `program` (default: the program currently in context). """
    if isinstance(x, Expr):
        return x
    if program is None:
        program = _curr()
    return Expr(_py_to_term(x, program), program)


class Expr():
    """ Wrapper for a native Term with operator overloads that create terms in the
    associated EvaProgram. """

    def __init__(self, term, program):
        self.term = term
        self.program = program

    def _bin(self, op, a, b):
        return Expr(self.program._make_term(op, [a, b]), self.program)

    def __add__(self, other):
        return self._bin(Op.Add, self.term, _py_to_term(other, self.program))

    def __radd__(self, other):
        return self._bin(Op.Add, _py_to_term(other, self.program), self.term)

    def __sub__(self, other):
        return self._bin(Op.Sub, self.term, _py_to_term(other, self.program))

    def __rsub__(self, other):
        return self._bin(Op.Sub, _py_to_term(other, self.program), self.term)

    def __mul__(self, other):
        return self._bin(Op.Mul, self.term, _py_to_term(other, self.program))

    def __rmul__(self, other):
        return self._bin(Op.Mul, _py_to_term(other, self.program), self.term)

    def __pow__(self, exponent):
        """ Exponentiation as nested multiplication terms """
        if exponent < 1:
            raise ValueError("exponent must be greater than zero, got " + str(exponent))
        result = self.term
        for _ in range(exponent - 1):
            result = self.program._make_term(Op.Mul, [result, self.term])
        return Expr(result, self.program)

    def __lshift__(self, rotation):
        return Expr(self.program._make_left_rotation(self.term, rotation), self.program)

    def __rshift__(self, rotation):
        return Expr(self.program._make_right_rotation(self.term, rotation), self.program)

    def __neg__(self):
        return Expr(self.program._make_term(Op.Negate, [self.term]), self.program)


class EvaProgram(Program):
    """ Native Program that also acts as a context manager selecting the program the Input
    and Output free functions operate on. """

    def __init__(self, name, vec_size):
        super().__init__(name, vec_size)

    def __enter__(self):
        global _current_program
        if _current_program is not None:
            raise RuntimeError("There is already an EVA Program in context")
        _current_program = self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        global _current_program
        if _current_program is not self:
            raise RuntimeError("This program is not currently in context")
        _current_program = None


def Input(name, is_encrypted=True):
    """ Create a new named input term in the current EvaProgram """
    program = _curr()
    return Expr(program._make_input(name, Type.Cipher if is_encrypted else Type.Raw), program)


def Output(name, expr):
    """ Create a new named output term in the current EvaProgram """
    program = _curr()
    program._make_output(name, _py_to_term(expr, program))
