"""eva_amd — MI355X-native execution backend behind EVA's Python surface.

Same names as the reference's `eva` package (/root/reference/python/eva/__init__.py:57-163):
EvaProgram, Expr, Input, Output, py_to_eva, evaluate, set_num_threads; submodules `ckks`
(CKKSCompiler), `seal` (generate_keys -> public/secret contexts; execute() runs on the GPU through
libeva_hip.so), `metric`, `std.numeric`.  `eva_amd.backend` is the raw ctypes view of the C-ABI.
Importing this package does not import torch.
"""
import numbers
import operator

from . import _hipruntime  # noqa: F401  (one HIP runtime per process; must precede the native modules)
from ._eva import *  # noqa: F401,F403  (Program, Term, Op, Type, evaluate, save, load, set_num_threads)
from . import _eva

__version__ = "0.1.0"

class _Scope:
    """The one EvaProgram a `with` block has made current (Input / Output / constants go there)."""
    active = None

    @classmethod
    def program(cls):
        if cls.active is None:
            raise RuntimeError("No Program in context (Input, Output and constants need an enclosing `with EvaProgram(...)` block)")
        return cls.active


def _as_term(value, program):
    """native Term for an Expr, a Term, a list (dense constant) or a number (uniform constant)"""
    if isinstance(value, Expr):
        return value.term
    if isinstance(value, _eva.Term):
        return value
    if isinstance(value, list):
        return program._make_dense_constant(value)
    if isinstance(value, numbers.Number):
        return program._make_uniform_constant(value)
    raise TypeError(f"cannot use a {type(value).__name__} as an EVA term: {value!r}")


def py_to_eva(x, program=None):
    """Expr for an Expr, Term, list or number; constants are created in `program` (default: the
    program of the enclosing `with` block)."""
    if isinstance(x, Expr):
        return x
    target = program if program is not None else _Scope.program()
    return Expr(_as_term(x, target), target)


class Expr:
    """A value of an EvaProgram under construction: a native Term plus the program it belongs to.
    Python arithmetic on it appends terms to that program (lists and numbers become constants)."""

    def __init__(self, term, program):
        self.term = term
        self.program = program

    def _node(self, op, *operands):
        terms = [_as_term(o, self.program) for o in operands]
        return Expr(self.program._make_term(op, terms), self.program)

    def __add__(self, other):
        return self._node(Op.Add, self, other)

    def __radd__(self, other):
        return self._node(Op.Add, other, self)

    def __sub__(self, other):
        return self._node(Op.Sub, self, other)

    def __rsub__(self, other):
        return self._node(Op.Sub, other, self)

    def __mul__(self, other):
        return self._node(Op.Mul, self, other)

    def __rmul__(self, other):
        return self._node(Op.Mul, other, self)

    def __neg__(self):
        return self._node(Op.Negate, self)

    def __pow__(self, exponent):
        """x ** n for a positive integer n: a left-leaning chain of n - 1 products (the compiler's
        reduction balancer reshapes it); the reference lowers powers the same way"""
        try:  # any integral type (int, numpy integers, ...), as the reference accepts; bool is not a count
            if isinstance(exponent, bool):
                raise TypeError
            exponent = operator.index(exponent)
        except TypeError:
            raise ValueError(f"only positive integer powers are supported, got {exponent!r}") from None
        if exponent < 1:
            raise ValueError(f"only positive integer powers are supported, got {exponent!r}")
        power = self
        for _ in range(exponent - 1):
            power = power * self
        return power

    def __lshift__(self, steps):
        return Expr(self.program._make_left_rotation(self.term, steps), self.program)

    def __rshift__(self, steps):
        return Expr(self.program._make_right_rotation(self.term, steps), self.program)


class EvaProgram(Program):
    """Program that can be made current with `with`: inside the block Input, Output and constants
    refer to it."""

    def __init__(self, name, vec_size):
        super().__init__(name, vec_size)

    def __enter__(self):
        if _Scope.active is not None:
            raise RuntimeError("`with EvaProgram` blocks do not nest: another program is already current")
        _Scope.active = self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        if _Scope.active is not self:
            raise RuntimeError("leaving a `with EvaProgram` block that is not the current one")
        _Scope.active = None


def Input(name, is_encrypted=True):
    """A named input of the current program: encrypted (Cipher) unless is_encrypted is False (Raw)."""
    program = _Scope.program()
    return Expr(program._make_input(name, Type.Cipher if is_encrypted else Type.Raw), program)


def Output(name, expr):
    """Names `expr` as an output of the current program."""
    program = _Scope.program()
    program._make_output(name, _as_term(expr, program))
