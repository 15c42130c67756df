"""Drop-in alias: `import eva` resolves to the MI355X-native package `eva_amd`, so programs and
tests written against microsoft/EVA's Python API run unchanged."""
import sys as _sys

import eva_amd as _impl
from eva_amd import *  # noqa: F401,F403
from eva_amd import Expr, EvaProgram, Input, Output, py_to_eva, evaluate, save, load, set_num_threads  # noqa: F401
from eva_amd import ckks, seal, metric, std  # noqa: F401
import eva_amd.std.numeric as _numeric

for _name, _mod in (("ckks", ckks), ("seal", seal), ("metric", metric), ("std", std), ("std.numeric", _numeric)):
    _sys.modules[__name__ + "." + _name] = _mod
