#!/usr/bin/env python
"""Compile / key generation / client / server as four steps that only share files — the flow of the
reference's examples/serialization.py.  The program, parameters and signature files are in EVA's
own wire format (they load in microsoft/EVA too); valuations and key contexts use this
repository's container (SEAL's binary object format is not implemented).

    python examples/client_server_files.py [directory]"""
import os
import sys
import tempfile

from eva import EvaProgram, Input, Output, evaluate, save, load
from eva.ckks import CKKSCompiler
from eva.metric import valuation_mse
from eva.seal import generate_keys

work = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="eva_files_")
at = lambda name: os.path.join(work, name)

# ---- developer: compile once
poly = EvaProgram('Polynomial', vec_size=8)
with poly:
    x = Input('x')
    Output('y', 3 * x ** 2 + 5 * x - 2)
poly.set_output_ranges(20)
poly.set_input_scales(20)
compiled, params, signature = CKKSCompiler().compile(poly)
save(compiled, at('poly.eva'))
save(params, at('poly.evaparams'))
save(signature, at('poly.evasignature'))

# ---- key owner
public_ctx, secret_ctx = generate_keys(load(at('poly.evaparams')))
save(public_ctx, at('poly.public'))
save(secret_ctx, at('poly.secret'))

# ---- client: encrypt
signature = load(at('poly.evasignature'))
inputs = {'x': [float(i) for i in range(signature.vec_size)]}
save(load(at('poly.public')).encrypt(inputs, signature), at('inputs.vals'))

# ---- server: evaluate on the GPU, never sees the secret key
server_ctx = load(at('poly.public'))
save(server_ctx.execute(load(at('poly.eva')), load(at('inputs.vals'))), at('outputs.vals'))

# ---- client: decrypt and compare with the computation in the clear
outputs = load(at('poly.secret')).decrypt(load(at('outputs.vals')), load(at('poly.evasignature')))
reference = evaluate(load(at('poly.eva')), inputs)
print("y =", [round(v, 3) for v in outputs['y']])
print("MSE vs clear evaluation:", valuation_mse(outputs, reference), " files in", work)
