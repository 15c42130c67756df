#!/usr/bin/env python
"""Sobel edge detection and Harris corner detection on an encrypted image — the two workloads of the
reference's examples/image_processing.py (BASELINE configs 2 and 3) on the MI355X backend.

    python examples/image_filters.py [sobel|harris] [--size 64] [--pgm out.pgm]

The image is synthetic (a few rectangles and a gradient: the reference's baboon.png is not part of
this repository); the encrypted result is compared with the same filter computed in the clear."""
import argparse
import math
import time

from eva import EvaProgram, Input, Output, evaluate
from eva.ckks import CKKSCompiler
from eva.metric import valuation_mse
from eva.seal import generate_keys


def window(image, width, weights):
    """sum over a 3x3 window: rotations of the row-major image times per-tap weights"""
    total = None
    for dy in range(3):
        for dx in range(3):
            tap = (image << (dy * width + dx)) * weights[dy][dx]
            total = tap if total is None else total + tap
    return total


def sqrt_poly(x):
    """cubic least-squares fit of sqrt on [0, 64] (the approximation the reference's Sobel uses)"""
    return x * 2.214 + (x ** 2) * -1.098 + (x ** 3) * 0.173


def sobel_program(h, w):
    prog = EvaProgram('sobel', vec_size=h * w)
    with prog:
        image = Input('image')
        gx = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]
        gy = [[gx[j][i] for j in range(3)] for i in range(3)]
        horizontal, vertical = window(image, w, gx), window(image, w, gy)
        Output('image', sqrt_poly(horizontal ** 2 + vertical ** 2))
    prog.set_input_scales(25)
    prog.set_output_ranges(10)
    return prog


def harris_program(h, w, kappa=0.04):
    prog = EvaProgram('harris', vec_size=h * w)
    with prog:
        image = Input('image')
        gx = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]
        gy = [[gx[j][i] for j in range(3)] for i in range(3)]
        box = [[1, 1, 1]] * 3
        ix, iy = window(image, w, gx), window(image, w, gy)
        sxx, syy, sxy = window(ix * ix, w, box), window(iy * iy, w, box), window(ix * iy, w, box)
        Output('image', (sxx * syy - sxy * sxy) - (sxx + syy) ** 2 * kappa)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    return prog


def synthetic_image(h, w):
    img = [[0.15 + 0.5 * x / w for x in range(w)] for _ in range(h)]
    for (y0, y1, x0, x1, v) in ((h // 8, h // 2, w // 8, w // 3, 0.9), (h // 2, 7 * h // 8, w // 2, 7 * w // 8, 0.05)):
        for y in range(y0, y1):
            for x in range(x0, x1):
                img[y][x] = v
    return [p for row in img for p in row]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("filter", nargs="?", default="sobel", choices=["sobel", "harris"])
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--pgm", help="write the decrypted result as a binary PGM image")
    args = ap.parse_args()
    h = w = args.size
    assert h * w == 2 ** math.ceil(math.log2(h * w)), "the image must fill a power-of-two vector"
    prog = sobel_program(h, w) if args.filter == "sobel" else harris_program(h, w)
    compiled, params, signature = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    print(f"{args.filter}: N = {params.poly_modulus_degree}, primes = {list(params.prime_bits)}, {len(params.rotations)} rotation keys")
    public_ctx, secret_ctx = generate_keys(params)
    inputs = {'image': synthetic_image(h, w)}
    encrypted = public_ctx.encrypt(inputs, signature)
    public_ctx.execute(compiled, encrypted)            # first call: eager walk
    public_ctx.execute(compiled, encrypted)            # second: hipGraph capture
    public_ctx.synchronize()
    t0 = time.perf_counter()
    result = public_ctx.execute(compiled, encrypted)   # replay; the valuations stay in HBM, the call does not wait
    public_ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    outputs = secret_ctx.decrypt(result, signature)
    clear = evaluate(compiled, inputs)
    print(f"execute(): {ms:.2f} ms on the GPU;  MSE vs the filter in the clear: {valuation_mse(outputs, clear):.3e}")
    if args.pgm:
        px = outputs['image']
        lo, hi = min(px), max(px)
        data = bytes(int(255 * (v - lo) / (hi - lo + 1e-12)) for v in px)
        with open(args.pgm, "wb") as f:
            f.write(f"P5 {w} {h} 255\n".encode() + data)
        print("wrote", args.pgm)


if __name__ == "__main__":
    main()
