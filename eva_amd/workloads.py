"""The EVA programs BASELINE.json's configurations name, written once: bench.py, scripts/ and tests/ all build
them from here (r6: they used to live in test modules).

  C1  readme_polynomial()      3x^2 + 5x - 2                         /root/reference/README.md:134
  C2  sobel(64, 64, 4096)      Sobel 3x3 filter, N forced to 2^13    /root/reference/examples/image_processing.py:39-63
  C3  harris()                 Harris corner detector, N = 2^15      /root/reference/examples/image_processing.py:65-100
  C4  sobel(64, 64, 4096)      256 instances, N = 2^14               (the same program, execute_batch)
  C5  conv_depth8()            3x3 convolution + 8 squarings, 2^16   /root/reference/tests/large_programs.py:10-53 style

`pad_chain` is SURVEY.md 8(d)'s way of reaching the stated number of data limbs; `image` is the synthetic
input (the reference's baboon.png is not part of this repository)."""
from . import EvaProgram, Input, Output


def pad_chain(params, n_primes, N):
    """force N and pad prime_bits with 60-bit primes (after the output prime) up to
    n_primes = L + 1; legal because the reference builds its context with sec_level none
    (/root/reference/eva/seal/seal.cpp:169)."""
    params.poly_modulus_degree = N
    pb = list(params.prime_bits)
    if len(pb) < n_primes:
        params.prime_bits = pb[:1] + [60] * (n_primes - len(pb)) + pb[1:]


def image(n, shift=0, scale=255.0):
    """{'image': n synthetic pixels in [0, 1]}; `shift` gives the instances of a batch distinct images"""
    return {'image': [((37 * i + shift) % 256) / scale for i in range(n)]}


def readme_polynomial():
    poly = EvaProgram('Polynomial', vec_size=1024)
    with poly:
        x = Input('x')
        Output('y', 3 * x ** 2 + 5 * x - 2)
    poly.set_output_ranges(30)
    poly.set_input_scales(30)
    return poly


def _convolution_xy(img, width, filt):
    """both directional derivatives from ONE set of rotations (examples/image_processing.py:22-34)"""
    for i in range(3):
        for j in range(3):
            rotated = img << (i * width + j)
            horizontal = rotated * filt[i][j]
            vertical = rotated * filt[j][i]
            if i == 0 and j == 0:
                Ix, Iy = horizontal, vertical
            else:
                Ix += horizontal
                Iy += vertical
    return Ix, Iy


def _convolution(img, width, filt):
    for i in range(3):
        for j in range(3):
            partial = (img << i * width + j) * filt[i][j]
            convolved = partial if (i == 0 and j == 0) else convolved + partial
    return convolved


SOBEL_FILTER = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]


def sobel(h, w, vec):
    """Sobel magnitude through the cubic square-root fit; scales / ranges are the caller's (the example uses 25 / 10)"""
    prog = EvaProgram('sobel', vec_size=vec)
    with prog:
        img = Input('image')
        a1, a2, a3 = 2.2137874823876622, -1.0984324107372518, 0.17254603006834726
        ch, cv = _convolution_xy(img, w, SOBEL_FILTER)
        x = ch ** 2 + cv ** 2
        Output('image', x * a1 + x ** 2 * a2 + x ** 3 * a3)
    return prog


def sobel_example():
    """the example's own settings: 64x64 image, input scale 2^25, output range 2^10"""
    prog = sobel(64, 64, 4096)
    prog.set_input_scales(25)
    prog.set_output_ranges(10)
    return prog


def harris(h=64, w=64, c=0.04):
    prog = EvaProgram('harris', vec_size=h * w)
    with prog:
        img = Input('image')
        pool = [[1, 1, 1], [1, 1, 1], [1, 1, 1]]
        Ix, Iy = _convolution_xy(img, w, SOBEL_FILTER)
        Ixx, Iyy, Ixy = Ix ** 2, Iy ** 2, Ix * Iy
        Sxx, Syy, Sxy = _convolution(Ixx, w, pool), _convolution(Iyy, w, pool), _convolution(Ixy, w, pool)
        det = Sxx * Syy - Sxy * Sxy
        trace = Sxx + Syy
        Output('image', det - trace ** 2 * c)
    prog.set_input_scales(30)
    prog.set_output_ranges(20)
    return prog


def conv_depth8():
    deep = EvaProgram('conv+depth8', vec_size=4096)
    with deep:
        img = Input('image')
        acc = None
        for i in range(3):
            for j in range(3):
                t = (img << (i * 64 + j)) * (1.0 / 9.0)
                acc = t if acc is None else acc + t
        for _ in range(8):
            acc = acc * acc
        Output('y', acc)
    deep.set_input_scales(30)
    deep.set_output_ranges(20)
    return deep


def compile_config(name):
    """-> (compiled, params, signature, inputs) of BASELINE config `name` in {"c1", …, "c5"} at its stated size"""
    from .ckks import CKKSCompiler
    prog, N, n_primes, inputs = {
        "c1": (readme_polynomial, None, 0, {'x': [i / 1024.0 for i in range(1024)]}),
        "c2": (sobel_example, 8192, 0, image(4096)),
        "c3": (harris, 32768, 9, image(4096)),
        "c4": (sobel_example, 16384, 6, image(4096)),
        "c5": (conv_depth8, 65536, 13, image(4096)),
    }[name]
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog())
    if n_primes:
        pad_chain(params, n_primes, N)
    elif N:
        params.poly_modulus_degree = N
    return compiled, params, sig, inputs
