"""Pure-Python big-integer references used to pin oracle/ (independent of any C code).

Definitions follow SURVEY.md Appendix A (restating SEAL 3.6; call sites
/root/reference/eva/seal/seal_executor.h:124-213).
"""
from functools import reduce


def bitrev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def naive_ntt(a, psi, q):
    """out[i] = sum_j a_j psi^((2 br(i)+1) j) mod q  (A.2)"""
    n = len(a)
    logn = n.bit_length() - 1
    out = []
    for i in range(n):
        e = 2 * bitrev(i, logn) + 1
        w = pow(psi, e, q)
        acc, p = 0, 1
        for j in range(n):
            acc = (acc + a[j] * p) % q
            p = p * w % q
        out.append(acc)
    return out


def negacyclic_mul(a, b, q):
    n = len(a)
    out = [0] * n
    for i in range(n):
        if a[i] == 0:
            continue
        for j in range(n):
            k = i + j
            if k < n:
                out[k] = (out[k] + a[i] * b[j]) % q
            else:
                out[k - n] = (out[k - n] - a[i] * b[j]) % q
    return out


def crt(residues, primes):
    Q = reduce(lambda x, y: x * y, primes, 1)
    x = 0
    for r, p in zip(residues, primes):
        Qi = Q // p
        x = (x + r * Qi * pow(Qi, -1, p)) % Q
    return x, Q


def divide_round(X, d):
    """floor((X + d//2) / d) — the rounding rule of rescale / key-switch mod-down (A.5, A.6)"""
    return (X + (d >> 1)) // d


def apply_galois_coeff(a, elt, q):
    """a(X) -> a(X^elt) mod (X^n + 1), coefficient form"""
    n = len(a)
    out = [0] * n
    for j in range(n):
        e = (j * elt) % (2 * n)
        if e < n:
            out[e] = (out[e] + a[j]) % q
        else:
            out[e - n] = (out[e - n] - a[j]) % q
    return out


def centered(x, Q):
    x %= Q
    return x - Q if x > Q // 2 else x
