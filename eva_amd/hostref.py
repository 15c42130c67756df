"""Small pure-Python host helpers (no oracle, no torch): SEAL-conformant prime chain selection.

Mirrors CoeffModulus::Create as called at /root/reference/eva/seal/seal.cpp:181-182; the C++
twin lives in csrc/hostmath.h.  Used by scripts and bench.py to name a context's primes.
"""


def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def coeff_modulus_create(N, bit_sizes):
    counts = {}
    for b in bit_sizes:
        counts[b] = counts.get(b, 0) + 1
    table = {}
    for b, cnt in counts.items():
        step = 2 * N
        v = ((1 << b) - 1) // step * step + 1
        lst = []
        while len(lst) < cnt and v > (1 << (b - 1)):
            if _is_prime(v):
                lst.append(v)
            v -= step
        if len(lst) != cnt:
            raise ValueError("failed to find enough qualifying primes")
        table[b] = lst
    return [table[b].pop() for b in bit_sizes]
