"""The half-size-scalar reformulation of the verification equation (halfgcd.h) is EXACT: same
verdict as the reference equation for every on-curve input, including points with a small-order
component.  Integer model only (CPU); the GPU kernels are checked against the oracle in
tests/test_gpu_parity.py."""
import random

import pymodel as M

rnd = random.Random(2024)
N = 8 * M.R_ORDER


def _sqrt(n, p=M.Q):
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(n, q, p), pow(n, (q + 1) // 2, p)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c, t, r = i, b * b % p, t * b * b % p, r * b % p
    return r


def order8_point():
    while True:
        v = rnd.randrange(M.Q)
        u2 = (v * v - 1) * pow(1 + M.D * v * v, -1, M.Q) % M.Q
        if pow(u2, (M.Q - 1) // 2, M.Q) != 1:
            continue
        p = (_sqrt(u2), v)
        assert M.on_curve(p)
        t = M.pmul(p, M.R_ORDER)
        if M.pmul(t, 4) != M.IDENTITY:
            return t


def test_half_scalars_congruence_parity_and_size():
    sizes = []
    for c in [0, 1, 2, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, (1 << 250) - 1, 1 << 249,
              (1 << 200) + 1, N >> 6, (N >> 5) - 1] + \
             [rnd.getrandbits(250) for _ in range(3000)]:
        a, b, bn = M.half_scalars(c)
        sb = -b if bn else b
        assert (a - sb * c) % N == 0
        assert b & 1 and 0 < b < (1 << 160) and 0 <= a < (1 << 250)
        sizes.append(max(a.bit_length(), b.bit_length()))
    sizes.sort()
    assert sizes[len(sizes) // 2] <= 130 and sizes[int(0.99 * len(sizes))] <= 137


def test_half_scalar_check_is_exact_with_torsion():
    t8 = order8_point()
    valid = 0
    for _ in range(12):
        sk, m, rr = rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER)
        k1 = rnd.randrange(8)
        pk = M.padd(M.pmul(M.GEN, sk), M.pmul(t8, k1))      # key with a small-order component
        for k2 in range(8):
            R = M.padd(M.pmul(M.GEN, rr), M.pmul(t8, k2))   # nonce point with one too
            c = M.challenge(R, m)
            u = (rr - c * sk) % M.R_ORDER
            want = M.verify_single(u, R, pk, m)
            assert M.verify_single_half(u, R, pk, m) == want
            assert want == ((c * k1 - k2) % 8 == 0)          # valid iff the torsion parts cancel
            valid += want
            assert not M.verify_single_half((u + 1) % M.R_ORDER, R, pk, m)
    assert valid >= 3  # each (k1, k2) pair is valid with probability 1/8


def test_lattice3_congruences_parity_and_size():
    """lattice3.h (var-generator kernel): x = z*u, y = z*c (mod 8r), z odd, all three ~170 bits"""
    sizes = []
    edge = [(0, 0), (1, 1), (0, 5), (M.R_ORDER - 1, (1 << 250) - 1), (12345, 0), (M.R_ORDER - 1, 1),
            (1 << 200, 1 << 100), (M.R_ORDER - 1, (1 << 250) - 2), (7, (1 << 250) - 3)]
    for u, c in edge + [(rnd.randrange(M.R_ORDER), rnd.getrandbits(250)) for _ in range(300)]:
        x, y, z = M.lattice3(u, c)
        assert (x - z * u) % N == 0 and (y - z * c) % N == 0
        assert z & 1 and 0 < abs(z) < M.R_ORDER
        assert max(abs(x), abs(y), abs(z)) < (1 << 252)
        sizes.append(max(abs(v).bit_length() for v in (x, y, z)))
    rand = sorted(sizes[len(edge):])
    assert rand[len(rand) // 2] <= 171 and rand[-1] <= 176


def test_lattice3_check_is_exact_with_torsion():
    """The three-scalar form gives the reference's verdict for generators, keys and nonce points with a
    small-order component: valid iff the torsion parts cancel (u*k0 + c*k1 = k2 mod 8)."""
    t8 = order8_point()
    valid = 0
    for _ in range(6):
        sk, m, rr, g = (rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER),
                        rnd.randrange(1, M.R_ORDER))
        k0, k1 = rnd.randrange(8), rnd.randrange(8)
        gen0 = M.pmul(M.GEN, g)
        gen = M.padd(gen0, M.pmul(t8, k0))                   # generator with a small-order component
        pk = M.padd(M.pmul(gen0, sk), M.pmul(t8, k1))        # so has the key
        for k2 in range(8):
            R = M.padd(M.pmul(gen0, rr), M.pmul(t8, k2))
            c = M.challenge(R, m)
            u = (rr - c * sk) % M.R_ORDER
            want = M.verify_vargen(u, R, pk, gen, m)
            assert M.verify_vargen_lattice(u, R, pk, gen, m) == want
            assert want == ((u * k0 + c * k1 - k2) % 8 == 0)
            valid += want
            assert not M.verify_vargen_lattice((u + 1) % M.R_ORDER, R, pk, gen, m)
    assert valid >= 2


def test_euclidean_inversion_model_is_exact_and_bounded():
    """inv29.h's algorithm on Python integers and floats (pymodel.inv_euclid): the quotient estimates
    never exceed the true quotient (asserted inside the model at every step), the result is the
    inverse for random values and for values built to stress it — neighbours of powers of two,
    consecutive-Fibonacci ratios (all quotients 1: the longest chains), q / k (one huge first
    quotient), values one apart from their partner after a step —, the irregular inputs are reported
    as fall-backs (the device then runs Fermat), and the half-step count stays far below the cap."""
    import random
    rnd = random.Random(99)
    q = M.Q
    fib = [1, 2]
    while fib[-1] < q:
        fib.append(fib[-1] + fib[-2])
    cases = [q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, q // 3, fib[-2], fib[-3], q - fib[-3], (1 << 254) + 1]
    cases += [(1 << k) + d for k in range(33, 255, 13) for d in (-1, 0, 1)]
    cases += [q // k for k in (3, 5, 1000003, (1 << 30) + 7)]
    random_cases = [rnd.randrange(1 << 230, q) for _ in range(3000)]
    cases += random_cases + [rnd.randrange(1, 1 << rnd.randrange(40, 255)) for _ in range(1000)]
    worst = 0
    regular = set(random_cases)
    for x in cases:
        inv, steps, fell_back = M.inv_euclid(x)
        assert inv * x % q == 1, x
        worst = max(worst, steps)
        if x in regular:             # a quotient beyond 31 bits has probability ~2^-31 per step
            assert not fell_back, x
    assert worst < 2 * 400           # ~1.44 * 255 division steps at most, twice that in half-steps
    # irregular inputs: the device falls back to Fermat, the value is still the inverse (0 for 0)
    for x in (0, 1, 2, 3, 1 << 31, (1 << 100) + 1):
        inv, _, fell_back = M.inv_euclid(x)
        assert fell_back and inv == (pow(x, -1, q) if x else 0)
    # the scalar field works the same way (the model is generic in the modulus)
    for _ in range(300):
        x = rnd.randrange(1, M.R_ORDER)
        assert M.inv_euclid(x, M.R_ORDER)[0] * x % M.R_ORDER == 1
