"""CPU model of the 29-bit-limb arithmetic of the bucket accumulation (csrc/field29.hip.h, curve29.hip.h) with Python integers.

The GPU parity tests feed the kernels random data; they cannot show that the LAZY representation never overflows: a limb that leaves 32 bits,
a column sum that leaves 64 bits, a subtraction whose limb goes negative.  This model restates every operation of xyzz29_madd limb by limb with
assertions on exactly those events, drives it with random operands, with operands pushed to the stated entry bounds, and with long chains, and
checks every result against plain modular arithmetic.  Test infrastructure only (like oracle/): the product never runs it."""
import random

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # BN254 Fq (fq.hpp:11-41)
M29 = (1 << 29) - 1
R1 = 1 << 261  # R'
INV29 = (-pow(P, -1, 1 << 29)) % (1 << 29)
U32, U64 = 1 << 32, 1 << 64


def limbs(v):
    assert 0 <= v < (1 << (29 * 8 + 32))
    return [(v >> (29 * j)) & M29 for j in range(8)] + [v >> 232]


def val(a):
    return sum(x << (29 * i) for i, x in enumerate(a))


P29 = limbs(P)


def from_fe(x, shift):  # f29_from_fe<P, S>: the limb split of x << S
    assert x < (1 << 256)
    return limbs(x << shift)


def carry(a):  # f29_carry
    assert all(0 <= x < U32 for x in a)
    r = [a[0] & M29] + [(a[i] & M29) + (a[i - 1] >> 29) for i in range(1, 8)] + [a[8] + (a[7] >> 29)]
    assert all(x < U32 for x in r)
    return r


def spread(mult, e):  # Spread29<P, J, M, E>
    q = limbs(mult * P)
    up, down = 1 << e, 1 << (e - 29)
    s = [q[0] + up] + [q[j] + up - down for j in range(1, 8)] + [q[8] - down]
    assert val(s) == mult * P and s[8] >= 0
    return s


def sub(a, b, mult, e=30):  # f29_sub<M, E>: a + (M p, spread) - b limbwise, unsigned 32-bit arithmetic
    s = spread(mult, e)
    r = []
    for x, y, c in zip(a, b, s):
        assert c - y >= 0, "subtrahend limb above the spread constant"
        t = x + (c - y)
        assert t < U32, "limb overflow in f29_sub"
        r.append(t)
    return r


def mont(chains):  # product scanning over 29-bit limbs; chains = [(a, b), ...]: sum of a*b over all chains, one reduction
    acc, m, r = 0, [0] * 9, [0] * 9
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        for a, b in chains:
            for i in range(lo, hi + 1):
                acc += a[i] * b[k - i]
                assert acc < U64, "column overflow (a*b)"
        for i in range(lo, hi + 1 if k > 8 else k):
            acc += m[i] * P29[k - i]
            assert acc < U64, "column overflow (m*p)"
        if k <= 8:
            m[k] = ((acc & 0xFFFFFFFF) * INV29) & M29
            acc += m[k] * P29[0]
            assert acc < U64 and acc & M29 == 0
        else:
            r[k - 9] = acc & M29
        acc >>= 29
    assert acc < U32
    r[8] = acc
    want = sum(val(a) * val(b) for a, b in chains)
    assert (val(r) * R1 - want) % P == 0 and val(r) < want // R1 + P + 1
    return r


def mul(a, b):
    return mont([(a, b)])


def sqr(a):  # f29_sqr: cross terms once against the doubled operand -- the same column sums, so the same overflow behaviour as mont([(a, a)])
    d = [2 * x for x in a]
    assert all(x < U32 for x in d)
    return mont([(a, a)])


def mul_sub2(a, b, c, d):  # f29_mul_sub2: a*b + (64p - c)*d
    nc = sub([0] * 9, c, 64, 30)
    r = mont([(a, b), (nc, d)])
    assert (val(r) * R1 - (val(a) * val(b) - val(c) * val(d))) % P == 0
    return r


ONE = limbs(R1 % P)


def madd(acc, px, py):  # xyzz29_madd; px, py: table coordinates (canonical R-form words), the sign already applied
    x1, y1, zz1, zzz1 = acc
    x2, y2 = from_fe(px, 5), from_fe(py, 5)
    u2, s2 = mul(x2, zz1), mul(y2, zzz1)
    pp_, rr_ = carry(sub(u2, x1, 34)), carry(sub(s2, y1, 34))
    pp, rr = sqr(pp_), sqr(rr_)
    ppp, q = mul(pp_, pp), mul(x1, pp)
    s = [a + 2 * b for a, b in zip(ppp, q)]
    x3 = carry(sub(rr, s, 12, 31))
    t = carry(sub(q, x3, 24))
    y3 = mul_sub2(rr_, t, y1, ppp)
    zzz3 = mul(zzz1, ppp)
    zz3 = mul(zz1, pp)
    return [x3, y3, zz3, zzz3]


def madd_mod(acc, px, py):  # the same formulas on residues (values are x * R' mod p; a product of two such values carries one factor R' too many)
    x1, y1, zz1, zzz1 = acc
    ri = pow(R1, -1, P)
    x2, y2 = px * 32 % P, py * 32 % P
    m = lambda a, b: a * b * ri % P
    u2, s2 = m(x2, zz1), m(y2, zzz1)
    p_, r_ = (u2 - x1) % P, (s2 - y1) % P
    pp, ppp = m(p_, p_), None
    ppp = m(p_, pp)
    q = m(x1, pp)
    x3 = (m(r_, r_) - ppp - 2 * q) % P
    y3 = (m(r_, q - x3) - m(y1, ppp)) % P
    return [x3, y3, m(zz1, pp), m(zzz1, ppp)]


def start(px, py):
    return [from_fe(px, 5), from_fe(py, 5), list(ONE), list(ONE)]


def check(acc, ref):
    for a, r in zip(acc, ref):
        assert val(a) % P == r


def test_products_at_the_limb_bounds():
    rng = random.Random(29)
    top = (32 * P) >> 232
    worst = [[(1 << 29) + 7] * 8 + [top], [M29] * 8 + [top], limbs(P - 1), limbs(32 * (P - 1))]
    for a in worst:
        for b in worst:
            mul(a, b)
            sqr(a)
    for _ in range(200):
        a, b = limbs(rng.randrange(32 * P)), limbs(rng.randrange(32 * P))
        assert val(mul(a, b)) < 32 * 32 * P * P // R1 + P + 1
    # the double product with the operand bounds of the mixed addition: R, T carried (< 35.3p, < 26.6p), Y1 < 32p, PPP < 2.8p
    hi = lambda bound: carry(limbs(bound * P - 1))
    mul_sub2(hi(36), hi(27), limbs(32 * P - 1), limbs(3 * P))


def test_mixed_addition_chains_against_modular_arithmetic():
    rng = random.Random(2929)
    for chain in range(20):
        px, py = rng.randrange(1, P), rng.randrange(1, P)
        acc = start(px, py)
        ref = [px * 32 % P, py * 32 % P, R1 % P, R1 % P]
        check(acc, ref)
        for step in range(40):
            px, py = rng.randrange(1, P), rng.randrange(1, P)
            if step % 7 == 3:
                px, py = P - 1, P - 1            # largest canonical coordinates
            if step % 7 == 5:
                px, py = 1, P - 1
            acc, ref = madd(acc, px, py), madd_mod(ref, px, py)
            check(acc, ref)
            x3, y3, zz3, zzz3 = acc
            assert val(x3) < 21 * P and val(y3) < 8 * P and val(zz3) < 2 * P and val(zzz3) < 2 * P  # the exit bounds curve29.hip.h states
            assert all(l < (1 << 29) + 8 for l in x3[:8] + y3[:8] + zz3[:8] + zzz3[:8])


def test_mixed_addition_from_the_entry_bounds():
    """One addition from accumulators pushed to the stated entry bounds (X, Y < 32p, ZZ, ZZZ < 1.4p, limbs < 2^29 + 8)."""
    rng = random.Random(31)
    for _ in range(100):
        def lazy(bound_num, bound_den):
            v = rng.randrange(bound_num * P // bound_den - (1 << 240), bound_num * P // bound_den)
            a = limbs(v)
            k = rng.randrange(8)
            if a[k + 1] > 0 and a[k] + (1 << 29) < (1 << 29) + 8:  # move one unit of the limb above down: a limb just over 29 bits
                a[k + 1] -= 1
                a[k] += 1 << 29
            return a
        acc = [lazy(32, 1), lazy(32, 1), lazy(14, 10), lazy(14, 10)]
        for a in acc:
            a[0] |= 7  # low limbs odd and close to the carry pass's maximum
        ref = [val(a) % P for a in acc]
        px, py = rng.choice([P - 1, rng.randrange(1, P)]), rng.choice([P - 1, rng.randrange(1, P)])
        check(madd(acc, px, py), madd_mod(ref, px, py))


def test_mixed_addition_from_extreme_limbs():
    """Every limb of every accumulator coordinate at the carry pass's maximum (2^29 + 7), top limbs at the stated value bounds."""
    full = (1 << 29) + 7
    x_top = ((32 * P) >> 232) - 2
    z_top = (14 * P // 10) >> 232
    acc = [[full] * 8 + [x_top], [full] * 8 + [x_top], [full] * 8 + [z_top], [full] * 8 + [z_top]]
    ref = [val(a) % P for a in acc]
    for px, py in ((P - 1, P - 1), (1, 1), (P - 1, 1), ((1 << 253) - 1, (1 << 253) + 12345)):
        check(madd(acc, px, py), madd_mod(ref, px, py))


def test_division_by_32_conversion():
    """f29_div32_to_fe: x R' / 32 = x R from any lazily reduced value the accumulation can hold (< 21p), result < 2p, exact."""
    rng = random.Random(5)
    ninv5 = (-pow(P, -1, 32)) % 32
    for _ in range(500):
        v = rng.randrange(21 * P)
        a = limbs(v)
        m = (a[0] * ninv5) & 31
        t = [x + y for x, y in zip(a, limbs(m * P))]
        assert all(x < U32 for x in t)
        tv = val(t)
        assert tv % 32 == 0 and tv >> 5 < 2 * P and tv >> 5 < (1 << 256)
        assert ((tv >> 5) * 32 - v) % P == 0
