"""CPU model of the lazily reduced 29-bit-limb radix-8 NTT step (csrc/ntt29.hip.h) with Python integers, for the FIELD Fr.

Like tests/test_limb29_model.py for the bucket accumulation: random GPU data cannot show that the lazy representation never overflows (a limb
leaving 32 bits, a column sum leaving 64, a subtraction limb going negative, a value leaving the bound its consumer was sized for, the
table-driven reduction picking a row whose borrowed top limb goes negative).  The model restates n29_step8 / ntt29_reduce operation by
operation with assertions on exactly those events and on every bound stated in the header's comments, drives it with random values, with
inputs pushed to the stated entry bound (every register just below 3p, every limb as large as `carried` allows), and through chains of
steps, and checks every output against plain modular arithmetic.  Test infrastructure only: the product never runs it."""
import random

import pytest

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # BN254 Fr (fr.hpp:12-15)
M29 = (1 << 29) - 1
R1 = 1 << 261
INV29 = (-pow(R_MOD, -1, 1 << 29)) % (1 << 29)
U32, U64 = 1 << 32, 1 << 64
P = R_MOD


def limbs(v):
    assert 0 <= v < (1 << (29 * 8 + 32))
    return [(v >> (29 * j)) & M29 for j in range(8)] + [v >> 232]


def val(a):
    return sum(x << (29 * i) for i, x in enumerate(a))


P29 = limbs(P)
P_TOP = P >> 232
INV_TOP = (1 << 32) // (P_TOP + 1)


def carry(a):
    assert all(0 <= x < U32 for x in a)
    r = [a[0] & M29] + [(a[i] & M29) + (a[i - 1] >> 29) for i in range(1, 8)] + [a[8] + (a[7] >> 29)]
    assert all(x < U32 for x in r) and val(r) == val(a)
    return r


def add(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(x < U32 for x in r), "limb overflow in f29_add"
    return r


def spread(mult, e):
    q = limbs(mult * P)
    up, down = 1 << e, 1 << (e - 29)
    s = [q[0] + up] + [q[j] + up - down for j in range(1, 8)] + [q[8] - down]
    assert val(s) == mult * P and s[8] >= 0
    return s


def sub(a, b, mult, e=30):
    s = spread(mult, e)
    r = []
    for x, y, c in zip(a, b, s):
        assert c - y >= 0, "subtrahend limb above the spread constant"
        t = x + (c - y)
        assert t < U32, "limb overflow in f29_sub"
        r.append(t)
    assert val(r) == val(a) - val(b) + mult * P
    return r


def mul(a, b):
    acc, m, r = 0, [0] * 9, [0] * 9
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
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
    assert (val(r) * R1 - val(a) * val(b)) % P == 0 and val(r) < val(a) * val(b) // R1 + P + 1
    return r


PBAR29 = limbs(R1 - P)


def mulc(a, w):
    """f29_mulc (field29c.hip.h): x * w mod p for a TABLE constant w < p through the precomputed quotient multiplier wq = floor(w 2^261 / p):
    q^ from columns 7 .. 16 of x * wq, r = low 261 bits of x w + q^ (2^261 - p).  Asserts every column sum, the quotient estimate (q or q - 1)
    and the result: exact limbs, value below (2 + V / 169) p."""
    wv = val(w)
    assert wv < P and max(w) <= M29
    wq = limbs(wv * R1 // P)
    assert max(wq) <= M29 and all(x <= (1 << 31) + (1 << 29) for x in a) and val(a) < 64 * P
    acc, q = 0, [0] * 9
    for k in range(7, 17):
        lo, hi = max(0, k - 8), min(k, 8)
        for i in range(lo, hi + 1):
            acc += a[i] * wq[k - i]
            assert acc < U64, "column overflow (x * wq)"
        if k >= 9:
            q[k - 9] = acc & M29
        acc >>= 29
    assert acc < U32
    q[8] = acc
    qtrue = val(a) * val(wq) >> 261
    assert val(q) in (qtrue, qtrue - 1), "quotient estimate off by more than one"
    acc, r = 0, [0] * 9
    for k in range(9):
        for i in range(k + 1):
            acc += a[i] * w[k - i]
            assert acc < U64, "column overflow (x * w)"
        for i in range(k + 1):
            acc += q[i] * PBAR29[k - i]
            assert acc < U64, "column overflow (q * pbar)"
        r[k] = acc & M29
        acc >>= 29
    v = val(a) * wv - val(q) * P
    assert val(r) == v and 0 <= v and v * 169 < (2 * 169 + val(a) // P + 1) * P, "remainder"
    return r


SHOUP = True  # the variant under test (ntt29.hip.h BBG_NTT_SHOUP); the tests run both


def mulw(a, w):  # a product by a per-radix table value
    return mulc(a, w) if SHOUP else mul(a, w)


def vp(mont, shoup):  # a value bound that differs between the two multipliers
    return shoup if SHOUP else mont


def kp():  # multiple of p a subtraction adds when its subtrahend is a product / a sum of two products
    return 4 if SHOUP else 3


def kpp():
    return 6 if SHOUP else 4


def reduce_table():
    rows = []
    for k in range(32):
        kp = limbs(k * P)
        zs = [1 << 30] + [(1 << 30) - 2] * 7 + [-2]
        rows.append([0] * 9 if k == 0 else [(z - x) % U32 for z, x in zip(zs, kp)])
    return rows


RED = reduce_table()


def reduce(x):  # ntt29_reduce: value < 32p in, carried value < 3p out
    assert val(x) < 32 * P
    c = carry(x)
    q = (c[8] * INV_TOP) >> 32
    assert q <= val(x) // P <= q + 1, "quotient estimate off by more than one"
    k = q - 1 if q > 1 else 0
    row = RED[k]
    t = []
    for i, (a, b) in enumerate(zip(c, row)):
        s = (a + b) % U32
        if i < 8:
            assert a + b < U32, "limb overflow in the reduction"
        elif k:
            assert a - 2 - (k * P >> 232) >= 0, "top limb of the reduction went negative"
        t.append(s)
    assert val(t) == val(x) - k * P
    r = carry(t)
    assert val(r) < 3 * P and max(r[:8]) < (1 << 29) + 8
    return r


def vbound(a, v):  # value < v p (v may be fractional: checked with integers)
    assert val(a) * 1000 < int(v * 1000) * P + P, ("value bound", val(a) / P, v)


def step8(x, w1, w2, w3, tw):  # n29_step8<true>
    for a in x:
        vbound(a, 3)
        assert max(a[:8]) < (1 << 29) + 8
    x = [list(a) for a in x]

    def bfly(i, j, k, e=30):
        u, d = add(x[i], x[j]), sub(x[i], x[j], k, e)
        x[i], x[j] = u, d
    for i in range(4):
        bfly(i, i + 4, 4)
    x[5], x[6], x[7] = mulw(x[5], w1), mulw(x[6], w2), mulw(x[7], w3)
    for j in (5, 6, 7):
        vbound(x[j], vp(1.05, 2.05))
    for j in range(4):
        vbound(x[j], 6)
    vbound(x[4], 7)
    bfly(0, 2, 7, 31); bfly(1, 3, 7, 31); bfly(4, 6, kp()); bfly(5, 7, kp())
    x[3] = carry(x[3])
    x[3], x[7] = mulw(x[3], w2), mulw(x[7], w2)
    vbound(x[3], vp(1.08, 2.08)); vbound(x[7], vp(1.03, 2.04))
    for j in (0, 1, 2, 4, 6):
        x[j] = carry(x[j])
    vbound(x[0], 12); vbound(x[1], 12); vbound(x[2], 13); vbound(x[4], vp(8.05, 9.05)); vbound(x[6], vp(10, 11)); vbound(x[5], vp(2.1, 4.1))
    bfly(0, 1, 13); bfly(2, 3, kp()); bfly(4, 5, kpp()); bfly(6, 7, kp())
    for j, v in enumerate((24, 25) + vp((14.1, 16, 10.2, 12.05, 11.03, 13), (15.1, 17, 13.15, 15.05, 13.05, 15))):
        vbound(x[j], v)
    for j in range(1, 8):
        x[j] = mulw(x[j], tw[j])
        vbound(x[j], vp(1.15, 2.15))
    x[0] = reduce(x[0])
    return x


def step8_mod(x, w1, w2, w3, tw):  # the same butterfly on residues (x R' form: a product with a table value w R' divides by R' again; Shoup: plain w)
    rinv = 1 if SHOUP else pow(R1, -1, P)
    x = list(x)

    def b(i, j, w=None):
        u, d = (x[i] + x[j]) % P, (x[i] - x[j]) % P
        x[i], x[j] = u, d if w is None else d * w * rinv % P
    b(0, 4); b(1, 5, w1); b(2, 6, w2); b(3, 7, w3)
    b(0, 2); b(1, 3, w2); b(4, 6); b(5, 7, w2)
    b(0, 1); b(2, 3); b(4, 5); b(6, 7)
    for j in range(1, 8):
        x[j] = x[j] * tw[j] * rinv % P
    return x


def check_lazy(a, v, lbound):  # what a consumer was sized for
    vbound(a, v)
    assert max(a) < lbound, ("limb bound", max(a), lbound)


def step4(x, w2):  # n29_step4
    x = [list(a) for a in x]
    for a in x:
        vbound(a, 3)

    def bfly(i, j, k):
        u, d = add(x[i], x[j]), sub(x[i], x[j], k)
        x[i], x[j] = u, d
    bfly(0, 2, 4); bfly(1, 3, 4); bfly(4, 6, 4); bfly(5, 7, 4)
    x[3], x[7] = mulw(x[3], w2), mulw(x[7], w2)
    for j in (0, 1, 4, 5, 2, 6):
        x[j] = carry(x[j])
    bfly(0, 1, 7); bfly(2, 3, kp()); bfly(4, 5, 7); bfly(6, 7, kp())
    for a in x:
        check_lazy(a, 13, (1 << 31) + 8)
    return x


def step2(x):  # n29_step2
    x = [list(a) for a in x]
    for i in (0, 2, 4, 6):
        u, d = add(x[i], x[i + 1]), sub(x[i], x[i + 1], 4)
        x[i], x[i + 1] = u, d
    for a in x:
        check_lazy(a, 7, (1 << 31) + 8)
    return x


def step8_raw(x, w1, w2, w3):  # n29_step8_raw: the radix-8 butterfly without step twiddles and without reductions
    x = [list(a) for a in x]

    def bfly(i, j, k, e=30):
        u, d = add(x[i], x[j]), sub(x[i], x[j], k, e)
        x[i], x[j] = u, d
    for i in range(4):
        bfly(i, i + 4, 4)
    x[5], x[6], x[7] = mulw(x[5], w1), mulw(x[6], w2), mulw(x[7], w3)
    bfly(0, 2, 7, 31); bfly(1, 3, 7, 31); bfly(4, 6, kp()); bfly(5, 7, kp())
    x[3] = carry(x[3])
    x[3], x[7] = mulw(x[3], w2), mulw(x[7], w2)
    for j in (0, 1, 2, 4, 6):
        x[j] = carry(x[j])
    bfly(0, 1, 13); bfly(2, 3, kp()); bfly(4, 5, kpp()); bfly(6, 7, kp())
    for a in x:
        check_lazy(a, 25, (1 << 31) + 8)
    return x


def finish(x, mult_rform=None):  # n29_finish / n29_finish_mul: -> the 8 words a device array holds (< 2p)
    check_lazy(x, 25, (1 << 31) + 8)
    if mult_rform is not None:
        assert mult_rform < 2 * P
        m = limbs(mult_rform << 5)  # f29_from_fe<5>: exact limbs, value < 64p
        x = mul(x, m)
        vbound(x, 10.5)
    assert val(x) < 32 * P
    c = carry(x)
    q = (c[8] * INV_TOP) >> 32
    assert q <= val(x) // P <= q + 1 and q < 32
    qp = limbs(q * P)
    t, cy = [], 0
    for i in range(8):
        d = c[i] - qp[i]
        assert -(1 << 31) <= d < (1 << 31), "signed limb difference leaves 32 bits"
        v = d + cy
        assert -(1 << 31) <= v < (1 << 31)
        t.append(v & M29)
        cy = v >> 29  # Python's >> on negative integers is the arithmetic shift
    top = c[8] - qp[8] + cy
    assert 0 <= top < U32, "top limb of the exact difference"
    t.append(top)
    v = val(t)
    assert v == val(x) - q * P and 0 <= v < 2 * P and v < (1 << 256) and all(limb <= M29 for limb in t[:8])
    return v


def exact(v):  # a table value: exact limbs, < p
    assert v < P
    return limbs(v)


def test_reduce_table_and_estimate_over_the_whole_range():
    rng = random.Random(29)
    for k in range(32):
        for x in (k * P, k * P + 1, (k + 1) * P - 1, k * P + rng.randrange(P)):
            if x < 32 * P:
                r = reduce(limbs(x))
                assert val(r) % P == x % P
    # lazily carried inputs: limbs above 29 bits
    for _ in range(2000):
        a, b = limbs(rng.randrange(16 * P)), limbs(rng.randrange(16 * P))
        r = reduce(add(a, b))
        assert val(r) % P == (val(a) + val(b)) % P


@pytest.mark.parametrize("shoup", [True, False])
def test_step8_random_and_chained(shoup):
    global SHOUP
    SHOUP = shoup
    rng = random.Random(2929)
    for trial in range(60):
        w = [exact(rng.randrange(P)) for _ in range(3)]
        tw = [None] + [exact(rng.randrange(P)) for _ in range(7)]
        x = [limbs(rng.randrange(2 * P)) for _ in range(8)]  # what a device array holds: coarse residues < 2p, re-limbed
        ref = [val(a) % P for a in x]
        for _ in range(6):  # six dependent steps: the outputs of one are the inputs of the next
            x = step8(x, w[0], w[1], w[2], tw)
            ref = step8_mod(ref, val(w[0]), val(w[1]), val(w[2]), [0] + [val(t) for t in tw[1:]])
            assert [val(a) % P for a in x] == ref


@pytest.mark.parametrize("shoup", [True, False])
def test_step8_at_the_entry_bounds(shoup):
    """Every register just below 3p with every limb as large as a carried value allows; twiddles p - 1 (the largest table value)."""
    global SHOUP
    SHOUP = shoup
    big = 3 * P - 1
    fat = limbs(big)
    # move weight downwards: limb i gives 1 to limb i-1 as 2^29 where that keeps limb i-1 below 2^29 + 8
    for i in range(8, 0, -1):
        if fat[i] > 0 and fat[i - 1] + (1 << 29) < (1 << 29) + 8:
            fat[i] -= 1
            fat[i - 1] += 1 << 29
    assert val(fat) == big
    wmax = exact(P - 1)
    for x in ([fat] * 8, [limbs(big)] * 8, [limbs(0)] * 8, [limbs(big)] + [limbs(0)] * 7, [limbs(0)] * 7 + [limbs(big)]):
        out = step8([list(a) for a in x], wmax, wmax, wmax, [None] + [wmax] * 7)
        ref = step8_mod([val(a) % P for a in x], P - 1, P - 1, P - 1, [0] + [P - 1] * 7)
        assert [val(a) % P for a in out] == ref


@pytest.mark.parametrize("shoup", [True, False])
def test_partial_last_steps_and_the_way_out(shoup):
    """n29_step4 / n29_step2 / n29_step8_raw on inputs at the bound, followed by n29_finish with and without a multiplier: the words written
    back are the residues plain modular arithmetic gives, below 2p."""
    global SHOUP
    SHOUP = shoup
    rng = random.Random(404)
    rinv = pow(R1, -1, P)   # of the way out's Montgomery product (both variants)
    rinv_tw = 1 if SHOUP else rinv  # of the products by per-radix table values
    big = limbs(3 * P - 1)
    for trial in range(40):
        x = [big] * 8 if trial == 0 else [limbs(rng.randrange(3 * P)) for _ in range(8)]
        xm = [val(a) % P for a in x]
        w = [rng.randrange(P) for _ in range(3)] if trial else [P - 1] * 3
        # S = 2
        out = step4(x, exact(w[1]))
        ref = list(xm)

        def b(i, j, ww=None):
            u, d = (ref[i] + ref[j]) % P, (ref[i] - ref[j]) % P
            ref[i], ref[j] = u, d if ww is None else d * ww * rinv_tw % P
        b(0, 2); b(1, 3, w[1]); b(4, 6); b(5, 7, w[1]); b(0, 1); b(2, 3); b(4, 5); b(6, 7)
        assert [val(a) % P for a in out] == ref
        mult = rng.randrange(2 * P) if trial else 2 * P - 1
        for a, r in zip(out, ref):
            assert finish(a) % P == r
            # a multiplier stored as w R (R-form): x (= X R') * (w R << 5 = w R') / R' = X w R'
            assert finish(a, mult) % P == r * (mult << 5) * rinv % P
        # S = 1
        out = step2(x)
        ref = list(xm)
        b(0, 1); b(2, 3); b(4, 5); b(6, 7)
        assert [val(a) % P for a in out] == ref
        for a, r in zip(out, ref):
            assert finish(a, mult) % P == r * (mult << 5) * rinv % P
        # S = 3 without step twiddles
        out = step8_raw(x, exact(w[0]), exact(w[1]), exact(w[2]))
        ref = list(xm)
        b(0, 4); b(1, 5, w[0]); b(2, 6, w[1]); b(3, 7, w[2]); b(0, 2); b(1, 3, w[1]); b(4, 6); b(5, 7, w[1]); b(0, 1); b(2, 3); b(4, 5); b(6, 7)
        assert [val(a) % P for a in out] == ref
        for a, r in zip(out, ref):
            assert finish(a) % P == r and finish(a, mult) % P == r * (mult << 5) * rinv % P


def test_premultiplied_load():
    """The first pass's fused coset factor: x (re-limbed R-form, < 2p) times pre (R-form << 5, < 64p) is a valid step input (V < 3, exact limbs)."""
    rng = random.Random(7)
    for _ in range(200):
        x, pre = rng.randrange(2 * P), rng.randrange(2 * P)
        r = mul(limbs(x), limbs(pre << 5))
        vbound(r, 1.76 + 0.01)
        assert max(r[:8]) < (1 << 29)


def test_constant_operand_product_at_its_bounds():
    """f29_mulc (field29c.hip.h) on its own: random and extreme constants (0, 1, p - 1), operands from 0 to the largest lazily reduced value a
    step hands it (V < 26, limbs up to 2^31 + 2^29 after an uncarried subtraction), every column and the quotient estimate asserted inside mulc."""
    rng = random.Random(261)
    consts = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 253] + [rng.randrange(P) for _ in range(40)]
    for w in consts:
        for v in (0, 1, P - 1, P, 3 * P - 1, 26 * P - 1, rng.randrange(26 * P), rng.randrange(P)):
            r = mulc(limbs(v), exact(w))
            assert val(r) % P == v * w % P and max(r) <= M29
        # an uncarried operand: the difference a - b + 13 p of two carried values just below 12 p, limbs raised by the spread constant
        a, b = limbs(12 * P - 1), limbs(rng.randrange(12 * P))
        d = sub(a, b, 13)
        r = mulc(d, exact(w))
        assert val(r) % P == (val(a) - val(b)) * w % P
        fat = [min(x, (1 << 31) + (1 << 29)) for x in [(1 << 31) + (1 << 29)] * 8] + [limbs(20 * P)[8]]
        if val(fat) < 64 * P:
            r = mulc(fat, exact(w))
            assert val(r) % P == val(fat) * w % P
