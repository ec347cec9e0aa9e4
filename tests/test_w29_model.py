"""CPU model of csrc/w29.hip.h -- lazily reduced 9 x 29-bit-limb Fr arithmetic whose bounds live in the C++ types -- and of the quotient widget
kernels written on it (csrc/quotient29.hip.h).

On the device the bound RULES (how V, the limb bound and the scale class of a result follow from its operands; which multiple of p and which
limb raise a subtraction uses; when a multiplier column may hold a sum of terms) are static_asserts: an expression that breaks them does not
compile.  What a compiler cannot show is that the rules themselves are right.  This model restates every operation limb by limb with Python
integers, computes the declared bounds with the same formulas, and asserts after EVERY operation that the actual limbs, top limb, value and
column sums stay inside what the type declares, that no limb difference goes negative, and that the value is the field element it should be
(x * 2^256 * 32^class mod p).  It then evaluates the widget kernels' expressions exactly as quotient29.hip.h writes them, on random inputs, on
inputs at the top of the coarse range (2p - 1: what the declared V bounds are priced for) and on limb patterns of all ones, against plain
modular arithmetic.  Test infrastructure only: the product never runs it."""
import random

P = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # BN254 Fr (fr.hpp:12-15)
M29 = (1 << 29) - 1
U32, U64 = 1 << 32, 1 << 64
R = 1 << 256
RP = 1 << 261
INV29 = (-pow(P, -1, 1 << 29)) % (1 << 29)
P_TOP = P >> 232
PTOP1 = P_TOP + 1
INV_TOP = (1 << 32) // (P_TOP + 1)
RP_OVER_P = 169
assert RP // P == RP_OVER_P


def limbs_exact(v):
    assert 0 <= v < (1 << (29 * 8 + 32))
    return [(v >> (29 * j)) & M29 for j in range(8)] + [v >> 232]


P29 = limbs_exact(P)


def val(a):
    return sum(x << (29 * i) for i, x in enumerate(a))


class W:
    """limbs + the type's template arguments (cls, vq, lm) + the field element it stands for (fv, for checking only)"""

    def __init__(self, limbs, cls, vq, lm, fv):
        self.l, self.cls, self.vq, self.lm, self.fv = list(limbs), cls, vq, lm, fv % P
        self.top = (vq * PTOP1 + 63) // 64 + 2
        self.any = max(self.top, lm)
        assert self.top < U32 and lm < U32, "w29: a limb would leave 32 bits (the C++ static_assert)"
        self.check()

    def check(self):
        assert all(0 <= x < U32 for x in self.l)
        assert all(x <= self.lm for x in self.l[:8]), ("limb bound", max(self.l[:8]), self.lm)
        assert self.l[8] <= self.top, ("top limb bound", self.l[8], self.top)
        assert val(self.l) * 64 <= self.vq * P, ("value bound", val(self.l) / P, self.vq / 64)  # (<=: 0 - 0 + M p is exactly M p)
        assert val(self.l) % P == self.fv * R * pow(32, self.cls, P) % P, "wrong value"


def ld(x, cls):
    """x: the words of a device array, a coarse residue x_field * R mod p + {0, p}, as an integer < 2p"""
    assert 0 <= x < 2 * P and cls in (0, 1)
    return W(limbs_exact(x << (5 * cls)), cls, (64 if cls else 2) * 64, M29, x * pow(R, -1, P))


def dot(*terms):
    cls = terms[0][0].cls + terms[0][1].cls - 1
    assert all(a.cls + b.cls - 1 == cls for a, b in terms) and cls in (0, 1), "scale classes (the C++ static_assert)"
    col = sum(9.0 * a.any * b.any for a, b in terms) + 9.0 * 2.0**58 + 2.0**36
    assert col < 2.0**64 * 0.9999, "column bound (the C++ static_assert)"
    vv = sum(a.vq * b.vq for a, b in terms)
    vq = (vv + 64 * RP_OVER_P - 1) // (64 * RP_OVER_P) + 64
    acc, m, r = 0, [0] * 9, [0] * 9
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        for a, b in terms:
            acc += sum(a.l[i] * b.l[k - i] for i in range(lo, hi + 1))
            assert acc < U64
        acc += sum(m[i] * P29[k - i] for i in range(lo, min(k - 1, 8) + 1) if k - i <= 8)
        assert acc < U64
        if k <= 8:
            m[k] = ((acc & 0xFFFFFFFF) * INV29) & M29
            acc += m[k] * P29[0]
            assert acc < U64 and acc & M29 == 0
        else:
            r[k - 9] = acc & M29
        acc >>= 29
    assert acc < U32
    r[8] = acc
    fv = sum(a.fv * b.fv for a, b in terms)
    out = W(r, cls, vq, M29, fv)
    assert val(r) * RP == sum(val(a.l) * val(b.l) for a, b in terms) + val(m) * P
    return out


def mul(a, b):
    return dot((a, b))


def sqr(a):
    assert a.cls == 1 and a.any < (1 << 31)
    assert all(2 * x < U32 for x in a.l)  # the doubled operand of f29_sqr
    return dot((a, a))


def add(a, b):
    assert a.cls == b.cls
    return W([x + y for x, y in zip(a.l, b.l)], a.cls, a.vq + b.vq, a.lm + b.lm, a.fv + b.fv)


def dbl(a):
    return add(a, a)


def sub(a, b):
    assert a.cls == b.cls
    m = (b.vq + 63) // 64 + 1
    e = max(30, (b.lm + 1).bit_length())
    assert e <= 31 and m <= 168 and m * P_TOP >= b.top + (1 << (e - 29)), "w29::sub preconditions (the C++ static_asserts)"
    mp = m * P
    q = [(mp >> (29 * j)) & M29 for j in range(9)]
    assert mp >> 232 < (1 << 29)
    up, down = 1 << e, 1 << (e - 29)
    spread = [q[0] + up] + [q[j] + up - down for j in range(1, 8)] + [q[8] - down]
    assert val(spread) == mp
    out = []
    for j in range(9):
        assert spread[j] >= b.l[j], ("a limb difference went negative", j)
        out.append(a.l[j] + (spread[j] - b.l[j]))
    return W(out, a.cls, a.vq + 64 * m, a.lm + (1 << e) + (1 << 29), a.fv - b.fv)


def neg(b):
    return sub(W([0] * 9, b.cls, 0, 0, 0), b)


def carry(a):
    r = [a.l[0] & M29] + [(a.l[i] & M29) + (a.l[i - 1] >> 29) for i in range(1, 8)] + [a.l[8] + (a.l[7] >> 29)]
    return W(r, a.cls, a.vq, M29 + 8, a.fv)


def up(a):
    assert a.cls == 0 and a.top < (1 << 27)
    r = [(a.l[0] << 5) & M29] + [(((a.l[i] << 5) & 0xFFFFFFFF) & M29) + (a.l[i - 1] >> 24) for i in range(1, 8)]
    r.append((a.l[8] << 5) + (a.l[7] >> 24))
    return W(r, 1, a.vq * 32, M29 + 256, a.fv)


def finish(a):
    """n29_finish: the [0, 2p) words of a class-0 value below 32p"""
    assert a.cls == 0 and a.vq <= 31 * 64
    c = carry(a).l
    q = (c[8] * INV_TOP) >> 32
    assert q < 32
    qp = limbs_exact(q * P)
    t, cy = [], 0
    for i in range(8):
        d = c[i] - qp[i]
        assert -(1 << 30) < d < (1 << 30)
        v = d + cy
        t.append(v & M29)
        cy = v >> 29
    t.append(c[8] - qp[8] + cy)
    assert 0 <= t[8] < (1 << 24)
    out = val(t)
    assert out == val(a.l) - q * P and 0 <= out < 2 * P, ("finish left [0, 2p)", out / P)
    assert out % P == a.fv * R % P
    return out


# ---- the 8 x u32 word arithmetic of field.hip.h the kernels use for linear combinations of loaded values: coarse residues in [0, 2p)
def fe_add(a, b):
    r = a + b
    return r - 2 * P if r >= 2 * P else r


def fe_sub(a, b):
    r = a - b
    return r + 2 * P if r < 0 else r


def x3(v):
    return fe_add(fe_add(v, v), v)


def x4(v):
    d = fe_add(v, v)
    return fe_add(d, d)


def mont(k):
    return k * R % P


def fld(x):
    """the field element a word array entry stands for"""
    return x * pow(R, -1, P) % P


def quad_from(d2, d):
    u = carry(sub(d2, ld(x3(d), 1)))
    return carry(add(sqr(u), dbl(u)))


class Setup:
    """QuotientSetup: every entry a coarse residue (words)"""

    def __init__(self, rnd, alpha_f=None):
        self.alpha_f = rnd.randrange(P) if alpha_f is None else alpha_f  # (one alpha per widget chain; each widget its own alpha_base)
        self.ab_f = rnd.randrange(P)
        co = lambda f: mont(f) + rnd.choice((0, P))  # either representative
        self.alpha = co(self.alpha_f)
        self.alpha2 = co(self.alpha_f**2)
        self.alpha3x2 = co(2 * self.alpha_f**3)
        self.ap = [co(self.ab_f * self.alpha_f**k) for k in range(7)]
        self.ab2 = co(self.ab_f**2)
        self.beta_f, self.gamma_f, self.delta_f = (rnd.randrange(P) for _ in range(3))
        self.beta, self.gamma, self.delta = co(self.beta_f), co(self.gamma_f), co(self.delta_f)
        self.k_f = [rnd.randrange(P) for _ in range(3)]
        self.k = [co(f) for f in self.k_f]
        self.one, self.c2, self.c3, self.c7, self.c17, self.c81, self.c83 = (mont(k) for k in (1, 2, 3, 7, 17, 81, 83))


def arith_part(s, w1, w2, w3, w4, qc, d, d2, sel):
    qa = sel["qarith"]
    w12 = mul(ld(w1, 1), ld(w2, 0))
    u4 = sub(mul(ld(w4, 1), ld(w4, 0)), ld(w4, 0))
    t2 = mul(mul(u4, ld(fe_sub(w4, s.c2), 1)), ld(s.alpha, 1))
    gate = dot((w12, ld(sel["qm"], 1)), (ld(w1, 0), ld(sel["q1"], 1)), (ld(w2, 0), ld(sel["q2"], 1)), (ld(w3, 0), ld(sel["q3"], 1)),
               (ld(w4, 0), ld(sel["q4"], 1)), (t2, ld(sel["q5"], 1)))
    g2 = add(gate, ld(qc, 0))
    d8 = fe_add(x4(d), x4(d))
    lin = fe_sub(fe_add(d8, d), s.c7)
    h1 = mul(sub(ld(lin, 1), dbl(d2)), ld(d, 0))
    qq = sub(sqr(ld(qa, 1)), ld(qa, 1))
    return dot((g2, ld(qa, 1)), (h1, qq))


def range_part(r, w1, w2, w3, w4n, d, d2):
    d2_, d3_, d4_ = fe_sub(w2, x4(w3)), fe_sub(w1, x4(w2)), fe_sub(w4n, x4(w1))
    f1 = quad_from(d2, d)
    f2 = quad_from(sqr(ld(d2_, 1)), d2_)
    f3 = quad_from(sqr(ld(d3_, 1)), d3_)
    f4 = quad_from(sqr(ld(d4_, 1)), d4_)
    return dot((f1, ld(r.ap[0], 0)), (f2, ld(r.ap[1], 0)), (f3, ld(r.ap[2], 0)), (f4, ld(r.ap[3], 0)))


def logic_part(l, w1, w2, w3, w4, w1n, w2n, w4n, qc):
    qa, qb, qcq = fe_sub(w1n, x4(w1)), fe_sub(w2n, x4(w2)), fe_sub(w4n, x4(w4))
    sm = fe_add(qa, qb)
    sum3 = x3(sm)
    sum9 = x3(sum3)
    sum18 = fe_add(sum9, sum9)
    sum81 = fe_add(x4(sum18), sum9)
    c3 = x3(qcq)
    c9 = x3(c3)
    a2, b2 = sqr(ld(qa, 1)), sqr(ld(qb, 1))
    fa, fb = quad_from(a2, qa), quad_from(b2, qb)
    abw = carry(sub(mul(ld(qa, 1), ld(qb, 0)), ld(w3, 0)))
    in1 = fe_add(fe_sub(x4(w3), sum18), l.c81)
    lin2 = fe_sub(l.c83, sum81)
    w3_9 = x3(x3(w3))
    x = mul(ld(w3, 1), ld(in1, 1))
    e = dot((ld(w3, 0), add(x, ld(lin2, 1))), (ld(fe_add(w3_9, w3_9), 0), add(a2, b2)))
    idv = dot((abw, ld(l.alpha3x2, 1)), (fa, ld(l.alpha2, 0)), (fb, ld(l.alpha, 0)), (ld(fe_sub(c9, sum3), 0), ld(qc, 1)))
    tail = carry(sub(add(idv, ld(fe_add(c3, sum3), 0)), dbl(e)))
    return mul(tail, ld(l.ap[0], 1))


def quad_f(D):
    return D * (D - 1) * (D - 2) * (D - 3)


def kernel_arith_range_logic(s, sr, sl, v):
    """k_quotient29_turbo_arith_range_logic<7>; v: dict of coarse words.  Returns (the stored words, the expected field value)."""
    w1, w2, w3, w4, w1n, w2n, w4n, qc = (v[k] for k in ("w1", "w2", "w3", "w4", "w1n", "w2n", "w4n", "qc"))
    d = fe_sub(w3, x4(w4))
    d2 = sqr(ld(d, 1))
    inner = arith_part(s, w1, w2, w3, w4, qc, d, d2, v)
    rng = range_part(sr, w1, w2, w3, w4n, d, d2)
    lgc = logic_part(sl, w1, w2, w3, w4, w1n, w2n, w4n, qc)
    total = dot((inner, ld(s.ap[0], 1)), (rng, ld(v["qrange"], 1)), (lgc, ld(v["qlogic"], 1)))
    out = finish(add(total, ld(v["quot"], 0)))
    f = {k: fld(x) for k, x in v.items()}
    F1, F2, F3, F4 = f["w1"], f["w2"], f["w3"], f["w4"]
    al = s.alpha_f
    gate = f["qm"] * F1 * F2 + f["q1"] * F1 + f["q2"] * F2 + f["q3"] * F3 + f["q4"] * F4 + f["qc"] + al * f["q5"] * F4 * (F4 - 1) * (F4 - 2)
    dd = F3 - 4 * F4
    ar = s.ab_f * (f["qarith"] * gate + (f["qarith"] ** 2 - f["qarith"]) * dd * (9 * dd - 2 * dd * dd - 7))
    rg = f["qrange"] * sum(sr.ab_f * sr.alpha_f**k * quad_f(D) for k, D in enumerate((dd, F2 - 4 * F3, F1 - 4 * F2, f["w4n"] - 4 * F1)))
    a_, b_, c_ = f["w1n"] - 4 * F1, f["w2n"] - 4 * F2, f["w4n"] - 4 * F4
    E = F3 * (F3 * (4 * F3 - 18 * (a_ + b_) + 81) + 18 * (a_ * a_ + b_ * b_) - 81 * (a_ + b_) + 83)
    la = sl.alpha_f
    lg = f["qlogic"] * sl.ab_f * (2 * (a_ * b_ - F3) * la**3 + quad_f(a_) * la**2 + quad_f(b_) * la + 3 * (a_ + b_ + c_) - 2 * E
                                  + f["qc"] * (9 * c_ - 3 * (a_ + b_)))
    return out, (f["quot"] + ar + rg + lg) % P


def kernel_fixed_base_linear(s, v):
    w4, w3n, w1, w3, qc = v["w4"], v["w3n"], v["w1"], v["w3"], v["qc"]
    delta = fe_sub(v["w4n"], x4(w4))
    dsq = mul(ld(delta, 1), ld(delta, 0))
    q1t = mul(dsq, ld(v["q1"], 1))
    dw = mul(ld(delta, 1), ld(w3n, 0))
    w2x2 = fe_add(v["w2"], v["w2"])
    y = dot((ld(fe_sub(v["w1n"], w1), 0), ld(s.ap[3], 1)), (ld(w2x2, 0), ld(s.ap[2], 1)))
    t3 = mul(dw, up(y))
    sel = dot((ld(s.ap[5], 0), ld(v["q4"], 1)), (ld(s.ap[6], 0), ld(v["qm"], 1)))
    i5 = mul(ld(fe_sub(s.one, w4), 0), ld(s.ap[5], 1))
    init = dot((sel, ld(w3, 1)), (i5, ld(v["q5"], 1)))
    lin = dot((q1t, ld(s.ap[1], 1)), (ld(s.ap[1], 0), ld(v["q2"], 1)), (t3, ld(v["q3"], 1)), (init, ld(qc, 1)))
    out = finish(add(mul(lin, ld(v["qecc"], 1)), ld(v["quot"], 0)))
    f = {k: fld(x) for k, x in v.items()}
    ap = [s.ab_f * s.alpha_f**k for k in range(7)]
    dl = f["w4n"] - 4 * f["w4"]
    want = f["qecc"] * (f["q1"] * ap[1] * dl * dl + f["q2"] * ap[1] + f["q3"] * dl * f["w3n"] * (ap[3] * (f["w1n"] - f["w1"]) + 2 * ap[2] * f["w2"])
                        + f["qc"] * (f["w3"] * (ap[5] * f["q4"] + ap[6] * f["qm"]) + ap[5] * (1 - f["w4"]) * f["q5"]))
    return out, (f["quot"] + want) % P


def kernel_fixed_base_gate(s, v):
    w1, w2, w3, w4, w1n, w3n, qc, qe = (v[k] for k in ("w1", "w2", "w3", "w4", "w1n", "w3n", "qc", "qecc"))
    delta = fe_sub(v["w4n"], x4(w4))
    dsq = sqr(ld(delta, 1))
    acc = mul(carry(sub(dsq, ld(s.one, 1))), carry(sub(dsq, ld(x3(s.c3), 1))))
    dx = fe_sub(w3n, w1)
    dx2 = sqr(ld(dx, 1))
    xa2 = sqr(ld(w3n, 1))
    dy = mul(ld(delta, 1), ld(w2, 0))
    nw3n, nw2 = carry(neg(ld(w3n, 0))), carry(neg(ld(w2, 0)))
    qe2 = fe_add(qe, qe)
    xid = dot((ld(fe_add(fe_add(w1n, w1), w3n), 0), dx2), (nw3n, xa2), (nw2, ld(w2, 1)), (dy, ld(qe2, 1)))
    xid17 = add(xid, ld(s.c17, 0))
    qd = mul(ld(qe, 1), ld(delta, 0))
    ym = carry(sub(ld(w2, 0), qd))
    yid = dot((ld(fe_add(v["w2n"], w2), 0), ld(dx, 1)), (ym, ld(fe_sub(w1, w1n), 1)))
    w4m1 = fe_sub(w4, s.one)
    i1 = mul(ld(w4m1, 1), ld(fe_sub(w4m1, w3), 0))
    i2 = mul(ld(w1, 1), ld(w3, 0))
    i3 = dot((ld(fe_sub(s.one, w4), 0), ld(qc, 1)), (nw2, ld(w3, 1)))
    init = dot((i1, ld(s.ap[4], 1)), (neg(i2), ld(s.ap[5], 1)), (i3, ld(s.ap[6], 1)))
    gate = dot((acc, ld(s.ap[0], 0)), (nw3n, ld(s.ap[1], 1)), (xid17, ld(s.ap[2], 1)), (yid, ld(s.ap[3], 1)), (init, ld(qc, 1)))
    out = finish(add(mul(carry(gate), ld(qe, 1)), ld(v["quot"], 0)))
    f = {k: fld(x) for k, x in v.items()}
    ap = [s.ab_f * s.alpha_f**k for k in range(7)]
    dl = f["w4n"] - 4 * f["w4"]
    accf = (dl + 1) * (dl + 3) * (dl - 1) * (dl - 3)
    dxf = f["w3n"] - f["w1"]
    xf = (f["w1n"] + f["w1"] + f["w3n"]) * dxf * dxf - (f["w3n"] ** 3 + f["w2"] ** 2 - 17) + 2 * dl * f["w2"] * f["qecc"]
    yf = (f["w2n"] + f["w2"]) * dxf + (f["w1"] - f["w1n"]) * (f["w2"] - f["qecc"] * dl)
    initf = (f["w4"] - 1) * (f["w4"] - 1 - f["w3"]) * ap[4] - f["w1"] * f["w3"] * ap[5] + ((1 - f["w4"]) * f["qc"] - f["w2"] * f["w3"]) * ap[6]
    want = f["qecc"] * (accf * ap[0] - f["w3n"] * ap[1] + xf * ap[2] + yf * ap[3] + initf * f["qc"])
    return out, (f["quot"] + want) % P


def kernel_permutation(s, v, rb, width):
    """one point of k_quotient29_permutation<width>; rb = beta g w^i as coarse words"""
    rb1 = ld(rb, 1)
    wg1 = fe_add(v["w1"], s.gamma)
    num = add(ld(wg1, 0), ld(rb, 0))
    den = add(ld(wg1, 0), mul(ld(v["s1"], 0), ld(s.beta, 1)))
    for j, (wk, sk) in enumerate((("w2", "s2"), ("w3", "s3"), ("w4", "s4"))[: width - 1]):
        wg = ld(fe_add(v[wk], s.gamma), 1)
        num = mul(num, add(wg, mul(rb1, ld(s.k[j], 1))))
        den = mul(den, add(wg, mul(ld(v[sk], 1), ld(s.beta, 1))))
    z, zw = v["z"], v["zw"]
    t1 = mul(ld(fe_sub(zw, s.delta), 0), ld(s.ap[0], 1))
    t2 = mul(ld(fe_sub(z, s.one), 0), ld(s.ab2, 1))
    inn = dot((num, ld(z, 1)), (neg(den), ld(zw, 1)), (t1, ld(v["l_end"], 1)), (t2, ld(v["l1"], 1)))
    out = finish(mul(inn, ld(s.ap[0], 1)))
    f = {k: fld(x) for k, x in v.items()}
    rbf = fld(rb)
    ks = [1] + s.k_f
    n_ = f["z"]
    d_ = f["zw"]
    for j in range(width):
        wv = f["w%d" % (j + 1)]
        n_ *= wv + s.gamma_f + ks[j] * rbf
        d_ *= wv + s.gamma_f + s.beta_f * f["s%d" % (j + 1)]
    want = s.ab_f * (n_ - d_ + (f["zw"] - s.delta_f) * s.ab_f * f["l_end"] + (f["z"] - 1) * s.ab_f**2 * f["l1"])
    return out, want % P


NAMES = ("w1", "w2", "w3", "w4", "w1n", "w2n", "w3n", "w4n", "qc", "qm", "q1", "q2", "q3", "q4", "q5", "qarith", "qecc", "qrange", "qlogic", "quot",
         "z", "zw", "s1", "s2", "s3", "s4", "l1", "l_end")


def inputs(rnd, mode):
    if mode == "random":
        return {k: rnd.randrange(2 * P) for k in NAMES}
    if mode == "top":  # the top of the coarse range: what the V bounds are priced for
        return {k: 2 * P - 1 - rnd.randrange(3) for k in NAMES}
    if mode == "ones":  # limbs of all ones wherever the value allows
        return {k: min(2 * P - 1, (1 << rnd.choice((254, 253, 232, 200, 116, 58, 29))) - 1) for k in NAMES}
    if mode == "small":  # base-4 digit rows: the identities' own zero sets, and zeros
        return {k: mont(rnd.randrange(4)) + rnd.choice((0, P)) for k in NAMES}
    raise ValueError(mode)


def check(out_words, want_field):
    assert 0 <= out_words < 2 * P and fld(out_words) == want_field


def run(kernel, seeds=6):
    for mode in ("random", "top", "ones", "small"):
        for seed in range(seeds):
            rnd = random.Random(1000 * seed + len(mode))
            kernel(rnd, inputs(rnd, mode))


def test_operation_bounds_at_their_limits():
    """every operation on operands pushed to the bounds their types declare"""
    rnd = random.Random(7)
    top = 2 * P - 1
    a1, a0 = ld(top, 1), ld(top, 0)
    p1 = mul(a1, a0)          # class 0
    q1 = sqr(a1)              # class 1
    s1 = sub(q1, a1)          # class 1, lm 2^31
    c1 = carry(s1)
    u1 = up(p1)
    for w in (p1, q1, s1, c1, u1, add(a1, a1), neg(a0), dbl(dbl(a0))):
        w.check()
    # six-term dot with exact operands: the largest sum the column bound admits
    six = dot(*[(ld(top - rnd.randrange(9), 0), ld(top - rnd.randrange(9), 1)) for _ in range(6)])
    assert six.cls == 0
    # finish at the table's last rows
    big = add(six, ld(top, 0))
    for _ in range(3):
        big = add(big, ld(top, 0))
    check(finish(big), big.fv)
    # a seven-term dot of exact operands must be refused
    try:
        dot(*[(ld(top, 0), ld(top, 1)) for _ in range(7)])
    except AssertionError:
        pass
    else:
        raise AssertionError("a seven-term column was accepted")


def test_fused_arithmetic_range_logic_kernel():
    def k(rnd, v):
        s = Setup(rnd)
        check(*kernel_arith_range_logic(s, Setup(rnd, s.alpha_f), Setup(rnd, s.alpha_f), v))
    run(k)


def test_fixed_base_kernels():
    run(lambda rnd, v: check(*kernel_fixed_base_linear(Setup(rnd), v)))
    run(lambda rnd, v: check(*kernel_fixed_base_gate(Setup(rnd), v)))


def test_permutation_kernels():
    for width in (4, 3):
        run(lambda rnd, v: check(*kernel_permutation(Setup(rnd), v, rnd.randrange(2 * P), width)), seeds=4)
