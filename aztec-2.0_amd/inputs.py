"""Deterministic synthetic inputs: the reference's RNG is unseeded
(numeric/random/engine.cpp:122-149), so tests and bench feed BOTH the oracle and the GPU from this generator."""
import numpy as np

_MASK60 = np.uint64(0x0FFFFFFFFFFFFFFF)


def splitmix64_limbs(seed, count, offset=0):
    """Words [offset, offset+count) of the splitmix64 stream started at `seed` (index-based: word i uses state
    seed + (i+1)*gamma, so any slice can be generated on its own -- ranks generate only their shard)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1 + offset, count + offset + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synthetic_scalars(seed, n, start=0):
    """Elements [start, start+n) of the stream: n x 4 limbs, top limb masked to 60 bits (value < 2^252 < r), read as
    Montgomery-form Fr residues."""
    a = splitmix64_limbs(seed, 4 * n, 4 * start).reshape(n, 4).copy()
    a[:, 3] &= _MASK60
    return a


def synthetic_scalars_strided(seed, count, first, stride):
    """Elements first, first + stride, ... (count of them) of the same stream as synthetic_scalars(seed, .): the residue class a rank
    of the sharded NTT owns, generated without materialising the whole array."""
    with np.errstate(over="ignore"):
        elem = np.uint64(first) + np.uint64(stride) * np.arange(count, dtype=np.uint64)
        idx = (np.uint64(4) * elem)[:, None] + np.arange(1, 5, dtype=np.uint64)[None, :]
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        a = z ^ (z >> np.uint64(31))
    a[:, 3] &= _MASK60
    return a


_FR_MODULUS = (0x43E1F593F0000001, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029)


def fr_reduce_once(a):
    """Canonical representative of coarsely reduced Fr residues (values in [0, 2r), 4 little-endian u64 limbs): subtract r where
    a >= r -- what the reference's tests compare on (field::reduce_once).  Host-side numpy, used by the self-checking bench legs."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    ge = np.ones(a.shape[0], dtype=bool)       # a >= r so far (all higher limbs equal)
    decided = np.zeros(a.shape[0], dtype=bool)
    for k in (3, 2, 1, 0):
        m = np.uint64(_FR_MODULUS[k])
        gt, lt = (a[:, k] > m) & ~decided, (a[:, k] < m) & ~decided
        ge[lt] = False
        decided |= gt | lt
    out = a.copy()
    borrow = np.zeros(a.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for k in range(4):
            m = np.uint64(_FR_MODULUS[k])
            d = a[:, k] - m
            b1 = a[:, k] < m
            d2 = d - borrow
            b2 = d < borrow
            out[:, k] = np.where(ge, d2, a[:, k])
            borrow = (b1 | b2).astype(np.uint64)
    return out


def mixed_scalars(seed, n, to_montgomery):
    """The distribution of pippenger_short_inputs (scalar_multiplication.test.cpp:733-753): a quarter each of
    full-width, zero, 64-bit and <= 3-bit scalars.  `to_montgomery` converts plain integers (n,4) to Montgomery."""
    plain = synthetic_scalars(seed, n)
    q = np.arange(n) % 4
    plain[q == 1] = 0
    plain[q == 2, 1:] = 0
    plain[q == 3, 1:] = 0
    plain[q == 3, 0] &= np.uint64(7)
    return to_montgomery(plain)
