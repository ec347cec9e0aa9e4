// Per-domain device constants shared by ntt.hip and poly.hip.
#pragma once
#include "field.hip.h"

namespace bbg {

// evaluation_domain's scalar members (reference polynomials/evaluation_domain.hpp:40-52) plus power tables
struct DomainConsts {
    Fr root, root_inv, n_inv, gen, gen_inv;
    Fr pow2_root[32];     // root^(2^b)
    Fr pow2_root_inv[32]; // root_inv^(2^b)
    Fr pow2_tmp[32];      // scratch table for arbitrary bases (generator shift paths)
    Fr constant;          // staged caller constant (ops 4..7)
    Fr gk;                // running generator of the split coset FFT
};

// base^e from a table of base^(2^b)
__device__ __forceinline__ Fr pow_from_table(const Fr* __restrict__ pow2, uint64_t e)
{
    Fr acc = Fr::one();
    int b = 0;
    while (e) {
        if (e & 1) acc = fe_mul(acc, pow2[b]);
        e >>= 1;
        b++;
    }
    return acc;
}

} // namespace bbg
