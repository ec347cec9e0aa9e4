// Issue cost of the instructions the BN254 multipliers are made of, on gfx950: straight-line asm blocks, many waves per SIMD, timed with
// HIP events.  Reports cycles per wave-instruction per SIMD at an assumed 2.4 GHz and relative to v_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 8192;
#define R8(X) X X X X X X X X

template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    uint32_t x = seed * 2654435761u + tid, y = seed ^ (tid * 40503u), c = 0, d = 1, e = 2, f = 3;
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) // 8 independent v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        if (OP == 1) // 8 (mad, addc) pairs on ONE accumulator chain each (the shipping multiplier's shape)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %10, vcc, 0, %10, vcc\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"
                         "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %12, vcc, 0, %12, vcc\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %13, vcc, 0, %13, vcc\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_addc_co_u32 %10, vcc, 0, %10, vcc\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"
                         "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_addc_co_u32 %12, vcc, 0, %12, vcc\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\tv_addc_co_u32 %13, vcc, 0, %13, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(x), "+v"(y), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "vcc");
        if (OP == 2) // 16 v_add_u32
            asm volatile(R8("v_add_u32 %0, %1, %0\n\tv_add_u32 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 3) // 16 v_addc_co_u32
            asm volatile(R8("v_addc_co_u32 %0, vcc, %1, %0, vcc\n\tv_addc_co_u32 %2, vcc, %3, %2, vcc\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "vcc");
        if (OP == 4) // 16 v_mul_lo_u32
            asm volatile(R8("v_mul_lo_u32 %0, %1, %0\n\tv_mul_lo_u32 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 5) // 16 v_mul_hi_u32
            asm volatile(R8("v_mul_hi_u32 %0, %1, %0\n\tv_mul_hi_u32 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 6) // 16 v_mad_u32_u24
            asm volatile(R8("v_mad_u32_u24 %0, %1, %0, %0\n\tv_mad_u32_u24 %2, %3, %2, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 7) // 8 v_lshrrev_b64
            asm volatile("v_lshrrev_b64 %0, 29, %0\n\tv_lshrrev_b64 %1, 29, %1\n\tv_lshrrev_b64 %2, 29, %2\n\tv_lshrrev_b64 %3, 29, %3\n\t"
                         "v_lshrrev_b64 %4, 29, %4\n\tv_lshrrev_b64 %5, 29, %5\n\tv_lshrrev_b64 %6, 29, %6\n\tv_lshrrev_b64 %7, 29, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 8) // 8 v_lshl_add_u64
            asm volatile("v_lshl_add_u64 %0, %1, 0, %0\n\tv_lshl_add_u64 %1, %2, 0, %1\n\tv_lshl_add_u64 %2, %3, 0, %2\n\tv_lshl_add_u64 %3, %4, 0, %3\n\t"
                         "v_lshl_add_u64 %4, %5, 0, %4\n\tv_lshl_add_u64 %5, %6, 0, %5\n\tv_lshl_add_u64 %6, %7, 0, %6\n\tv_lshl_add_u64 %7, %0, 0, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 9) // 8 v_mad_u64_u32 with an SGPR factor
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "s"(seed) : "vcc");
        if (OP == 10) // 8 v_mad_u64_u32 with a ZERO 64-bit addend written to a fresh register (start of a column)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, 0\n\tv_mad_u64_u32 %1, vcc, %8, %9, 0\n\tv_mad_u64_u32 %2, vcc, %8, %9, 0\n\tv_mad_u64_u32 %3, vcc, %8, %9, 0\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, 0\n\tv_mad_u64_u32 %5, vcc, %8, %9, 0\n\tv_mad_u64_u32 %6, vcc, %8, %9, 0\n\tv_mad_u64_u32 %7, vcc, %8, %9, 0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        if (OP == 11) // 8 v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %8, %0\n\tv_fma_f64 %1, %1, %8, %1\n\tv_fma_f64 %2, %2, %8, %2\n\tv_fma_f64 %3, %3, %8, %3\n\t"
                         "v_fma_f64 %4, %4, %8, %4\n\tv_fma_f64 %5, %5, %8, %5\n\tv_fma_f64 %6, %6, %8, %6\n\tv_fma_f64 %7, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
        if (OP == 12) // 16 v_and_b32
            asm volatile(R8("v_and_b32 %0, %1, %0\n\tv_and_b32 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 13) // 16 v_mul_u32_u24
            asm volatile(R8("v_mul_u32_u24 %0, %1, %0\n\tv_mul_u32_u24 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 14) // 16 v_mul_hi_u32_u24
            asm volatile(R8("v_mul_hi_u32_u24 %0, %1, %0\n\tv_mul_hi_u32_u24 %2, %3, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 15) // 16 v_lshrrev_b32
            asm volatile(R8("v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %2, 5, %2\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        if (OP == 16) // 16 v_cndmask_b32
            asm volatile(R8("v_cndmask_b32 %0, %1, %0, vcc\n\tv_cndmask_b32 %2, %3, %2, vcc\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "vcc");
        if (OP == 18) { // 16 v_cndmask_b32 under an SGPR-pair mask (the form the compiler emits for selects)
            const unsigned long long msk = __ballot((tid & 3) == 1);
            asm volatile(R8("v_cndmask_b32_e64 %0, %1, %0, %4\n\tv_cndmask_b32_e64 %2, %3, %2, %4\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(msk));
        }
        if (OP == 19) { // 16 v_cndmask_b32 alternating two destinations (no back-to-back dependence)
            const unsigned long long msk = __ballot((tid & 3) == 1);
            asm volatile(R8("v_cndmask_b32_e64 %0, %1, %3, %4\n\tv_cndmask_b32_e64 %2, %3, %1, %4\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(msk));
        }
        if (OP == 20) // 8 v_mad_u64_u32, an s_nop 0 after each (what hipcc puts behind every asm statement)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\ts_nop 0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\ts_nop 0\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\ts_nop 0\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\ts_nop 0\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\ts_nop 0\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\ts_nop 0\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\ts_nop 0\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\ts_nop 0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        if (OP == 21) // 8 v_mad_u64_u32 in a DEPENDENT chain on one accumulator (a column of the 29-bit multiplier)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %0, vcc, %9, %8, %0\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %0, vcc, %9, %8, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %0, vcc, %9, %8, %0\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %0, vcc, %9, %8, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        if (OP == 17) // 16 v_alignbit_b32
            asm volatile(R8("v_alignbit_b32 %0, %1, %0, 29\n\tv_alignbit_b32 %2, %3, %2, 29\n\t") : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    }
    out[tid] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) ^ c ^ d ^ e ^ f;
}

template <int OP> void run(const char* name, int per_iter, uint32_t* out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256; // 4 waves per block = 1 per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<OP><<<blocks, threads>>>(out, 12345u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        k<OP><<<blocks, threads>>>(out, 12345u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double wave_instr_per_simd = (double)waves_per_simd * ITERS * per_iter;
    const double cycles = best * 1e-3 * 2.4e9;
    printf("%-46s %d waves/SIMD  %8.3f ms  %6.2f cycles per wave-instruction (at 2.4 GHz)\n", name, waves_per_simd, best, cycles / wave_instr_per_simd);
}

int main()
{
    uint32_t* out;
    CK(hipMalloc(&out, 256 * 16 * 256 * 4));
    for (int w : { 3, 8 }) {
        run<2>("v_add_u32", 16, out, w);
        run<12>("v_and_b32", 16, out, w);
        run<3>("v_addc_co_u32", 16, out, w);
        run<0>("v_mad_u64_u32 (VGPR factors)", 8, out, w);
        run<9>("v_mad_u64_u32 (SGPR factor)", 8, out, w);
        run<10>("v_mad_u64_u32 (addend 0)", 8, out, w);
        run<1>("v_mad_u64_u32 + v_addc_co_u32 pairs (per instr)", 16, out, w);
        run<4>("v_mul_lo_u32", 16, out, w);
        run<5>("v_mul_hi_u32", 16, out, w);
        run<6>("v_mad_u32_u24", 16, out, w);
        run<13>("v_mul_u32_u24", 16, out, w);
        run<14>("v_mul_hi_u32_u24", 16, out, w);
        run<7>("v_lshrrev_b64", 8, out, w);
        run<8>("v_lshl_add_u64", 8, out, w);
        run<11>("v_fma_f64", 8, out, w);
        run<15>("v_lshrrev_b32", 16, out, w);
        run<17>("v_alignbit_b32", 16, out, w);
        run<20>("v_mad_u64_u32 + s_nop 0 (per mad)", 8, out, w);
        run<21>("v_mad_u64_u32 dependent chain", 8, out, w);
        run<18>("v_cndmask_b32 (SGPR mask)", 16, out, w);
        run<19>("v_cndmask_b32 (SGPR mask, independent)", 16, out, w);
    }
    return 0;
}
