// bench_micro/h2d_paths.hip -- how a 32 MiB host array (a wire of a 2^20-gate proof) reaches the device, and what each way costs on this host
// (round 5, DESIGN "Leads 6"): pageable hipMemcpyAsync (what round 1 of the resident prover did), pinned DMA, hipHostRegister + DMA,
// and a staging ring: T host threads copy interleaved chunks into pinned slots and queue each slot's DMA themselves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench_micro/h2d_paths bench_micro/h2d_paths.hip -lpthread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)32 << 20;
    void* d = nullptr;
    CK(hipMalloc(&d, bytes));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto fresh = [&]() { // a page-touched pageable array, like a witness polynomial the composer has just filled
        char* p = (char*)aligned_alloc(4096, bytes);
        for (size_t i = 0; i < bytes; i += 64) p[i] = (char)i;
        return p;
    };
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] * 1e3; };
    char* h = fresh();
    std::vector<double> t;
    for (int r = 0; r < 9; r++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        double t1 = now();
        CK(hipStreamSynchronize(st));
        t.push_back(now() - t0);
        if (r == 8) printf("pageable hipMemcpyAsync: call returns after %.3f ms\n", (t1 - t0) * 1e3);
    }
    printf("pageable 32 MiB H2D            %.3f ms (%.1f GB/s)\n", med(t), bytes / med(t) / 1e6);
    void* pin = nullptr;
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    memcpy(pin, h, bytes);
    t.clear();
    for (int r = 0; r < 9; r++) {
        double t0 = now();
        CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t.push_back(now() - t0);
    }
    printf("pinned 32 MiB H2D              %.3f ms (%.1f GB/s)\n", med(t), bytes / med(t) / 1e6);
    t.clear();
    std::vector<double> tu;
    for (int r = 0; r < 5; r++) {
        char* f = fresh();
        double t0 = now();
        CK(hipHostRegister(f, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, f, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipHostUnregister(f));
        tu.push_back(now() - t2);
        t.push_back(t1 - t0);
        if (r == 4) printf("registered copy %.3f ms\n", (t2 - t1) * 1e3);
        free(f);
    }
    printf("hipHostRegister 32 MiB         %.3f ms, unregister %.3f ms\n", med(t), med(tu));
    for (int T : { 1, 2, 4, 8, 16 }) {
        for (size_t chunk : { (size_t)1 << 20, (size_t)4 << 20 }) {
            const int slots_per_thread = 2;
            std::vector<void*> slot(T * slots_per_thread);
            std::vector<hipEvent_t> ev(T * slots_per_thread);
            std::vector<hipStream_t> sts(T);
            for (auto& s : slot) CK(hipHostMalloc(&s, chunk, hipHostMallocDefault));
            for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& s : sts) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            t.clear();
            for (int r = 0; r < 7; r++) {
                char* f = r == 0 ? h : h; // same source: the page cache state of a just-written witness
                double t0 = now();
                std::vector<std::thread> th;
                const size_t nchunks = bytes / chunk;
                for (int k = 0; k < T; k++)
                    th.emplace_back([&, k]() {
                        CK(hipSetDevice(0));
                        int use = 0;
                        for (size_t c = k; c < nchunks; c += T, use++) {
                            const int s = k * slots_per_thread + (use % slots_per_thread);
                            if (use >= slots_per_thread) CK(hipEventSynchronize(ev[s]));
                            memcpy(slot[s], f + c * chunk, chunk);
                            CK(hipMemcpyAsync((char*)d + c * chunk, slot[s], chunk, hipMemcpyHostToDevice, sts[k]));
                            CK(hipEventRecord(ev[s], sts[k]));
                        }
                    });
                for (auto& x : th) x.join();
                for (auto& s : sts) CK(hipStreamSynchronize(s));
                t.push_back(now() - t0);
            }
            printf("staging ring T=%2d chunk %zu MiB   %.3f ms (%.1f GB/s)  [incl. thread start]\n", T, chunk >> 20, med(t), bytes / med(t) / 1e6);
            for (auto& s : slot) CK(hipHostFree(s));
            for (auto& e : ev) CK(hipEventDestroy(e));
            for (auto& s : sts) CK(hipStreamDestroy(s));
        }
    }
    return 0;
}
