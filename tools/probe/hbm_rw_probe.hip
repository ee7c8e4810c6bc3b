// HBM read / write / copy rates of one MI355X with 16-byte lane accesses (what the GEMM epilogues issue): is a write-heavy kernel
// (FFN-1 + GELU: 25 MB read, 201 MB written) bound by the same 8 TB/s figure as a read-heavy one?
// Build: hipcc -O3 --offload-arch=gfx950 hbm_rw_probe.hip -o hbm_rw_probe ; run: ./hbm_rw_probe [MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f4;
template <int MODE>   // 0 read, 1 write, 2 write nontemporal, 3 copy, 4 copy nt-store, 5: read 1 : write 8 (FFN-1's mix), nt stores
__global__ __launch_bounds__(256) void k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n, float* sink) {
    f4 acc = {0, 0, 0, 0};
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) acc += src[i];
        else if (MODE == 1) dst[i] = v;
        else if (MODE == 2) __builtin_nontemporal_store(v, dst + i);
        else if (MODE == 3) dst[i] = src[i];
        else if (MODE == 4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else { f4 a = (i & 7) == 0 ? __builtin_nontemporal_load(src + (i >> 3)) : v; __builtin_nontemporal_store(a, dst + i); }
    }
    if (MODE == 0 && acc[0] == 12345.678f) sink[0] = acc[1];
}
template <int MODE> void run(const char* name, const f4* s, f4* d, size_t n, float* sink, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192}) {
        for (int i = 0; i < 3; ++i) k<MODE><<<grid, 256>>>(s, d, n, sink);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) k<MODE><<<grid, 256>>>(s, d, n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %5d: %8.1f us  %7.2f TB/s\n", name, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    }
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 512;
    const size_t n = mb * 1024 * 1024 / 16;
    f4 *s, *d; float* sink;
    hipMalloc(&s, n * 16); hipMalloc(&d, n * 16); hipMalloc(&sink, 64);
    hipMemset(s, 1, n * 16); hipMemset(d, 0, n * 16);
    const double b = (double)n * 16;
    run<0>("read", s, d, n, sink, b);
    run<1>("write", s, d, n, sink, b);
    run<2>("write nontemporal", s, d, n, sink, b);
    run<3>("copy (r + w bytes)", s, d, n, sink, 2 * b);
    run<4>("copy nt (r + w bytes)", s, d, n, sink, 2 * b);
    run<5>("read 1 : write 8, nt", s, d, n, sink, b * 1.125);
    return 0;
}
