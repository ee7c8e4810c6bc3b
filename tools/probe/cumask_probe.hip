// cumask_probe.hip -- which XCDs / CUs a stream created with hipExtStreamCreateWithCUMask runs on (MI355X: 8 XCDs x 32 CUs).
// Each workgroup records its XCC id (HW_REG_XCC_ID) and its hardware CU/SE ids (HW_REG_HW_ID); the host prints, per mask,
// the number of workgroups seen per XCD and the number of distinct (XCD, SE, CU) triples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <tuple>

__global__ void census(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
    }
    // keep the workgroup resident a little so that the launch spreads over every enabled CU
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    if (mask.empty()) { if (hipStreamCreate(&st) != hipSuccess) { printf("%s: stream create failed\n", name); return; } }
    else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: masked stream create failed\n", name); return; }
    const int nwg = 2048;
    uint32_t* d; hipMalloc(&d, nwg * 8);
    census<<<nwg, 256, 0, st>>>(d);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(nwg * 2); hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0};
    std::set<std::tuple<int, int, int>> cus;
    for (int i = 0; i < nwg; ++i) {
        const int xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
        const int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc]++;
        cus.insert({xcc, se * 2 + sh, cu});
    }
    printf("%-34s workgroups per XCD:", name);
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("   distinct CUs %zu\n", cus.size());
    hipFree(d); hipStreamDestroy(st);
}

int main() {
    run("no mask", {});
    run("bits 0-127", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0});
    run("bits 128-255", {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
    run("even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u});
    run("bits 0-31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
    run("bits with (i % 8) < 4", {0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu});
    return 0;
}
