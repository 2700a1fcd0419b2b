// Which compute units does a stream created with hipExtStreamCreateWithCUMask run on?  (Can the fabric-bound residual
// sweep and the issue-bound LDS sweep share the chip, each on its own CUs?)  Launches a spinning kernel on a masked
// stream and prints, per XCC, how many distinct (SE, CU) pairs its workgroups saw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, hipStream_t st) {
    const int nb = 4096;
    uint32_t* d;
    CHECK(hipMalloc(&d, nb * 8));
    hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, st, d, 200);
    CHECK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(2 * nb);
    CHECK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus[16];
    for (int b = 0; b < nb; b++) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    printf("{\"stream\": \"%s\", \"cus_per_xcc\": [", name);
    int tot = 0;
    for (int x = 0; x < 8; x++) { printf("%s%zu", x ? ", " : "", cus[x].size()); tot += (int)cus[x].size(); }
    printf("], \"total\": %d}\n", tot);
    CHECK(hipFree(d));
}

int main() {
    hipStream_t s0;
    CHECK(hipStreamCreate(&s0));
    run("unmasked", s0);
    for (int variant = 0; variant < 4; variant++) {
        uint32_t mask[8];
        // variant 0: the low 24 bits of every 32-bit word; 1: words 0-5 full, 6-7 empty; 2: every fourth bit cleared; 3: words 6-7 only
        for (int w = 0; w < 8; w++) mask[w] = variant == 0 ? 0x00ffffffu : variant == 1 ? (w < 6 ? 0xffffffffu : 0u) : variant == 2 ? 0x77777777u : (w >= 6 ? 0xffffffffu : 0u);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("{\"variant\": %d, \"error\": \"%s\"}\n", variant, hipGetErrorString(e)); continue; }
        char name[64];
        snprintf(name, sizeof name, "mask variant %d", variant);
        run(name, s);
        CHECK(hipStreamDestroy(s));
    }
    return 0;
}
