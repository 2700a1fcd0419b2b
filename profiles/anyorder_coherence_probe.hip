// Does a hipExtAnyOrderLaunch kernel see what its predecessor in the stream wrote from another XCD?  (anyorder_probe.hip:
// such a launch still starts after its predecessor has ended, with a 1-2 us shorter boundary -- if what it skips is the
// cache maintenance between the two kernels, a dependent kernel reads stale lines of its own XCD's L2.)
//   W(it): workgroup b writes `it` to its 4 KB region            (ordinary launch)
//   R(it): workgroup b checks the region of workgroup (37 b + 11) mod n, written on another XCD   (ordinary | any-order)
//   hipcc -O3 --offload-arch=gfx950 profiles/anyorder_coherence_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kWords = 1024;
__global__ void write_kernel(int* buf, int val) {
    int* p = buf + (size_t)blockIdx.x * kWords;
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) p[i] = val;
}
__global__ void read_kernel(const int* buf, int expect, unsigned* err, int* sink) {
    const int from = (blockIdx.x * 37 + 11) % gridDim.x;
    const int* p = buf + (size_t)from * kWords;
    int bad = 0, acc = 0;
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) { const int v = p[i]; bad += v != expect; acc += v; }
    if (bad) atomicAdd(err, (unsigned)bad);
    if (acc == 0x7fffffff) sink[0] = acc;
}
int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int nwg = 1024;
    int *buf, *sink; unsigned* err;
    CHECK(hipMalloc(&buf, (size_t)nwg * kWords * 4)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&err, 4));
    for (int mode = 0; mode < 2; mode++) {
        CHECK(hipMemset(buf, 0, (size_t)nwg * kWords * 4)); CHECK(hipMemset(err, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, st));
        const int iters = 5000;
        for (int it = 1; it <= iters; it++) {
            hipLaunchKernelGGL(write_kernel, dim3(nwg), dim3(256), 0, st, buf, it);
            hipExtLaunchKernelGGL(read_kernel, dim3(nwg), dim3(256), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0,
                                  (const int*)buf, it, err, sink);
        }
        CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
        unsigned h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"reader_launch\": \"%s\", \"iterations\": %d, \"stale_words\": %u, \"us_per_pair\": %.2f}\n",
               mode ? "hipExtAnyOrderLaunch" : "ordinary", iters, h, ms * 1e3 / iters);
    }
    return 0;
}
