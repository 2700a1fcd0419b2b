// Launch-boundary probe for gfx950 (round 3): what does a DEPENDENT phase boundary cost -- as a kernel launch in a
// stream (what the step program does 18 times per training step) against a grid-wide barrier inside ONE persistent
// kernel (what a device-side interpreter of the step program would do)?
//
// Every phase does the same small piece of dependent work: each workgroup reads 16 KB written by ANOTHER workgroup in
// the previous phase (so the boundary must really order memory), adds 1 and writes 16 KB for the next phase.
//
//   launches     one kernel per phase, same stream (in-order)
//   barrier      one kernel, phases separated by a sense-reversing grid barrier: one atomic add (device scope) per
//                workgroup, then a spin on the generation word with s_sleep
//   barrier_xcd  the same with a two-level barrier: per-XCD counter (workgroup b -> XCD b mod 8) + one arrival per XCD
//
// Output: JSON lines {probe, wgs_per_cu, phases, us_per_phase}.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 profiles/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 256, kFloatsPerWg = 4096;          // 16 KB per workgroup and phase

__device__ __forceinline__ void work(const float* __restrict__ src, float* __restrict__ dst, int wg, int nwg) {
    const int from = (wg * 37 + 11) % nwg;                  // another workgroup's output of the previous phase
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)from * kFloatsPerWg);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)wg * kFloatsPerWg);
    for (int i = threadIdx.x; i < kFloatsPerWg / 4; i += kThreads) {
        float4 v = s[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        d[i] = v;
    }
}

__global__ __launch_bounds__(kThreads) void phase_kernel(const float* src, float* dst) { work(src, dst, blockIdx.x, gridDim.x); }

struct Bar { unsigned count; unsigned gen; unsigned pad[30]; unsigned xcount[8 * 32]; };

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned nwg, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                            // release this workgroup's writes (device scope)
        const unsigned prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nwg - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    my_gen++;
    __syncthreads();
}

__device__ __forceinline__ void grid_barrier_xcd(Bar* b, unsigned nwg, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned x = blockIdx.x & 7u, per = (nwg + 7u - x) / 8u;          // workgroups of this XCD
        const unsigned prev = __hip_atomic_fetch_add(&b->xcount[x * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        bool last = false;
        if (prev == per - 1) {
            __hip_atomic_store(&b->xcount[x * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned p2 = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 == 7u) {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                last = true;
            }
        }
        if (!last)
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    my_gen++;
    __syncthreads();
}

template <int KIND>
__global__ __launch_bounds__(kThreads) void persistent_kernel(float* a, float* bbuf, Bar* bar, int phases) {
    unsigned my_gen = 0;
    if (threadIdx.x == 0) my_gen = __hip_atomic_load(&bar->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    my_gen = __builtin_amdgcn_readfirstlane(my_gen);
    // every thread needs the generation thread 0 read: broadcast through LDS
    __shared__ unsigned g0;
    if (threadIdx.x == 0) g0 = my_gen;
    __syncthreads();
    my_gen = g0;
    for (int p = 0; p < phases; p++) {
        work((p & 1) ? bbuf : a, (p & 1) ? a : bbuf, blockIdx.x, gridDim.x);
        if (KIND == 0) grid_barrier(bar, gridDim.x, my_gen);
        else grid_barrier_xcd(bar, gridDim.x, my_gen);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int phases = 64;
    for (int per_cu : {1, 2, 4}) {
        const int nwg = cus * per_cu;
        float *a, *b;
        Bar* bar;
        CHECK(hipMalloc(&a, (size_t)nwg * kFloatsPerWg * 4));
        CHECK(hipMalloc(&b, (size_t)nwg * kFloatsPerWg * 4));
        CHECK(hipMalloc(&bar, sizeof(Bar)));
        CHECK(hipMemset(a, 0, (size_t)nwg * kFloatsPerWg * 4));
        CHECK(hipMemset(b, 0, (size_t)nwg * kFloatsPerWg * 4));
        CHECK(hipMemset(bar, 0, sizeof(Bar)));
        int occ = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<0>, kThreads, 0));
        if (occ < per_cu) { fprintf(stderr, "not co-resident at %d per CU\n", per_cu); continue; }
        for (int probe = 0; probe < 3; probe++) {
            float best = 1e30f;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipMemsetAsync(a, 0, (size_t)nwg * kFloatsPerWg * 4, st));
                CHECK(hipEventRecord(e0, st));
                if (probe == 0) {
                    for (int p = 0; p < phases; p++)
                        hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(kThreads), 0, st, (p & 1) ? b : a, (p & 1) ? a : b);
                } else if (probe == 1) {
                    hipLaunchKernelGGL(persistent_kernel<0>, dim3(nwg), dim3(kThreads), 0, st, a, b, bar, phases);
                } else {
                    hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(kThreads), 0, st, a, b, bar, phases);
                }
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            // the dependency chain really ran: every element was incremented once per phase
            std::vector<float> h((size_t)nwg * kFloatsPerWg);
            CHECK(hipMemcpy(h.data(), (phases & 1) ? b : a, h.size() * 4, hipMemcpyDeviceToHost));
            bool ok = true;
            for (float v : h) ok = ok && v == (float)phases;
            printf("{\"probe\": \"%s\", \"wgs_per_cu\": %d, \"phases\": %d, \"us_per_phase\": %.3f, \"check\": %s}\n",
                   probe == 0 ? "launches" : probe == 1 ? "barrier" : "barrier_xcd", per_cu, phases, best * 1e3f / phases,
                   ok ? "true" : "false");
        }
        CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(bar));
    }
    return 0;
}
