// Host cost of replaying a step as a hipGraph whose kernel-node parameters are patched every step
// (hipGraphExecKernelNodeSetParams x N + one hipGraphLaunch) against N plain kernel launches.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_params profiles/graph_params/graph_params.hip && /tmp/graph_params
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void small(float* p, int n, float a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * a + 1.0f;
}
int main() {
    const int N = 33, steps = 2000;
    float* buf;
    CK(hipMalloc(&buf, 1 << 22));
    hipStream_t st, aux;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    using clk = std::chrono::steady_clock;
    // ---- A: plain launches, 22 on st, 11 on aux with 3 fork/join pairs (12 event operations)
    auto plain = [&](int s) {
        for (int k = 0; k < N; k++) {
            const bool on_aux = (k % 3) == 2;
            if (k % 11 == 2) { hipEventRecord(ef, st); hipStreamWaitEvent(aux, ef, 0); }
            hipLaunchKernelGGL(small, dim3(32 + (s & 3)), dim3(256), 0, on_aux ? aux : st, buf + k * 16384, 8192 + s, 1.0f);
            if (k % 11 == 10) { hipEventRecord(ej, aux); hipStreamWaitEvent(st, ej, 0); }
        }
    };
    for (int s = 0; s < 50; s++) plain(s);
    CK(hipDeviceSynchronize());
    auto t0 = clk::now();
    for (int s = 0; s < steps; s++) plain(s);
    auto t1 = clk::now();
    CK(hipDeviceSynchronize());
    auto t2 = clk::now();
    printf("plain launches: host %.1f us per step (%d kernels + 12 event ops), incl. drain %.1f us\n",
           std::chrono::duration<double>(t1 - t0).count() / steps * 1e6, N,
           std::chrono::duration<double>(t2 - t0).count() / steps * 1e6);
    // ---- B: the same step captured once, parameters patched per step
    hipGraph_t graph;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    plain(0);
    CK(hipStreamEndCapture(st, &graph));
    size_t nn = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(graph, nodes.data(), &nn));
    std::vector<hipGraphNode_t> knodes;
    for (auto n : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(n, &t)); if (t == hipGraphNodeTypeKernel) knodes.push_back(n); }
    printf("captured graph: %zu nodes, %zu kernel nodes\n", nn, knodes.size());
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::vector<hipKernelNodeParams> kp(knodes.size());
    for (size_t k = 0; k < knodes.size(); k++) CK(hipGraphKernelNodeGetParams(knodes[k], &kp[k]));
    float* pv[64]; int nv[64]; float av[64]; void* argv_[64][3];
    auto patched = [&](int s) -> int {
        for (size_t k = 0; k < knodes.size(); k++) {
            pv[k] = buf + k * 16384; nv[k] = 8192 + s; av[k] = 1.0f;
            argv_[k][0] = &pv[k]; argv_[k][1] = &nv[k]; argv_[k][2] = &av[k];
            hipKernelNodeParams p = kp[k];
            p.gridDim = dim3(32 + (s & 3));
            p.kernelParams = argv_[k];
            CK(hipGraphExecKernelNodeSetParams(exec, knodes[k], &p));
        }
        CK(hipGraphLaunch(exec, st));
        return 0;
    };
    for (int s = 0; s < 50; s++) if (patched(s)) return 1;
    CK(hipDeviceSynchronize());
    t0 = clk::now();
    for (int s = 0; s < steps; s++) if (patched(s)) return 1;
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    printf("patched graph:  host %.1f us per step (%zu SetParams + 1 launch), incl. drain %.1f us\n",
           std::chrono::duration<double>(t1 - t0).count() / steps * 1e6, knodes.size(),
           std::chrono::duration<double>(t2 - t0).count() / steps * 1e6);
    // ---- C: replay without patching (floor)
    t0 = clk::now();
    for (int s = 0; s < steps; s++) CK(hipGraphLaunch(exec, st));
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    printf("replay only:    host %.1f us per step, incl. drain %.1f us\n",
           std::chrono::duration<double>(t1 - t0).count() / steps * 1e6,
           std::chrono::duration<double>(t2 - t0).count() / steps * 1e6);
    return 0;
}
