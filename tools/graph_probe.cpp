// Launch-latency probe: a small call's shape -- one staged upload, three short dependent kernels, one 9 KB read-back, one synchronisation -- enqueued
// directly against replayed as a hipGraph (captured once).  hipcc --offload-arch=gfx950 -O2 tools/graph_probe.cpp -o gpurun_out/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_spin(unsigned *p, int iters) {
    unsigned v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    p[threadIdx.x & 63] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned *d, *h_in, *h_out; hipMalloc(&d, 1 << 16); hipHostMalloc(&h_in, 4096); hipHostMalloc(&h_out, 16384);
    const int iters = 3000;                                   // ~10 us per kernel for a single wave
    auto body = [&]() {
        hipMemcpyAsync(d, h_in, 192, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, d, 100);
        hipLaunchKernelGGL(k_spin, dim3(4), dim3(256), 0, st, d, iters);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, d, iters);
        hipMemcpyAsync(h_out, d, 9024, hipMemcpyDeviceToHost, st);
    };
    for (int i = 0; i < 50; i++) { body(); hipStreamSynchronize(st); }
    std::vector<double> a, b, c;
    for (int i = 0; i < 500; i++) { double t0 = now_us(); body(); double t1 = now_us(); hipStreamSynchronize(st); a.push_back(now_us() - t0); c.push_back(t1 - t0); }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); body(); hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 50; i++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    std::vector<double> e;
    for (int i = 0; i < 500; i++) { double t0 = now_us(); hipGraphLaunch(ge, st); double t1 = now_us(); hipStreamSynchronize(st); b.push_back(now_us() - t0); e.push_back(t1 - t0); }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("direct: call %.1f us (enqueue %.1f)   graph replay: call %.1f us (launch %.1f)\n", med(a), med(c), med(b), med(e));
    // kernel-only time for reference
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); body(); hipEventRecord(e1, st); hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, e0, e1); printf("GPU span direct %.1f us\n", ms * 1e3);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); printf("GPU span graph %.1f us\n", ms * 1e3);
    // (r6) the shape of a MID-SIZE call (mid.hip): seven dependent kernels of 5 .. 60 us with machine-filling grids on one stream, no copies (the record is published by the
    // last kernel), the kernel arguments of every call different (a replay must patch them: hipGraphExecKernelNodeSetParams per node)
    struct shape { int blocks, iters; };
    const shape mid[7] = {{64, 1500}, {400, 3000}, {1, 1500}, {1400, 9000}, {700, 6000}, {24, 5000}, {1, 300}};
    int iters_v[7];
    auto body_mid = [&](int salt) {
        for (int k = 0; k < 7; k++) { iters_v[k] = mid[k].iters + (salt & 1); hipLaunchKernelGGL(k_spin, dim3(mid[k].blocks), dim3(256), 0, st, d, iters_v[k]); }
    };
    for (int i = 0; i < 50; i++) { body_mid(i); hipStreamSynchronize(st); }
    a.clear(); b.clear(); c.clear(); e.clear();
    for (int i = 0; i < 500; i++) { double t0 = now_us(); body_mid(i); double t1 = now_us(); hipStreamSynchronize(st); a.push_back(now_us() - t0); c.push_back(t1 - t0); }
    hipGraph_t g2; hipGraphExec_t ge2;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); body_mid(0); hipStreamEndCapture(st, &g2);
    hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
    size_t nn = 0; hipGraphGetNodes(g2, nullptr, &nn);
    std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g2, nodes.data(), &nn);
    for (int i = 0; i < 50; i++) { hipGraphLaunch(ge2, st); hipStreamSynchronize(st); }
    for (int i = 0; i < 500; i++) { double t0 = now_us(); hipGraphLaunch(ge2, st); double t1 = now_us(); hipStreamSynchronize(st); b.push_back(now_us() - t0); e.push_back(t1 - t0); }
    printf("mid shape (7 kernels), direct: call %.1f us (enqueue %.1f)   graph replay, same arguments: call %.1f us (launch %.1f)\n", med(a), med(c), med(b), med(e));
    // ... and with every node's arguments patched before the replay (what a real call needs: other pointers, other n)
    std::vector<double> f, f2;
    for (int i = 0; i < 500; i++) {
        double t0 = now_us();
        for (size_t k = 0; k < nn && k < 7; k++) {
            hipKernelNodeParams kp;
            if (hipGraphKernelNodeGetParams(nodes[k], &kp) != hipSuccess) continue;
            int it = mid[k].iters + (i & 1);
            void *args[2] = {(void *)&d, (void *)&it};
            kp.kernelParams = args;
            hipGraphExecKernelNodeSetParams(ge2, nodes[k], &kp);
        }
        hipGraphLaunch(ge2, st); double t1 = now_us(); hipStreamSynchronize(st); f.push_back(now_us() - t0); f2.push_back(t1 - t0);
    }
    printf("mid shape, graph replay with seven patched nodes: call %.1f us (patch + launch %.1f)\n", med(f), med(f2));
    hipEventRecord(e0, st); body_mid(0); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); printf("mid shape GPU span direct %.1f us\n", ms * 1e3);
    hipEventRecord(e0, st); hipGraphLaunch(ge2, st); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms, e0, e1); printf("mid shape GPU span graph %.1f us\n", ms * 1e3);
    return 0;
}
