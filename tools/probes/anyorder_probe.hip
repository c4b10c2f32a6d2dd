// anyorder_probe.hip -- does hipExtAnyOrderLaunch (an AQL packet without the barrier bit) let two kernels of ONE stream overlap on gfx950?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel. Idea it would serve (round 6): a GEMM's
// remaining-rows launch enqueued FIRST (ordered behind the producer), the main launch behind it WITHOUT the barrier bit -- it may start once the
// small launch has been dispatched, i.e. after the producer completed, and the next kernel (barrier bit set) waits for both. No events, no second queue.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o ab/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    int* sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long c100us = 10000;  // wall_clock64 ticks at 100 MHz: 100 us
    auto run = [&](int flags, int blocks_a, int blocks_b) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(blocks_a), dim3(64), 0, s, c100us, sink);
            hipExtLaunchKernelGGL(spin, dim3(blocks_b), dim3(64), 0, s, nullptr, nullptr, flags, c100us, sink);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best * 1e3f;
    };
    printf("{\"two 100 us kernels on one stream, 8 + 8 blocks\": {\"ordered_us\": %.1f, \"second_any_order_us\": %.1f},\n", run(0, 8, 8), run(hipExtAnyOrderLaunch, 8, 8));
    printf(" \"8 + 512 blocks\": {\"ordered_us\": %.1f, \"second_any_order_us\": %.1f}}\n", run(0, 8, 512), run(hipExtAnyOrderLaunch, 8, 512));
    return 0;
}
