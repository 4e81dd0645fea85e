// Diagnostic (not part of the product): what 2 816 workgroups' atomicMax calls on 128 per-sample slots cost, by the distance between the slots.
//   hipcc -O3 --offload-arch=gfx950 tools/atomic_probe.hip -o tools/atomic_probe.bin && tools/atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool GUARD>
__global__ __launch_bounds__(256) void commit(unsigned* slots, int stride_words, int per_wave, unsigned salt)
{
    // every workgroup (or wave) reports a value for sample blockIdx.y; values rise with blockIdx.x so that every call is live
    const unsigned v = 0x3f800000u + blockIdx.x * 16 + (threadIdx.x >> 6) + salt;
    if (per_wave ? (threadIdx.x & 63) == 0 : threadIdx.x == 0) {
        unsigned* s = slots + (long)blockIdx.y * stride_words;
        if (!GUARD || v > *reinterpret_cast<volatile unsigned*>(s)) atomicMax(s, v);
    }
}
int main()
{
    unsigned* slots; hipMalloc(&slots, 128 * 4096 * 4); hipMemset(slots, 0, 128 * 4096 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int guard = 1; guard >= 0; --guard)
    for (int per_wave = 0; per_wave < 2; ++per_wave)
        for (int stride : {1, 16, 32, 64, 256, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipMemsetAsync(slots, 0, 128 * 4096 * 4, 0);
                hipEventRecord(a, 0);
                if (guard) hipLaunchKernelGGL(commit<true>, dim3(22, 128), dim3(256), 0, 0, slots, stride, per_wave, (unsigned)rep);
                else hipLaunchKernelGGL(commit<false>, dim3(22, 128), dim3(256), 0, 0, slots, stride, per_wave, (unsigned)rep);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep >= 3 && ms < best) best = ms;
            }
            printf("%s, %s, slots %5d bytes apart: %.1f us\n", guard ? "read first" : "atomic only", per_wave ? "one call per wave (11 264)" : "one call per workgroup (2 816)", stride * 4, best * 1e3f);
        }
    return 0;
}
