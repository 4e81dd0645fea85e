// Diagnostic (not part of the product): cycles per v_mfma_f32_32x32x16_{f16,bf16} in a dependent chain on 1, 2 and 4 accumulators, 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mfma_chain.hip -o tools/mfma_chain.bin && tools/mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool BF, int DATA>
__global__ __launch_bounds__(512) void chain(float* out, unsigned long long* cyc, int iters, const f16x8* src)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    f16x8 x, y; bf16x8 xb, yb;
    if (DATA == 0) { for (int i = 0; i < 8; ++i) { x[i] = (_Float16)0.f; y[i] = (_Float16)0.f; xb[i] = (__bf16)0.f; yb[i] = (__bf16)0.f; } }
    else { x = src[threadIdx.x & 63]; y = src[64 + (threadIdx.x & 63)]; for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(float)x[i]; yb[i] = (__bf16)(float)y[i]; } }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (BF) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[u % NACC], 0, 0, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u % NACC], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, bool BF, int DATA> static void run(const char* name, int threads, float* out, unsigned long long* cyc, const f16x8* src)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((chain<NACC, BF, DATA>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((chain<NACC, BF, DATA>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256;
    const double per_wave = c / (iters * 16.0);
    printf("%-40s threads %3d: %6.1f ticks per MFMA per wave, %6.1f per MFMA per SIMD; %.1f us, implied clock %.2f GHz\n", name, threads, per_wave,
           per_wave / (threads / 256.0), ms * 1e3, c / (ms * 1e3) / 1e3);
}
int main()
{
    float* out; unsigned long long* cyc; f16x8* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&src, 128 * 16);
    _Float16 h[128 * 8]; unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<1, false, 0>("f16 1 acc zeros", threads, out, cyc, src);
        run<1, false, 1>("f16 1 acc random", threads, out, cyc, src);
        run<2, false, 1>("f16 2 acc random", threads, out, cyc, src);
        run<4, false, 1>("f16 4 acc random", threads, out, cyc, src);
        run<1, true, 1>("bf16 1 acc random", threads, out, cyc, src);
        run<4, true, 1>("bf16 4 acc random", threads, out, cyc, src);
    }
    return 0;
}
