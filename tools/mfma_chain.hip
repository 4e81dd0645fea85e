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
// v_mfma_f32_16x16x32_f16: half the FLOP per instruction (16 384), 16 instead of 32 cycles -- equal FLOP per cycle; MI355X_MICROARCH.md
// (DVFS give-back, item 7) reports the 16x16x32 loop holding a higher clock on random data.  Same chain, NACC accumulators of 4 registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int DATA>
__global__ __launch_bounds__(512) void chain16(float* out, unsigned long long* cyc, int iters, const f16x8* src)
{
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;
    f16x8 x, y;
    if (DATA == 0) { for (int i = 0; i < 8; ++i) { x[i] = (_Float16)0.f; y[i] = (_Float16)0.f; } }
    else { x = src[threadIdx.x & 63]; y = src[64 + (threadIdx.x & 63)]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[u % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int DATA> static void run16(const char* name, int threads, float* out, unsigned long long* cyc, const f16x8* src)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((chain16<NACC, DATA>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((chain16<NACC, DATA>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i];
    c /= 256;
    const double nmfma = 256.0 * (threads / 64) * iters * 32.0;
    printf("%-40s threads %3d: %6.1f ticks per MFMA per wave; %.1f us, %.0f TFLOP/s, implied clock %.2f GHz\n", name, threads, c / (iters * 32.0),
           ms * 1e3, nmfma * 16384.0 / (ms * 1e-3) / 1e12, c / (ms * 1e3) / 1e3);
}
// the same chain, but every MFMA takes a DIFFERENT operand pair (NOPS random pairs held in registers, rotating): operand buses toggle as in a real kernel
template <int NOPS>
__global__ __launch_bounds__(512) void chain_ops(float* out, unsigned long long* cyc, int iters, const f16x8* src)
{
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f16x8 a[NOPS], b[NOPS];
    for (int k = 0; k < NOPS; ++k) { a[k] = src[(k * 2) * 64 + (threadIdx.x & 63)]; b[k] = src[(k * 2 + 1) * 64 + (threadIdx.x & 63)]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u % NOPS], b[(u * 5 + 1) % NOPS], acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NOPS> static void run_ops(const char* name, int threads, float* out, unsigned long long* cyc, const f16x8* src)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((chain_ops<NOPS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((chain_ops<NOPS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double nmfma = 256.0 * (threads / 64) * iters * 16.0;
    printf("%-40s threads %3d: %.1f us, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, threads, ms * 1e3, nmfma * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (nmfma / 1024.0));
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
    const double nmfma = 256.0 * (threads / 64) * iters * 16.0;
    printf("%-40s threads %3d: %6.1f ticks per MFMA per wave, %6.1f per MFMA per SIMD; %.1f us, %.0f TFLOP/s, implied clock %.2f GHz\n", name, threads, per_wave,
           per_wave / (threads / 256.0), ms * 1e3, nmfma * 32768.0 / (ms * 1e-3) / 1e12, c / (ms * 1e3) / 1e3);
}
int main()
{
    float* out; unsigned long long* cyc; f16x8* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&src, 2048 * 16);
    static _Float16 h[2048 * 8]; unsigned s = 12345u;
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
    for (int threads : {256, 512}) {
        run16<1, 0>("f16 16x16x32 1 acc zeros", threads, out, cyc, src);
        run16<1, 1>("f16 16x16x32 1 acc random", threads, out, cyc, src);
        run16<4, 1>("f16 16x16x32 4 acc random", threads, out, cyc, src);
        run16<16, 1>("f16 16x16x32 16 acc random", threads, out, cyc, src);
    }
    for (int threads : {256, 512}) {
        run_ops<1>("f16 1 operand pair", threads, out, cyc, src);
        run_ops<4>("f16 4 rotating operand pairs", threads, out, cyc, src);
        run_ops<8>("f16 8 rotating operand pairs", threads, out, cyc, src);
        run_ops<16>("f16 16 rotating operand pairs", threads, out, cyc, src);
    }
    return 0;
}
