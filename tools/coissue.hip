// Diagnostic (not part of the product): do a matrix-phase wave and a vector-phase wave of ONE SIMD overlap?
// 512-thread workgroups (two waves per SIMD), one per CU.  Waves 0-3 run a dependent chain of v_mfma_f32_32x32x16_f16, waves 4-7 a stream of
// independent vector instructions of one kind; each role alone (the other half exits at once) and both together.  If "together" ~ max of the
// two: they overlap; ~ sum: they do not.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/coissue.hip -o tools/coissue.bin && tools/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int MODE>      // MODE 0: MFMA half only, 1: vector half only, 2: both, 3: both halves vector, 4: both halves MFMA
__global__ __launch_bounds__(512) void co(float* out, int iters, const f16x8* src)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = MODE == 4 ? true : MODE == 3 ? false : wave < 4;
    if ((MODE == 0 && !mf) || (MODE == 1 && mf)) return;
    float r = 0.f;
    if (mf) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const f16x8 a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
        unsigned q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float c1 = 1.0001f, c2 = 1e-6f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) {          // 8 x 16 = 128 vector instructions per iteration (an MFMA iteration is 16 x 32 = 512 cycles)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (KIND == 0) v[i] = fmaf(v[i], c1, c2);
                    if (KIND == 1) { const f32x2 t = {v[i], v[(i + 1) & 15]}; q[i & 7] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2)); }
                    if (KIND == 2) { unsigned rr; asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rr) : "v"(q[i & 7]), "v"(v[i])); q[(i + 1) & 7] = rr; }
                    if (KIND == 3) v[i] = v[i] > 0.5f ? v[(i + 1) & 15] : c2;
                    if (KIND == 4) v[i] = fmaxf(v[i], c2) + c1;
                }
            }
        }
        for (int i = 0; i < 16; ++i) r += v[i];
        for (int i = 0; i < 8; ++i) r += (float)q[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int KIND, int MODE> static float run(float* out, const f16x8* src, int iters)
{
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((co<KIND, MODE>), dim3(256), dim3(512), 0, 0, out, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((co<KIND, MODE>), dim3(256), dim3(512), 0, 0, out, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}
template <int KIND> static void kind(const char* name, float* out, const f16x8* src)
{
    const int iters = 2000;
    const float m = run<KIND, 0>(out, src, iters), v = run<KIND, 1>(out, src, iters), both = run<KIND, 2>(out, src, iters), vv = run<KIND, 3>(out, src, iters), mm = run<KIND, 4>(out, src, iters);
    printf("%-28s MFMA half alone %7.1f us | vector half alone %7.1f us (%.1f ns per instruction) | together %7.1f us (max %.1f, sum %.1f) | vector on BOTH halves %7.1f | MFMA on both %7.1f\n",
           name, m, v, v * 1e3 / (iters * 128.0), both, m > v ? m : v, m + v, vv, mm);
}
template <int NV>      // one wave per SIMD: every MFMA followed by NV independent v_fma_f32 of the same wave
__global__ __launch_bounds__(256) void self(float* out, int iters, const f16x8* src)
{
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const f16x8 a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float c1 = 1.0001f, c2 = 1e-6f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 7] = fmaf(v[k & 7], c1, c2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NV> static void selfrun(float* out, const f16x8* src)
{
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((self<NV>), dim3(256), dim3(256), 0, 0, out, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((self<NV>), dim3(256), dim3(256), 0, 0, out, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("one wave per SIMD, %2d v_fma_f32 behind every MFMA: %7.1f us = %.1f ns per MFMA\n", NV, ms * 1e3, ms * 1e6 / (iters * 16.0));
}
int main()
{
    float* out; f16x8* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 2048 * 16);
    static _Float16 h[2048 * 8]; unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    selfrun<0>(out, src); selfrun<2>(out, src); selfrun<4>(out, src); selfrun<6>(out, src); selfrun<8>(out, src); selfrun<12>(out, src); selfrun<16>(out, src);
    kind<0>("v_fma_f32", out, src);
    kind<1>("v_cvt_pk_f16_f32 (+xor)", out, src);
    kind<2>("v_fma_mixlo_f16", out, src);
    kind<3>("v_cmp + v_cndmask", out, src);
    kind<4>("v_max + v_add", out, src);
    return 0;
}
// (second experiment, appended main2): ONE wave per SIMD, its own vector instructions between its MFMAs
