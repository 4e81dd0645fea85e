// Diagnostic (not part of the product): do vector instructions hide in the gaps of a wave's MFMA stream on gfx950 -- IN CYCLES.
// tools/coissue.hip answered in microseconds on a chip that moves its clock with the instruction mix; this one stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop of every wave and prints cycles per MFMA, the in-kernel clock and ns.
//   experiment A  one stream per wave: [MFMA, n x v_fma_f32] x 16 per iteration, hand-placed (MFMA builtin + sched_barrier(0), fillers asm volatile: the order is the written one, no s_nop),
//                 4 independent accumulators, one or two waves per SIMD, both MFMA shapes (32x32x16: 32-cycle gap, 16x16x32: 16-cycle gap)
//   experiment B  two waves per SIMD with different roles: waves 0-3 MFMA only, waves 4-7 v_fma_f32 only; each alone, then together
// MI355X_MICROARCH.md (cycle table): an MFMA holds the SIMD's vector issue for 8 of its 32 (16) cycles, a 4-cycle filler costs 4; <= 5 fillers per
// 32x32x16 gap hide (32.4 cycles per MFMA with exactly 5), past that each costs its 4.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/coissue_cycles.hip -o tools/coissue_cycles.bin && tools/coissue_cycles.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, rt; };

// SHAPE 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_16x16x32_f16.  NV fillers behind every MFMA.  ROLE: 0 every wave runs the mixed stream;
// 1 waves 0-3 MFMA only and waves 4-7 exit; 2 waves 4-7 fillers only (8 per "gap") and waves 0-3 exit; 3 both roles side by side.
template <int SHAPE, int NV, int ROLE>
__global__ __launch_bounds__(512) void co(float* out, Stamp* st, int iters, const f16x8* src)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = ROLE == 0 || (ROLE != 4 && wave < 4);      // ROLE 4: fillers on both halves; ROLE 5: waves 0-3 [MFMA + NV fillers], waves 4-7 fillers only
    if ((ROLE == 1 && !mf) || (ROLE == 2 && mf)) return;
    f32x16 A0, A1, A2, A3;
    f32x4 B0, B1, B2, B3;
    for (int i = 0; i < 16; ++i) { A0[i] = A1[i] = A2[i] = A3[i] = 0.f; }
    for (int i = 0; i < 4; ++i) { B0[i] = B1[i] = B2[i] = B3[i] = 0.f; }
    const f16x8 a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
    const float c1 = 1.0001f, c2 = 1e-6f;
    typedef float f32x2q __attribute__((ext_vector_type(2)));
    f32x2q pp = {v0, v1}, pq = {c1, c1};
#define FILL(k) do { if ((k) == 0) FINS(v0); if ((k) == 1) FINS(v1); if ((k) == 2) FINS(v2); if ((k) == 3) FINS(v3); if ((k) == 4) FINS(v4); if ((k) == 5) FINS(v5); if ((k) == 6) FINS(v6); if ((k) == 7) FINS(v7); } while (0)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#ifndef FK
#define FK 0
#endif
#if FK == 1      /* v_fma_mixlo_f16: the second H3 piece */
#define FINS(v) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v) : "v"(c1))
#elif FK == 2    /* v_cvt_pk_f16_f32 */
#define FINS(v) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v) : "v"(c1))
#elif FK == 3    /* v_max_f32 */
#define FINS(v) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(c2))
#elif FK == 4    /* v_pk_mul_f32 on a register pair */
#define FINS(v) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pp) : "v"(pq))
#elif FK == 5    /* v_cmp + v_cndmask (a gate) */
#define FINS(v) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(v) : "v"(c1) : "vcc")
#else
#define FINS(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2))
#endif
#define FILL_OLD(k) do { \
        if ((k) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(c1), "v"(c2)); \
        if ((k) == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(c1), "v"(c2)); \
        if ((k) == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(c1), "v"(c2)); \
        if ((k) == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(c1), "v"(c2)); \
        if ((k) == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v4) : "v"(c1), "v"(c2)); \
        if ((k) == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v5) : "v"(c1), "v"(c2)); \
        if ((k) == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v6) : "v"(c1), "v"(c2)); \
        if ((k) == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v7) : "v"(c1), "v"(c2)); } while (0)
    if (mf) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (SHAPE == 0) {
                    if ((u & 3) == 0) { A0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, A0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 1) { A1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, A1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 2) { A2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, A2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 3) { A3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, A3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                } else {
                    if ((u & 3) == 0) { B0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, B0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 1) { B1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, B1, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 2) { B2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, B2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                    if ((u & 3) == 3) { B3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, B3, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                }
                if (ROLE == 0 || ROLE == 5) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) FILL(k & 7);
                }
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int k = 0; k < (ROLE == 5 ? 8 : NV); ++k) FILL(k & 7);
            }
        }
    }
#undef FILL
#undef FILL_OLD
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + pp[0] + pp[1];
    for (int i = 0; i < 16; ++i) r += A0[i] + A1[i] + A2[i] + A3[i];
    for (int i = 0; i < 4; ++i) r += B0[i] + B1[i] + B2[i] + B3[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { st[blockIdx.x * 8 + wave].cyc = t1 - t0; st[blockIdx.x * 8 + wave].rt = r1 - r0; }
}

struct Res { double cyc_m, cyc_v, ghz, us; };       // median wave cycles of the MFMA role / the filler role, in-kernel clock, kernel time
template <int SHAPE, int NV, int ROLE> static Res run(float* out, Stamp* st, const f16x8* src, int threads)
{
    const int iters = 1500;
    std::vector<Stamp> h(256 * 8);
    hipMemset(st, 0, sizeof(Stamp) * 256 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((co<SHAPE, NV, ROLE>), dim3(256), dim3(threads), 0, 0, out, st, iters, src);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((co<SHAPE, NV, ROLE>), dim3(256), dim3(threads), 0, 0, out, st, iters, src);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), st, sizeof(Stamp) * 256 * 8, hipMemcpyDeviceToHost);
    std::vector<double> cm, cv, clk;
    for (int g = 0; g < 256; ++g)
        for (int w = 0; w < threads / 64; ++w) {
            const Stamp& s = h[g * 8 + w];
            if (!s.cyc) continue;
            const bool mf = ROLE == 0 || (ROLE != 4 && w < 4);
            (mf ? cm : cv).push_back((double)s.cyc / (iters * 16.0));
            if (s.rt) clk.push_back((double)s.cyc / ((double)s.rt * 10.0) );      // cycles per ns: s_memrealtime ticks at 100 MHz
        }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    Res r; r.cyc_m = med(cm); r.cyc_v = med(cv); r.ghz = med(clk); r.us = ms * 1e3;
    return r;
}

template <int SHAPE, int NV> static void rowA(float* out, Stamp* st, const f16x8* src)
{
    const Res one = run<SHAPE, NV, 0>(out, st, src, 256), two = run<SHAPE, NV, 0>(out, st, src, 512);
    const int gap = SHAPE == 0 ? 32 : 16;
    printf("%-9s | %d fillers per gap | 1 wave/SIMD: %6.2f cycles per MFMA (%+5.1f %% over %d) at %.2f GHz = %5.2f ns | 2 waves/SIMD: %6.2f cycles per MFMA of the SIMD at %.2f GHz = %5.2f ns\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", NV, one.cyc_m, 100.0 * (one.cyc_m / gap - 1.0), gap, one.ghz, one.cyc_m / one.ghz,
           two.cyc_m / 2.0, two.ghz, two.cyc_m / 2.0 / two.ghz);
}
template <int SHAPE> static void rowsB(float* out, Stamp* st, const f16x8* src)
{
    // fillers-only role: 8 per "gap" slot, 16 slots per iteration
    const Res m = run<SHAPE, 8, 1>(out, st, src, 512), v = run<SHAPE, 8, 2>(out, st, src, 512), both = run<SHAPE, 8, 3>(out, st, src, 512);
    printf("%-9s | roles on the two waves of a SIMD | MFMA wave alone %6.2f cycles per MFMA (%.2f GHz) | filler wave alone %5.2f cycles per v_fma_f32 (%.2f GHz) | together: MFMA wave %6.2f cycles per MFMA, "
           "filler wave %5.2f cycles per v_fma_f32 (%.2f GHz); kernel %.0f / %.0f / %.0f us\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", m.cyc_m, m.ghz, v.cyc_v / 8.0, v.ghz, both.cyc_m, both.cyc_v / 8.0, both.ghz, m.us, v.us, both.us);
}
template <int SHAPE, int NV> static void rowC(float* out, Stamp* st, const f16x8* src)
{
    const Res r = run<SHAPE, NV, 5>(out, st, src, 512);
    printf("%-9s | waves 0-3: [MFMA + %d fillers], waves 4-7: fillers only | MFMA wave %6.2f cycles per MFMA (its %d fillers included), filler wave %5.2f cycles per v_fma_f32 = %.1f of its instructions per MFMA of the partner (%.2f GHz)\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", NV, r.cyc_m, NV, r.cyc_v / 8.0, r.cyc_m / (r.cyc_v / 8.0), r.ghz);
}
int main()
{
    float* out; f16x8* src; Stamp* st;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 2048 * 16); hipMalloc(&st, sizeof(Stamp) * 256 * 8);
    static _Float16 h[2048 * 8]; unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    printf("# experiment A: one stream per wave, [MFMA, n x v_fma_f32] hand-placed, 4 independent accumulators, random operands, every CU busy\n");
    rowA<0, 0>(out, st, src); rowA<0, 2>(out, st, src); rowA<0, 4>(out, st, src); rowA<0, 5>(out, st, src); rowA<0, 6>(out, st, src); rowA<0, 8>(out, st, src); rowA<0, 12>(out, st, src);
    rowA<1, 0>(out, st, src); rowA<1, 1>(out, st, src); rowA<1, 2>(out, st, src); rowA<1, 3>(out, st, src); rowA<1, 4>(out, st, src); rowA<1, 6>(out, st, src);
    printf("# experiment B: waves 0-3 MFMA only, waves 4-7 v_fma_f32 only (one of each per SIMD)\n");
    rowsB<0>(out, st, src); rowsB<1>(out, st, src);
    {
        const Res one = run<0, 8, 2>(out, st, src, 512), two = run<0, 8, 4>(out, st, src, 512);
        printf("# v_fma_f32 only: one wave per SIMD %5.2f cycles per instruction (%.2f GHz); two waves per SIMD %5.2f cycles per instruction of EACH wave = %5.2f per instruction of the SIMD (%.2f GHz)\n",
               one.cyc_v / 8.0, one.ghz, two.cyc_v / 8.0, two.cyc_v / 16.0, two.ghz);
    }
    printf("# experiment C: a matrix wave that carries its own fillers beside a pure vector wave\n");
    rowC<0, 0>(out, st, src); rowC<0, 2>(out, st, src); rowC<0, 4>(out, st, src); rowC<0, 6>(out, st, src); rowC<1, 0>(out, st, src); rowC<1, 2>(out, st, src);
    return 0;
}
