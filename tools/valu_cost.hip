// Diagnostic (not part of the product): issue cost IN CYCLES of the vector instructions the split-operand kernels are made of, one wave per SIMD
// (256-thread workgroups, one per CU) and two waves per SIMD (512 threads), every CU busy; s_memtime around a loop of 16 x 8 independent copies.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -w tools/valu_cost.hip -o tools/valu_cost.bin && tools/valu_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(512) void cost(float* out, unsigned long long* cyc, int iters)
{
    float v[8], w[8];
    f32x2 p[8], q[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; w[i] = v[i] * 0.5f + 1.f; p[i] = (f32x2){v[i], w[i]}; q[i] = (f32x2){w[i], v[i]}; u[i] = threadIdx.x * 2654435761u + i; }
    const float c1 = 1.0001f, c2 = 1e-6f;
    __shared__ unsigned lds[16384];
    const unsigned la = (threadIdx.x & 63) * 16, la8 = (threadIdx.x & 63) * 8, la4 = (threadIdx.x & 63) * 4, la2 = (threadIdx.x & 63) * 2;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v f4v = {v[0], v[1], v[2], v[3]};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#define K0(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
#define K1(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
#define K2(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(w[i]));
#define K3(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(v[i]));
#define K4(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
#define K5(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
#define K6(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q[i]));
#define K7(i) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(v[i]) : "v"(w[i]) : "vcc");
#define K8(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(u[i] & 1));
#define K9(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define K10(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define K11(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(v[i]) :: );
#define K12(i) asm volatile("v_cmp_lt_f32 vcc, 0, %0" :: "v"(w[i]) : "vcc");
#define K13(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(v[i]));
#define K14(i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));          /* a dependent pair */
#define K15(i) asm volatile("v_cmp_lt_f32 s[20:21], 0, %1\n\tv_cmp_lt_f32 s[22:23], 0, %2\n\tv_mul_f32 %0, %0, %3\n\tv_mul_f32 %4, %4, %3\n\tv_cndmask_b32 %0, 0, %0, s[20:21]\n\tv_cndmask_b32 %4, 0, %4, s[22:23]" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]), "v"(c1), "v"(w[(i + 2) & 7]) : "s20", "s21", "s22", "s23");     /* two gates, software-pipelined: 6 instructions */
#define K16(i) { const unsigned long long pv = __builtin_bit_cast(unsigned long long, p[i]); asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(la8), "v"(pv), "n"(i * 1024) : "memory"); }
#define K17(i) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(la2), "v"(u[i]), "n"(i * 1024) : "memory");
#define K18(i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(la), "v"(f4v), "n"(i * 1024) : "memory");
#define K19(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(la4), "v"(u[i]), "n"(i * 1024) : "memory");
#define K20(i) { unsigned long long rv; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(rv) : "v"(la8), "n"(i * 1024) : "memory"); u[i] ^= (unsigned)rv; }
#define K21(i) { f32x4v rv; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rv) : "v"(la), "n"(i * 1024) : "memory"); v[i] += rv[0]; }
#define K22(i) { const unsigned long long pv = __builtin_bit_cast(unsigned long long, p[i]); asm volatile("ds_write2st64_b64 %0, %1, %1 offset0:%2 offset1:%3" :: "v"(la8), "v"(pv), "n"(i), "n"(i + 8) : "memory"); }
            if (KIND == 0) { REP8(K0) } if (KIND == 1) { REP8(K1) } if (KIND == 2) { REP8(K2) } if (KIND == 3) { REP8(K3) }
            if (KIND == 4) { REP8(K4) } if (KIND == 5) { REP8(K5) } if (KIND == 6) { REP8(K6) } if (KIND == 7) { REP8(K7) }
            if (KIND == 8) { REP8(K8) } if (KIND == 9) { REP8(K9) } if (KIND == 10) { REP8(K10) } if (KIND == 11) { REP8(K11) }
            if (KIND == 12) { REP8(K12) } if (KIND == 13) { REP8(K13) } if (KIND == 14) { REP8(K14) } if (KIND == 15) { REP8(K15) }
            if (KIND == 16) { REP8(K16) } if (KIND == 17) { REP8(K17) } if (KIND == 18) { REP8(K18) }
            if (KIND == 19) { REP8(K19) } if (KIND == 20) { REP8(K20) } if (KIND == 21) { REP8(K21) } if (KIND == 22) { REP8(K22) }
        }
        if (KIND >= 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += v[i] + w[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 512 + threadIdx.x] = r + (float)lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND> static void run(const char* name, int per_slot, float* out, unsigned long long* cyc)
{
    const int iters = 400;
    double res[2];
    for (int two = 0; two < 2; ++two) {
        const int threads = two ? 512 : 256;
        hipMemset(cyc, 0, 256 * 8 * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((cost<KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (int g = 0; g < 256; ++g) for (int w = 0; w < threads / 64; ++w) c.push_back((double)h[g * 8 + w] / (iters * 16.0 * 8.0 * per_slot));
        std::sort(c.begin(), c.end());
        res[two] = c[c.size() / 2];
    }
    printf("%-56s one wave per SIMD: %6.2f cycles per instruction | two waves per SIMD: %6.2f per instruction of each wave = %5.2f of the SIMD\n", name, res[0], res[1], res[1] / 2);
}
int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("v_fma_f32", 1, out, cyc);
    run<1>("v_max_f32", 1, out, cyc);
    run<10>("v_mul_f32", 1, out, cyc);
    run<9>("v_add_u32", 1, out, cyc);
    run<8>("v_ldexp_f32", 1, out, cyc);
    run<2>("v_cvt_pk_f16_f32", 1, out, cyc);
    run<3>("v_fma_mixlo_f16", 1, out, cyc);
    run<13>("v_fma_mixhi_f16", 1, out, cyc);
    run<4>("v_pk_mul_f32", 1, out, cyc);
    run<5>("v_pk_add_f32", 1, out, cyc);
    run<6>("v_pk_fma_f32", 1, out, cyc);
    run<12>("v_cmp_lt_f32 -> vcc", 1, out, cyc);
    run<11>("v_cndmask_b32 (vcc)", 1, out, cyc);
    run<7>("v_cmp_lt_f32 vcc ; v_cndmask_b32 vcc (dependent pair)", 2, out, cyc);
    run<15>("2 x (v_cmp -> s[..]) ; 2 x v_mul ; 2 x v_cndmask (6)", 6, out, cyc);
    run<14>("v_mul_f32 ; v_max_f32 (dependent pair)", 2, out, cyc);
    run<16>("ds_write_b64 (conflict-free, drained every 128)", 1, out, cyc);
    run<17>("ds_write_b16", 1, out, cyc);
    run<18>("ds_write_b128", 1, out, cyc);
    run<19>("ds_write_b32", 1, out, cyc);
    run<22>("ds_write2st64_b64 (two 8-byte stores)", 1, out, cyc);
    run<20>("ds_read_b64_tr_b16", 1, out, cyc);
    run<21>("ds_read_b128", 1, out, cyc);
    return 0;
}
