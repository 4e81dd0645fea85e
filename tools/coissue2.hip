// Diagnostic: does a wave's VALU stream issue beside its own MFMAs when the MFMA's C/D are arch VGPRs (builtin, -amdgpu-mfma-vgpr-form) vs AGPRs (asm "+a")?
// One wave per SIMD, every CU busy; per variant: cycles per [MFMA + NF fillers] group.   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/coissue2.hip -o tools/coissue2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_out[4096 * 2];

template <int MODE, int NF, int FK>
__global__ __launch_bounds__(256, 1) void k(const f16x8* __restrict__ src, float* __restrict__ sink, int iters)
{
    const int lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) unsigned char ldsb[16384 + 4096];
    for (int i = lane; i < 4096 + 1024; i += 256) reinterpret_cast<unsigned*>(ldsb)[i] = i;
    __syncthreads();
    u32x4 lt[4] = {};
    f16x8 a = src[lane], b = src[lane + 256];
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float f[16];
    for (int r = 0; r < 16; ++r) f[r] = (float)lane * 0.001f + r;
    const float c0 = 1.0001f, c1 = 0.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            else if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
            else if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));      // A operand in the accumulator half, C/D in the vector half (kernels_cw4.hip)
            else if (MODE == 4) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b)); u32x4 t = *reinterpret_cast<const u32x4*>(ldsb + lane * 16 + (u & 3) * 4096); asm volatile("" :: "v"(t)); }      // + one ds_read_b128 per gap (waited for: the asm consumes it)
            else if (MODE == 5) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b)); lt[u & 3] = *reinterpret_cast<const u32x4*>(ldsb + lane * 16 + (u & 3) * 4096); if ((u & 3) == 3) asm volatile("" :: "v"(lt[0]), "v"(lt[1]), "v"(lt[2]), "v"(lt[3])); }      // + one ds_read_b128 per gap, consumed four gaps later
            else if (MODE == 2) { if (u & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0); else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0); }
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                float& v = f[(u * NF + q) & 15];
                if (FK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c0), "v"(c1));
                else if (FK == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(c1));
                else if (FK == 2) { unsigned t; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(c0)); asm volatile("" :: "v"(t)); }
                else if (FK == 3) { unsigned t = 0; asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(__float_as_uint(c0)), "v"(v)); asm volatile("" :: "v"(t)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r] + f[r];
    sink[blockIdx.x * 256 + lane] = s;
    if ((lane & 63) == 0) g_out[(blockIdx.x * 4 + (lane >> 6))] = t1 - t0;
}

template <int MODE, int NF, int FK> static void run(const char* nm, const f16x8* src, float* sink)
{
    const int iters = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, NF, FK>), dim3(256), dim3(256), 0, 0, src, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> o(1024);
    hipMemcpyFromSymbol(o.data(), HIP_SYMBOL(g_out), 1024 * 8);
    std::sort(o.begin(), o.end());
    printf("%-44s fillers %d kind %d: %6.1f cycles per MFMA group\n", nm, NF, FK, (double)o[512] / (iters * 8.0));
}
int main()
{
    f16x8* src; float* sink;
    hipMalloc(&src, 512 * 16); hipMalloc(&sink, 256 * 256 * 4);
    std::vector<unsigned short> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) { _Float16 v = (_Float16)(((int)(i * 2654435761u % 2001) - 1000) * 0.001f); h[i] = *reinterpret_cast<unsigned short*>(&v); }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
#define R3(NF, FK) run<0, NF, FK>("VGPR-form builtin, one accumulator chain", src, sink); run<3, NF, FK>("A operand in a[], C/D in v[]", src, sink); run<4, NF, FK>("the same + ds_read_b128 per gap, waited", src, sink); run<5, NF, FK>("the same + ds_read_b128 per gap, 4 ahead", src, sink);
    R3(0, 0) R3(2, 0) R3(3, 0) R3(4, 0) R3(5, 0) R3(2, 2) R3(3, 2) R3(2, 3) R3(3, 3) R3(4, 3)
    return 0;
}
