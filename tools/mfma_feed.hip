// Diagnostic (not part of the product): what operand delivery costs a tap loop.  One "tap" = twelve v_mfma_f32_16x16x32_f16 on four accumulators
// (the K32 form of conv3_pp_kernel); beside them NL ds_read_b128 and NV buffer_load_dwordx4 per tap, requested `AHEAD` taps before they are used
// (register ring), conflict-free LDS image, filter-like global buffer of FB bytes that all waves of a workgroup read together.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/mfma_feed.hip -o tools/mfma_feed.bin && tools/mfma_feed.bin
// Prints cycles per tap (s_memtime, median over waves) for one tap-wave per SIMD (256 threads) and two (512 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
union Frag { u32x4 u; f16x8 h; };

template <int NL, int NV, int AHEAD, int NM = 12>
__global__ __launch_bounds__(512) void feed_kernel(const u32x4* __restrict__ wbuf, int wbytes, int taps, float* sink, unsigned long long* stamps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = wbuf[i & 1023];
    __syncthreads();
    constexpr int RD = AHEAD + 1;                       // ring depth
    Frag A[RD][4], W[RD][4];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wbuf), 0, wbytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < RD; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) { A[r][k].u = wbuf[lane + 64 * k]; W[r][k].u = wbuf[lane + 64 * (k + 4)]; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    int woff = 0, loff = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int t = 0; t < taps; t += RD) {
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            constexpr int dummy = 0; (void)dummy;
            const int q = (r + AHEAD) % RD;             // ring slot requested now (held the tap before this one)
#define MM(C, a, b) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[r][a].h, W[r][b].h, C, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
            MM(c0, 1, 0);
            if (NV > 0) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, woff, 0); W[q][0].u = v; }
            if (NV > 1) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, woff, 0); W[q][1].u = v; }
            __builtin_amdgcn_sched_barrier(0);
            MM(c1, 1, 2);
            if (NV > 2) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 2048, woff, 0); W[q][2].u = v; }
            if (NV > 3) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 3072, woff, 0); W[q][3].u = v; }
            __builtin_amdgcn_sched_barrier(0);
            MM(c2, 3, 0);
            if (NL > 0) A[q][0].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16);
            if (NL > 1) A[q][1].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 1024);
            __builtin_amdgcn_sched_barrier(0);
            MM(c3, 3, 2);
            if (NL > 2) A[q][2].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 2048);
            if (NL > 3) A[q][3].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 3072);
            __builtin_amdgcn_sched_barrier(0);
            MM(c0, 0, 1);
            if (NL > 4) A[q][0].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 4096);
            if (NL > 5) A[q][1].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 5120);
            __builtin_amdgcn_sched_barrier(0);
            MM(c1, 0, 3);
            if (NL > 6) A[q][2].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 6144);
            if (NL > 7) A[q][3].u = *reinterpret_cast<const u32x4*>(lds + loff + lane * 16 + 7168);
            __builtin_amdgcn_sched_barrier(0);
            MM(c2, 2, 1); MM(c3, 2, 3);
            MM(c0, 0, 0); MM(c1, 0, 2); MM(c2, 2, 0); MM(c3, 2, 2);
            if (NM > 12) {                              // a 64-voxel tile: twelve more on four more accumulators
                MM(c4, 1, 0); MM(c5, 1, 2); MM(c6, 3, 0); MM(c7, 3, 2); MM(c4, 0, 1); MM(c5, 0, 3); MM(c6, 2, 1); MM(c7, 2, 3); MM(c4, 0, 0); MM(c5, 0, 2); MM(c6, 2, 0); MM(c7, 2, 2);
            }
#undef MM
            woff += 4096; if (woff >= wbytes) woff = 0;
            loff = (loff + 4096) & 16383;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] + s[1] + s[2] + s[3] == 1234.5f) sink[threadIdx.x] = s[0];
}

template <int NL, int NV, int AHEAD, int NM = 12>
static void run(const char* name, const u32x4* w, int wbytes, float* sink, unsigned long long* st, int threads)
{
    const int taps = 27 * 40 / (AHEAD + 1) * (AHEAD + 1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<NL, NV, AHEAD, NM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((feed_kernel<NL, NV, AHEAD, NM>), dim3(256), dim3(threads), 150 * 1024, 0, w, wbytes, taps, sink, st);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < 256; ++b) for (int wv = 0; wv < threads / 64; ++wv) c.push_back((double)h[b * 8 + wv] / taps);
    std::sort(c.begin(), c.end());
    printf("%-58s %d waves/SIMD: %7.1f cycles per tap (p90 %.1f)%s\n", name, threads / 256, c[c.size() / 2], c[c.size() * 9 / 10], hipGetLastError() == hipSuccess ? "" : "  !! HIP error");
}

int main()
{
    const int WB = 27 * 4096;
    std::vector<unsigned> hw(WB / 4);
    unsigned long long s = 88172645463325252ull;
    for (auto& u : hw) {
        unsigned short hh[2];
        for (int q = 0; q < 2; ++q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; _Float16 hf = (_Float16)((float)((s >> 11) & 0xffffff) / 8388608.f - 1.f); hh[q] = *reinterpret_cast<unsigned short*>(&hf); }
        u = hh[0] | ((unsigned)hh[1] << 16);
    }
    u32x4* w; float* sink; unsigned long long* st;
    hipMalloc(&w, WB); hipMalloc(&sink, 4096); hipMalloc(&st, 256 * 8 * 8);
    hipMemcpy(w, hw.data(), WB, hipMemcpyHostToDevice);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0, 0, 2>("12 MFMA alone", w, WB, sink, st, threads);
        run<4, 0, 2>("+ 4 ds_read_b128, 2 taps ahead", w, WB, sink, st, threads);
        run<0, 4, 2>("+ 4 buffer_load (110 KB round), 2 taps ahead", w, WB, sink, st, threads);
        run<0, 4, 3>("+ 4 buffer_load (110 KB round), 3 taps ahead", w, WB, sink, st, threads);
        run<0, 2, 2>("+ 2 buffer_load (110 KB round), 2 taps ahead", w, WB, sink, st, threads);
        run<4, 4, 2>("+ 4 ds_read_b128 + 4 buffer_load, 2 taps ahead", w, WB, sink, st, threads);
        run<4, 4, 3>("+ 4 ds_read_b128 + 4 buffer_load, 3 taps ahead", w, WB, sink, st, threads);
        run<0, 4, 2>("+ 4 buffer_load (4 KB: L1 hits), 2 taps ahead", w, 4096, sink, st, threads);
        run<4, 4, 2>("+ 4 ds_read_b128 + 4 buffer_load (4 KB), 2 ahead", w, 4096, sink, st, threads);
        run<4, 2, 2>("+ 4 ds_read_b128 + 2 buffer_load, 2 ahead", w, WB, sink, st, threads);
        run<8, 2, 2>("+ 8 ds_read_b128 + 2 buffer_load, 2 ahead", w, WB, sink, st, threads);
        run<8, 0, 2>("+ 8 ds_read_b128, 2 ahead", w, WB, sink, st, threads);
        run<4, 1, 2>("+ 4 ds_read_b128 + 1 buffer_load, 2 ahead", w, WB, sink, st, threads);
        run<0, 0, 2, 24>("24 MFMA alone", w, WB, sink, st, threads);
        run<8, 4, 2, 24>("24 MFMA + 8 ds_read_b128 + 4 buffer_load, 2 ahead", w, WB, sink, st, threads);
        run<8, 4, 1, 24>("24 MFMA + 8 ds_read_b128 + 4 buffer_load, 1 ahead", w, WB, sink, st, threads);
        run<8, 0, 2, 24>("24 MFMA + 8 ds_read_b128, 2 ahead", w, WB, sink, st, threads);
        run<0, 4, 2, 24>("24 MFMA + 4 buffer_load, 2 ahead", w, WB, sink, st, threads);
    }
    return 0;
}
