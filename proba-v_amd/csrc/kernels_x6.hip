// fp32 products on the bf16 matrix pipe: "x6" split kernels for gfx950.
//
// Every fp32 operand x is cut into three bf16 pieces by truncation,
//     x0 = x & 0xffff0000,  x1 = (x - x0) & 0xffff0000,  x2 = x - x0 - x1        (x == x0 + x1 + x2 EXACTLY: 8+8+8 bits)
// and a product a*b is evaluated as the six largest of the nine piece products
//     a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)            (dropped: a1b2 + a2b1 + a2b2 <= 3 * 2^-24 |ab|)
// each of them EXACT in fp32 (8 x 8 significant bits) and summed in the MFMA's fp32 accumulator.  The error of one
// product is therefore of the order of the fp32 rounding error of that product, and the result is fp32-grade; the
// parity tests hold these kernels to the same tolerances as the native fp32-MFMA kernels.
// v_mfma_f32_32x32x16_bf16 retires 32*32*16 MACs in 32 cycles, v_mfma_f32_32x32x2_f32 32*32*2 in 64: six bf16
// instructions replace sixteen fp32 ones, 2.67x less matrix-pipe time.
//
// Operand lane maps (cdna_hip_programming.md §3): lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7, in one 16-byte fragment.  An accumulator tile (row (i&3)+8(i>>2)+4h in register i)
// feeds the next product's B operand when that product contracts over the tile's ROW index: registers 0..7 are the
// k-slots of k-block 0, registers 8..15 those of k-block 1, and the A operand is packed with the same permutation.
#include "kernels_x6.h"
#include "x6_device.h"

namespace probav {

// ---------------------------------------------------------------------------------------------------
// fused expConv + ReLU + decConv forward (1x1x1, 32 -> 256 -> D <= 32); one 32-voxel tile per wave and round.
// Both weight sets live in LDS as pre-split fragments (2 x 48 KB); X comes straight from HBM into B fragments;
// the 256-channel hidden tensor never leaves registers.
// ---------------------------------------------------------------------------------------------------
constexpr int PWF_WAVES = 8;

__global__ __launch_bounds__(64 * PWF_WAVES, 1) void pw_fwd_x6_kernel(const float* __restrict__ x, const uint4* __restrict__ w1frag,
                                                                     const uint4* __restrict__ w2frag, const float* __restrict__ b1,
                                                                     const float* __restrict__ b2, float* __restrict__ dec,
                                                                     long nvox, int D)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* sW1 = reinterpret_cast<uint4*>(lds_raw);             // [8 chunks][2 kb][3 pieces][64 lanes]  48 KB
    uint4* sW2 = sW1 + 8 * 6 * 64;                               // same
    float* sB1 = reinterpret_cast<float*>(sW2 + 8 * 6 * 64);    // 256
    float* sB2 = sB1 + 256;                                      // 32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    for (int i = tid; i < 8 * 6 * 64; i += 64 * PWF_WAVES) { sW1[i] = w1frag[i]; sW2[i] = w2frag[i]; }
    if (tid < 256) sB1[tid] = b1[tid];
    if (tid < 32) sB2[tid] = tid < D ? b2[tid] : 0.f;
    __syncthreads();

    const long ntiles = (nvox + 31) >> 5;
    const long wstride = (long)gridDim.x * PWF_WAVES;
    for (long tile = (long)blockIdx.x * PWF_WAVES + wave; tile < ntiles; tile += wstride) {
        long v = tile * 32 + col;
        const bool vok = v < nvox;
        if (!vok) v = nvox - 1;
        // B operand of the first product: X^T, k = cin 16kb + 8h + j
        Frag xb[2][3];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4* xp = reinterpret_cast<const float4*>(x + v * 32 + 16 * kb + 8 * h);
            const float4 t0 = xp[0], t1 = xp[1];
            const float xs[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            split8(xs, xb[kb]);
        }
        f32x16 T;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            f32x16 H;
#pragma unroll
            for (int r = 0; r < 16; ++r) H[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p].u = sW1[((c * 2 + kb) * 3 + p) * 64 + lane];
                H = mac6(a, xb[kb], H);
            }
            // bias + ReLU, then split the hidden tile: registers 8kb .. 8kb+7 are k-block kb of the second product
            Frag hb[2][3];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float hs[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 bb = *reinterpret_cast<const float4*>(sB1 + 32 * c + 8 * (2 * kb + g) + 4 * h);
                    hs[4 * g + 0] = fmaxf(H[8 * kb + 4 * g + 0] + bb.x, 0.f);
                    hs[4 * g + 1] = fmaxf(H[8 * kb + 4 * g + 1] + bb.y, 0.f);
                    hs[4 * g + 2] = fmaxf(H[8 * kb + 4 * g + 2] + bb.z, 0.f);
                    hs[4 * g + 3] = fmaxf(H[8 * kb + 4 * g + 3] + bb.w, 0.f);
                }
                split8(hs, hb[kb]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p].u = sW2[((c * 2 + kb) * 3 + p) * 64 + lane];
                T = mac6(a, hb[kb], T);
            }
        }
        if (vok) {
            float* o = dec + v * D;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = rowmap(r, h);
                if (ch < D) o[ch] = T[r] + sB2[ch];
            }
        }
    }
}

int x6_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                  long nvox, int D, hipStream_t s)
{
    static bool once = false;
    const size_t lds = (size_t)2 * 8 * 6 * 64 * 16 + (256 + 32) * sizeof(float);
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(pw_fwd_x6_kernel, dim3(256), dim3(64 * PWF_WAVES), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag,
                       b1, b2, dec, nvox, D);
    return check_launch("pw_fwd_x6");
}

}  // namespace probav
