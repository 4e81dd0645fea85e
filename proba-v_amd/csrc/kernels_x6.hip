// fp32 products on the bf16 matrix pipe: "x6" split kernels for gfx950.
//
// Every fp32 operand x is cut into three bf16 pieces by truncation,
//     x0 = x & 0xffff0000,  x1 = (x - x0) & 0xffff0000,  x2 = x - x0 - x1        (x == x0 + x1 + x2 EXACTLY: 8+8+8 bits)
// and a product a*b is evaluated as the six largest of the nine piece products
//     a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)            (dropped: a1b2 + a2b1 + a2b2 < 2^-21 |ab|, ~2^-23 |ab| on average)
// each of them EXACT in fp32 (8 x 8 significant bits) and summed in the MFMA's fp32 accumulator.  The error of one
// product is therefore a few fp32 ulps at most (an fp32 fma rounds to 2^-24) and the result is fp32-grade; the parity tests
// hold these kernels to the same tolerances as the native fp32-MFMA kernels (tests/test_x6_arith.py restates the arithmetic).
// v_mfma_f32_32x32x16_bf16 retires 32*32*16 MACs in 32 cycles, v_mfma_f32_32x32x2_f32 32*32*2 in 64: six bf16
// instructions replace sixteen fp32 ones, 2.67x less matrix-pipe time.
//
// Operand lane maps (cdna_hip_programming.md §3): lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7, in one 16-byte fragment.  An accumulator tile (row (i&3)+8(i>>2)+4h in register i)
// feeds the next product's B operand when that product contracts over the tile's ROW index: registers 0..7 are the
// k-slots of k-block 0, registers 8..15 those of k-block 1, and the A operand is packed with the same permutation.
#include "kernels_x6.h"
#include "x6_device.h"
#include <type_traits>
#include <cstdlib>

#include <mutex>

namespace probav {


// ---------------------------------------------------------------------------------------------------
// fused expConv + ReLU + decConv forward (1x1x1, 32 -> 256 -> D <= 32); one 32-voxel tile per wave and round.
// Both weight sets live in LDS as pre-split fragments (2 x 48 KB); X comes straight from HBM into B fragments;
// the 256-channel hidden tensor never leaves registers.
// ---------------------------------------------------------------------------------------------------
// Workgroup shape: the X6 weight images (2 x 48 KB) leave room for one workgroup per CU, so it is 12 waves wide; the H3 images
// (2 x 32 KB) fit twice: two workgroups of 8 waves = 4 waves per SIMD instead of 3.
#ifdef PROBAV_STAMP
#define PF_ACC(k) do { __builtin_amdgcn_sched_barrier(0); XS_ACC(k); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PF_ACC(k) do { } while (0)
#endif
template <class AR> struct PwfShape { static constexpr int WAVES = 12, WGS = 1; };
template <> struct PwfShape<H3> { static constexpr int WAVES = 8, WGS = 2; };

// Tiles never straddle two samples (vps voxels per sample are cut into ceil(vps / 32) tiles, the last one partial), and a wave owns a
// CONTIGUOUS run of tiles: the H3 scales of X and of the hidden tile are per-sample scalars that change once or twice per wave, and a
// sample's largest output magnitude is committed when the run leaves the sample.  DUMP (tests only): the post-ReLU hidden tile is
// also written to hdump [nvox][256] (at the hidden tile's own power-of-two scale: only signs and relative sizes mean anything).
template <class AR, bool DUMP>
__global__ __launch_bounds__(64 * PwfShape<AR>::WAVES, PwfShape<AR>::WGS) void pw_fwd_x6_kernel(const float* __restrict__ x, const uint4* __restrict__ w1frag,
                                                                     const uint4* __restrict__ w2frag, const float* __restrict__ b1,
                                                                     const float* __restrict__ b2, float* __restrict__ dec,
                                                                     long nvox, int vps, int D, PwAmax am, float* __restrict__ hdump)
{
    constexpr int NP = AR::NP, PWF_WAVES = PwfShape<AR>::WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* sW1 = reinterpret_cast<uint4*>(lds_raw);             // [8 chunks][2 kb][NP pieces][64 lanes]  48 / 32 KB
    uint4* sW2 = sW1 + 8 * 2 * NP * 64;                          // same
    float* sB1 = reinterpret_cast<float*>(sW2 + 8 * 2 * NP * 64);    // 256
    float* sB2 = sB1 + 256;                                      // 32
    int* sE2 = reinterpret_cast<int*>(sB2 + 32);                 // 32: H3 exponent of decConv's output column d (its weights are cut per column)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8 * 2 * NP * 64; i += 64 * PWF_WAVES) { sW1[i] = w1frag[i]; sW2[i] = w2frag[i]; }
    if (tid < 256) sB1[tid] = b1[tid];
    if (tid < 32) { sB2[tid] = tid < D ? b2[tid] : 0.f; sE2[tid] = (AR::SCALED && tid < D) ? h3_exp_w(am.w2c[tid]) : 0; }
    __syncthreads();
    // H3 scales.  The hidden tensor exists only in registers, so its scale comes from a bound: |h| <= 32 amax(x) amax(w1) + amax(b1), per sample.
    float sx = 1.f, sbias = 1.f, c1 = 1.f; int eh = 0;
    unsigned aw1 = 0u, ab1 = 0u; int ew1 = 0;
    if constexpr (AR::SCALED) { aw1 = *am.w1; ab1 = *am.b1; ew1 = h3_exp_w(aw1); }
    auto sample_scales = [&](int n) {
        if constexpr (AR::SCALED) {
            const unsigned ax = am.x[n];
            const int ex = h3_exp(ax);
            eh = h3_exp(32.f * __uint_as_float(ax) * __uint_as_float(aw1) + __uint_as_float(ab1));
            sx = pow2i(ex); sbias = pow2i(eh);
            const int k1 = -(ex + ew1);                              // accumulator of the first product -> true hidden values (bias added there, then 2^eh)
            c1 = pow2i(k1 < -126 ? -126 : k1);
        }
    };

    const int tps = (vps + 31) >> 5;                                 // tiles per sample
    const long ntiles = (nvox / vps) * tps;
    const long gw = (long)blockIdx.x * PWF_WAVES + wave, nw = (long)gridDim.x * PWF_WAVES;
    const long tb = ntiles * gw / nw, te = ntiles * (gw + 1) / nw;   // this wave's run of tiles
    int n = (int)(tb / tps), j = (int)(tb - (long)n * tps);
    float omax = 0.f;
    if (tb < te) sample_scales(n);
    // The X rows of a tile are requested one tile ahead, BEFORE the previous tile's output stores: consumed at the very top of a tile,
    // their whole latency was exposed, and behind the stores in the in-order vmcnt queue they also waited for those to retire.
    float4 nx[2][2];
    auto load_x = [&](int nn, int jj) {
        const int vl = 32 * jj + col;
        const long v = (long)nn * vps + (vl < vps ? vl : vps - 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4* xp = reinterpret_cast<const float4*>(x + v * 32 + 16 * kb + 8 * h);
            nx[kb][0] = xp[0]; nx[kb][1] = xp[1];
        }
    };
    if (tb < te) load_x(n, j);
    XS_DECL;
#ifdef PROBAV_STAMP
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (long tile = tb; tile < te; ++tile) {
        PF_ACC(5);
        const int vl = 32 * j + col;                                 // voxel inside the sample
        const bool vok = vl < vps;
        const long v = (long)n * vps + (vok ? vl : vps - 1);
        // B operand of the first product: X^T, k = cin 16kb + 8h + j
        Frag xb[2][NP];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 t0 = nx[kb][0], t1 = nx[kb][1];
            const float xs[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            cut8<AR>(xs, sx, xb[kb]);
        }
        if (tile + 1 < te) { const bool wrap = j + 1 == tps; load_x(wrap ? n + 1 : n, wrap ? 0 : j + 1); }
        PF_ACC(1);
        f32x16 T;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            f32x16 H;
#pragma unroll
            for (int r = 0; r < 16; ++r) H[r] = 0.f;
            // one address per chunk; the fragments of the chunk (k-block, piece; W2 behind W1) are immediate offsets from it
            const unsigned char* wc = reinterpret_cast<const unsigned char*>(sW1) + lane * 16 + c * (2 * NP * 64 * 16);
            constexpr int W2OFF = 8 * 2 * NP * 64 * 16;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag a[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) a[p].u = *reinterpret_cast<const uint4*>(wc + (kb * NP + p) * 1024);
                H = mac<AR>(a, xb[kb], H);
            }
#ifdef PROBAV_STAMP
            asm volatile("s_nop 0" :: "v"(H[0]));
#endif
            PF_ACC(2);
            // bias + ReLU, then split the hidden tile: registers 8kb .. 8kb+7 are k-block kb of the second product
            Frag hb[2][NP];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float hs[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 bb = *reinterpret_cast<const float4*>(sB1 + 32 * c + 8 * (2 * kb + g) + 4 * h);
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hs[4 * g + i] = fmaxf(fmaf(H[8 * kb + 4 * g + i], c1, bv[i]), 0.f);      // true hidden values (c1 = 1 without scaling)
                    }
                }
                cut8<AR>(hs, sbias, hb[kb]);                         // pieces of h 2^eh
                if constexpr (DUMP) {
                    if (vok) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) hdump[v * 256 + 32 * c + rowmap(8 * kb + i, h)] = hs[i];
                    }
                }
            }
            PF_ACC(3);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag a[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) a[p].u = *reinterpret_cast<const uint4*>(wc + W2OFF + (kb * NP + p) * 1024);
                T = mac<AR>(a, hb[kb], T);
            }
            PF_ACC(4);
        }
        if (vok) {
            // registers 4g .. 4g+3 are the four CONSECUTIVE output channels 8g + 4h + (0..3) of this lane's voxel: one 16-byte store each
            // (the rows of dec are D floats long, so the stores are only 4-byte aligned: the hardware splits them where it must)
            float* o = dec + v * D;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 8 * g + 4 * h;
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = T[4 * g + i];
                    if constexpr (AR::SCALED) t[i] = ldexpf(t[i], -(sE2[c0 + i] + eh));   // accumulator of the second product -> true values
                    t[i] += sB2[c0 + i];
                }
                if (c0 + 4 <= D) {
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                    f32x4u q = {t[0], t[1], t[2], t[3]};
                    *reinterpret_cast<f32x4u*>(o + c0) = q;
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (c0 + i < D) { o[c0 + i] = t[i]; omax = fmaxf(omax, fabsf(t[i])); }
                }
            }
        }
        if (++j == tps) {                                            // the run leaves sample n
            if (am.y) amax_commit(omax, am.y + n);
            omax = 0.f; j = 0; ++n;
            if (tile + 1 < te) sample_scales(n);
        }
    }
    if (am.y && j != 0 && tb < te) amax_commit(omax, am.y + n);
#ifdef PROBAV_STAMP
    xs_acc[6] = __builtin_amdgcn_s_memrealtime() - rt0;
#endif
    XS_OUT;
}

// ---------------------------------------------------------------------------------------------------
// The same fused forward on v_mfma_f32_16x16x32_f16 (H3 only): 32 input channels are ONE k-block of that shape, a hidden chunk of 32 is
// one k-block of the second product.  The kernel is the one above with another register tiling -- the LDS images of the two weight sets are
// read IN PLACE with another lane order, nothing is packed twice:
//   first product   H^T[hidden 16 s + .][voxel 16 u + .]: A = W1^T, lane (m, kq) wants k = cin 8 kq .. + 7 of hidden row 16 s + m =
//                   lane (16 s + m) + 32 (kq & 1) of the 32x32x16 fragment kb = kq >> 1 (one ds_read_b128); B = X^T, lane (n, kq): cin 8 kq .. of voxel 16 u + n
//   its accumulator lane (n, kq), register i = hidden 16 s + 4 kq + i of voxel 16 u + n: registers (s, i) of a lane are k-slots j = 4 s + i of
//                   the second product's B operand (k = 8 kq + j) once the matching A operand is packed with that order: hidden 16 (j >> 2) + 4 kq + (j & 3),
//                   which in the W2 image (k-slot kp = rowmap(8 kb + j', half)) is fragment kb = j >> 2, lane (16 o + m) + 32 (kq & 1), bytes 8 (kq >> 1) .. + 7
//                   (two ds_read_b64 per fragment)
//   second product  T^T[out 16 o + 4 kq + i][voxel 16 u + n]: a lane holds four consecutive output channels of one voxel -> 16-byte stores
// Why: cycles per FLOP are the same, but the chip holds a higher clock on this shape (docs/notebook_r1-r5.md 4.1f), and this kernel ran at the lowest clock of the set.
// The hidden tile's sums are NOT bit-identical to the 32x32x16 kernel's (32 against 16 products per instruction); the backward pass recomputes the tile
// with the 32x32x16 arrangement, and so does probav_debug_hidden: gates of pre-activations that sit at zero to the last bit can differ between the passes.
// ---------------------------------------------------------------------------------------------------
template <bool DUMP>      // DUMP (tests only): the post-ReLU hidden tile also goes to hdump [nvox][256], at the hidden tile's own power-of-two scale (as pw_fwd_x6_kernel's)
__global__ __launch_bounds__(512, 2) void pw_fwd_h3k_kernel(const float* __restrict__ x, const uint4* __restrict__ w1frag, const uint4* __restrict__ w2frag,
                                                            const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ dec,
                                                            long nvox, int vps, int D, PwAmax am, float* __restrict__ hdump)
{
    XS_ENTRY;
    using AR = H3;
    constexpr int NP = 2, WAVES = 8;
    typedef float f32x4k __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* sW1 = reinterpret_cast<uint4*>(lds_raw);             // [8 chunks][2 kb][NP pieces][64 lanes]  32 KB
    uint4* sW2 = sW1 + 8 * 2 * NP * 64;                          // same
    float* sB1 = reinterpret_cast<float*>(sW2 + 8 * 2 * NP * 64);    // 256
    float* sB2 = sB1 + 256;                                      // 32
    int* sE2 = reinterpret_cast<int*>(sB2 + 32);                 // 32: H3 exponent of decConv's output column d
    float* sBw = reinterpret_cast<float*>(sE2 + 32);             // [8 waves][256]: the expand biases at THIS WAVE's current sample's scale (2^(eh - 16)); private to the wave
    const int tid = threadIdx.x, lane = tid & 63, m16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the wave's run of tiles, and its first tile's rows requested before anything else: they are in flight while the weight images are copied
    const int tps = (vps + 31) >> 5;
    const long ntiles = (nvox / vps) * tps;
    // Partition: the workgroup's share first, then its four SIMD pairs (waves w and w + 4 sit on one SIMD and share its pipes: what must be even is
    // the PAIR's count), then the pair's two waves -- a SIMD then carries 8 or 9 tiles of this workgroup instead of 8 to 10 (17 536 tiles / 4 096 waves = 4.28:
    // dealt per wave, a SIMD's four waves of two workgroups held 16 to 19, and the launch waited for the 19s)
    const long wb = ntiles * blockIdx.x / gridDim.x, we = ntiles * (blockIdx.x + 1) / gridDim.x;
    const int pr = wave & 3, hf = wave >> 2;
    const long pb = wb + (we - wb) * pr / 4, pe = wb + (we - wb) * (pr + 1) / 4;
    const long tb = hf ? pb + (pe - pb + 1) / 2 : pb, te = hf ? pe : pb + (pe - pb + 1) / 2;
    int n = (int)(tb / tps), j = (int)(tb - (long)n * tps);
    float4 nx[2][2];                                             // [u][two float4]: cin 8 kq .. 8 kq + 7 of voxel 16 u + m16
    auto load_x = [&](int nn, int jj) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int vl = 32 * jj + 16 * u + m16;
            const long v = (long)nn * vps + (vl < vps ? vl : vps - 1);
            const float4* xp = reinterpret_cast<const float4*>(x + v * 32 + 8 * kq);
            nx[u][0] = xp[0]; nx[u][1] = xp[1];
        }
    };
    if (tb < te) load_x(n, j);
    {   // the two weight images: all eight 16-byte requests of a thread first, then the stores -- one round trip, not four in a row
        constexpr int NI = 8 * 2 * NP * 64 / (64 * WAVES);       // = 4
        uint4 q1[NI], q2[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) { q1[k] = w1frag[tid + 64 * WAVES * k]; q2[k] = w2frag[tid + 64 * WAVES * k]; }
        const float bq = tid < 256 ? b1[tid] : 0.f;
        float b2q = 0.f; int e2q = 0;
        if (tid < 32) { b2q = tid < D ? b2[tid] : 0.f; e2q = tid < D ? h3_exp_w(am.w2c[tid]) : 0; }
#pragma unroll
        for (int k = 0; k < NI; ++k) { sW1[tid + 64 * WAVES * k] = q1[k]; sW2[tid + 64 * WAVES * k] = q2[k]; }
        if (tid < 256) sB1[tid] = bq;
        if (tid < 32) { sB2[tid] = b2q; sE2[tid] = e2q; }
    }
    __syncthreads();
    // Bias + ReLU as ONE instruction: v_fma_f32 ... clamp computes min(max(H c + b, 0), 1), so the accumulator is brought to 2^-16 of the hidden tile's
    // scale (where the tile's bound is < 1/2: the upper clamp never acts), and an exact multiplication by 2^16 follows.  The bits are those of
    // max(fma(H, c1, b), 0) * 2^eh (powers of two commute with the fma's one rounding); what is gone is the v_max_f32 per element -- this kernel is bound by
    // its vector instructions at their real prices (docs/notebook_r1-r5.md 4.0: a hidden element costs fma 5.7 + max 5.6 + scale 6.2 + cvt 2.9 + mix 9.6 cycles of its SIMD).
    float sx = 1.f, c1c = 1.f, sb2 = 1.f; int eh = 0;
    const unsigned aw1 = *am.w1, ab1 = *am.b1;
    const int ew1 = h3_exp_w(aw1);
    auto sample_scales = [&](int n) {
        const unsigned ax = am.x[n];
        const int ex = h3_exp(ax);
        eh = h3_exp(32.f * __uint_as_float(ax) * __uint_as_float(aw1) + __uint_as_float(ab1));
        sx = pow2i(ex);
        const int k1 = -(ex + ew1), k1c = k1 < -126 ? -126 : k1;
        const int kc = k1c + eh - 16, kb = eh - 16;
        c1c = pow2i(kc < -126 ? -126 : kc);                      // accumulator of the first product -> hidden values at 2^(eh - 16)
        sb2 = pow2i(kb < -126 ? -126 : kb);                      // the biases likewise: rewritten once per sample of the wave's run (same wave writes and reads: program order)
#pragma unroll
        for (int q = 0; q < 4; ++q) sBw[wave * 256 + lane + 64 * q] = sB1[lane + 64 * q] * sb2;
    };
    float omax = 0.f;
    if (tb < te) sample_scales(n);
    // lane parts of the operand addresses inside the weight images (bytes)
    const int a1l = ((kq >> 1) * NP * 64 + m16 + 32 * (kq & 1)) * 16;                 // + s * 256 + piece * 1024 + chunk * (2 NP 1024)
    const int a2l = (m16 + 32 * (kq & 1)) * 16 + 8 * (kq >> 1);                      // + o * 256 + piece * 1024 + kb * (NP 1024) + chunk * (2 NP 1024)
    const unsigned char* w1b = reinterpret_cast<const unsigned char*>(sW1) + a1l;
    const unsigned char* w2b = reinterpret_cast<const unsigned char*>(sW2) + a2l;
    XS_DECL;
    for (long tile = tb; tile < te; ++tile) {
        Frag xb[2][NP];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 t0 = nx[u][0], t1 = nx[u][1];
            const float xs[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            cut8<AR>(xs, sx, xb[u]);
        }
        if (tile + 1 < te) { const bool wrap = j + 1 == tps; load_x(wrap ? n + 1 : n, wrap ? 0 : j + 1); }
        f32x4k T[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o) { T[o][0] = (f32x4k){0.f, 0.f, 0.f, 0.f}; T[o][1] = T[o][0]; }
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            const unsigned char* wc1 = w1b + c * (2 * NP * 1024);
            const unsigned char* wc2 = w2b + c * (2 * NP * 1024);
            Frag a1[2][NP];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < NP; ++p) a1[s][p].u = *reinterpret_cast<const uint4*>(wc1 + s * 256 + p * 1024);
            f32x4k H[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { H[s][0] = (f32x4k){0.f, 0.f, 0.f, 0.f}; H[s][1] = H[s][0]; }
#define PWK(C, a, pa, b, pb) C = __builtin_amdgcn_mfma_f32_16x16x32_f16((a)[pa].h, (b)[pb].h, C, 0, 0, 0)
            PWK(H[0][0], a1[0], 1, xb[0], 0); PWK(H[0][1], a1[0], 1, xb[1], 0); PWK(H[1][0], a1[1], 1, xb[0], 0); PWK(H[1][1], a1[1], 1, xb[1], 0);
            PWK(H[0][0], a1[0], 0, xb[0], 1); PWK(H[0][1], a1[0], 0, xb[1], 1); PWK(H[1][0], a1[1], 0, xb[0], 1); PWK(H[1][1], a1[1], 0, xb[1], 1);
            PWK(H[0][0], a1[0], 0, xb[0], 0); PWK(H[0][1], a1[0], 0, xb[1], 0); PWK(H[1][0], a1[1], 0, xb[0], 0); PWK(H[1][1], a1[1], 0, xb[1], 0);
            // the second product's A operand: k-slots j < 4 from the kb = 0 fragment, j >= 4 from kb = 1 (8 bytes each)
            Frag a2[2][NP];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(wc2 + o * 256 + p * 1024), hi = *reinterpret_cast<const uint2*>(wc2 + o * 256 + p * 1024 + NP * 1024);
                    a2[o][p].u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            // bias + ReLU + cut: registers (s, i) of H[s][u] are k-slots 4 s + i
            const float4 bq0 = *reinterpret_cast<const float4*>(sBw + wave * 256 + 32 * c + 4 * kq), bq1 = *reinterpret_cast<const float4*>(sBw + wave * 256 + 32 * c + 16 + 4 * kq);
            const float bv[8] = {bq0.x, bq0.y, bq0.z, bq0.w, bq1.x, bq1.y, bq1.z, bq1.w};
            Frag hb[2][NP];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float hs[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#ifdef H3K_OLDRELU            /* (A/B builds of tools/kbench.hip: fma, max, scale -- the same bits) */
                    hs[i] = fmaxf(fmaf(H[0][u][i], c1c, bv[i]), 0.f) * 65536.f; hs[4 + i] = fmaxf(fmaf(H[1][u][i], c1c, bv[4 + i]), 0.f) * 65536.f;
#else
                    // (med3(x, 0, 1) is the compiler's clamp pattern: it folds into the fma's clamp bit.  Not inline asm: the hazard recognizer does not
                    // see an asm statement's read of an MFMA result and would not insert the wait states between them)
                    const float t0 = __builtin_amdgcn_fmed3f(fmaf(H[0][u][i], c1c, bv[i]), 0.f, 1.f);
                    const float t1 = __builtin_amdgcn_fmed3f(fmaf(H[1][u][i], c1c, bv[4 + i]), 0.f, 1.f);
#ifdef H3K_MUL65536
                    hs[i] = t0 * 65536.f; hs[4 + i] = t1 * 65536.f;
#else
                    // x 2^16 as an integer add on the exponent field (the compiler packs the multiplications into v_pk_mul_f32: 12.6 cycles apiece beside MFMAs,
                    // tools/coissue_cycles.hip -DFK=4).  t is in [0, 1/2): no overflow; t = 0 or denormal becomes < 2^-110, both fp16 pieces of which are 0 -- as before
                    hs[i] = __uint_as_float(__float_as_uint(t0) + (16u << 23)); hs[4 + i] = __uint_as_float(__float_as_uint(t1) + (16u << 23));
#endif
#endif
                }
                cut8_scaled<AR>(hs, hb[u]);
                if constexpr (DUMP) {                            // hs[i] = hidden 32 c + 4 kq + i, hs[4 + i] = hidden 32 c + 16 + 4 kq + i of voxel 32 j + 16 u + m16
                    const int vl = 32 * j + 16 * u + m16;
                    if (vl < vps) {
                        float* hp_ = hdump + ((long)n * vps + vl) * 256 + 32 * c + 4 * kq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { hp_[i] = hs[i]; hp_[16 + i] = hs[4 + i]; }
                    }
                }
            }
            PWK(T[0][0], a2[0], 1, hb[0], 0); PWK(T[0][1], a2[0], 1, hb[1], 0); PWK(T[1][0], a2[1], 1, hb[0], 0); PWK(T[1][1], a2[1], 1, hb[1], 0);
            PWK(T[0][0], a2[0], 0, hb[0], 1); PWK(T[0][1], a2[0], 0, hb[1], 1); PWK(T[1][0], a2[1], 0, hb[0], 1); PWK(T[1][1], a2[1], 0, hb[1], 1);
            PWK(T[0][0], a2[0], 0, hb[0], 0); PWK(T[0][1], a2[0], 0, hb[1], 0); PWK(T[1][0], a2[1], 0, hb[0], 0); PWK(T[1][1], a2[1], 0, hb[1], 0);
#undef PWK
        }
        // T[o][u][i]: output channel 16 o + 4 kq + i of voxel 16 u + m16
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int vl = 32 * j + 16 * u + m16;
            if (vl < vps) {
                float* op = dec + ((long)n * vps + vl) * D;
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int c0 = 16 * o + 4 * kq;
                    float t[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = ldexpf(T[o][u][i], -(sE2[c0 + i] + eh)) + sB2[c0 + i];
                    if (c0 + 4 <= D) {
                        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                        f32x4u q = {t[0], t[1], t[2], t[3]};
                        *reinterpret_cast<f32x4u*>(op + c0) = q;
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (c0 + i < D) { op[c0 + i] = t[i]; omax = fmaxf(omax, fabsf(t[i])); }
                    }
                }
            }
        }
        if (++j == tps) {
            if (am.y) amax_commit(omax, am.y + n);
            omax = 0.f; j = 0; ++n;
            if (tile + 1 < te) sample_scales(n);
        }
    }
    if (am.y && j != 0 && tb < te) amax_commit(omax, am.y + n);
    XS_OUT;
}

static int g_pw_dump_h3k = 0;
void x6_pw_dump_from_forward_kernel(int on) { g_pw_dump_h3k = on ? 1 : 0; }

int x6_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                  long nvox, long vps, int D, int arith, const PwAmax& am, hipStream_t s, float* hdump)
{
    static std::once_flag once;
    std::call_once(once, [] {
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_x6_kernel<X6, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_x6_kernel<X6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_x6_kernel<H3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_h3k_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_fwd_h3k_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    if (vps <= 0 || vps > nvox) vps = nvox;                          // one "sample"
    if (nvox % vps || vps > 0x7fffffffL) { set_error("x6_pw_forward: nvox must be a multiple of the voxels per sample", hipSuccess); return PROBAV_EINVAL; }
    if (arith == 2) {
        if (!am.x || !am.w1 || !am.w2c || !am.b1) { set_error("x6_pw_forward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
        const size_t lds = (size_t)2 * 8 * 2 * H3::NP * 64 * 16 + (256 + 32 + 32 + 8 * 256) * sizeof(float);      // (+ pw_fwd_h3k_kernel's per-wave bias tables; two workgroups per CU still fit)
        // hdump (tests): from the 32x32x16 arrangement -- the order the reverse pass recomputes the tile in (pw_bwd_w4_kernel) -- or, after
        // x6_pw_dump_from_forward_kernel(1), from the forward kernel itself
        if (pf4_enabled() && pf4_forward_supported(nvox, vps, D) && (!hdump || g_pw_dump_h3k))          // one wave per SIMD (kernels_pf4.hip); its dump IS "from the forward kernel"
            return pf4_forward(x, w1frag, w2frag, b1, b2, dec, nvox, vps, D, am, s, hdump);
        if (hdump && g_pw_dump_h3k) hipLaunchKernelGGL(pw_fwd_h3k_kernel<true>, dim3(256 * 2), dim3(512), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag, b1, b2, dec, nvox, (int)vps, D, am, hdump);
        else if (hdump) hipLaunchKernelGGL((pw_fwd_x6_kernel<H3, true>), dim3(256 * PwfShape<H3>::WGS), dim3(64 * PwfShape<H3>::WAVES), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag,
                                      b1, b2, dec, nvox, (int)vps, D, am, hdump);
        else hipLaunchKernelGGL(pw_fwd_h3k_kernel<false>, dim3(256 * 2), dim3(512), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag, b1, b2, dec, nvox, (int)vps, D, am, nullptr);
    } else {
        const size_t lds = (size_t)2 * 8 * 2 * X6::NP * 64 * 16 + (256 + 32 + 32) * sizeof(float);
        if (hdump) hipLaunchKernelGGL((pw_fwd_x6_kernel<X6, true>), dim3(256 * PwfShape<X6>::WGS), dim3(64 * PwfShape<X6>::WAVES), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag,
                                      b1, b2, dec, nvox, (int)vps, D, am, hdump);
        else hipLaunchKernelGGL((pw_fwd_x6_kernel<X6, false>), dim3(256 * PwfShape<X6>::WGS), dim3(64 * PwfShape<X6>::WAVES), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag,
                                b1, b2, dec, nvox, (int)vps, D, am, hdump);
    }
    return check_launch("pw_fwd_x6");
}

// ---------------------------------------------------------------------------------------------------
// fused backward of expConv + ReLU + decConv (1x1x1), x6 form.  Same decomposition as pw_bwd2_mfma_kernel: one wave per
// 32-channel hidden chunk, chunk weights resident in registers (as bf16 pieces), 8 waves on one 32-voxel tile, one slab
// per workgroup.  Per tile and chunk:
//   (a) H^T  = W1c^T X^T (+ b1)          rows = hidden (registers), cols = voxel (lanes)
//   (b) dH^T = W2c dT^T                  same layout -> gate dH' = dH [H > 0], H' = relu(H): elementwise
//   (c) dX^T partial = W1c dH'^T         contracts over hidden = the ROW index: the cut accumulator registers are the B operand
//   (d) dW1c += X^T dH'                  contracts over the voxel = the COLUMN (lane) index: one in-wave transpose --
//   (e) dW2c^T += dT^T H'                the pieces are stored packed ([voxel][hidden], 8 bytes per lane and register quad) and
//                                        read back with ds_read_b64_tr_b16; X^T and dT^T come out of the staged X / dT piece
//                                        images the same way.
// 60 bf16 MFMAs per tile and chunk.  (A first x6 version produced the hidden tile in both orientations instead of
// transposing -- 84 MFMAs and three cuts per tile; it was bound by the vector issue port: 6.9 VALU instructions per MFMA.)
// ---------------------------------------------------------------------------------------------------
// the transpose image of an accumulator tile whose rows (registers) are channels and whose columns (lanes) are voxels: lane
// (voxel col, half h) owns channels 8G + 4h + (0..3) in registers 4G..4G+3, i.e. the dwords of its cut fragments in order
template <int NP>
__device__ __forceinline__ void store_pieces(unsigned char* img, int col, int h, const Frag (&f)[2][NP])
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned char* d = img + p * PT_IMG + col * PT_ROW + 8 * h;
        *reinterpret_cast<uint2*>(d) = make_uint2(f[0][p].u.x, f[0][p].u.y);          // G = 0: channels 4h ..
        *reinterpret_cast<uint2*>(d + 16) = make_uint2(f[0][p].u.z, f[0][p].u.w);     // G = 1: channels 8 + 4h ..
        *reinterpret_cast<uint2*>(d + 32) = make_uint2(f[1][p].u.x, f[1][p].u.y);     // G = 2
        *reinterpret_cast<uint2*>(d + 48) = make_uint2(f[1][p].u.z, f[1][p].u.w);     // G = 3
    }
}

// Tiles never straddle two samples (ceil(vps / 32) tiles per sample, the last one partial) and a workgroup owns a CONTIGUOUS run of
// tiles.  H3 scales are per sample: X and dT are cut with their own sample's power of two, (a)-(c) never mix samples, and the
// accumulators of (d) / (e) -- which contract over voxels, i.e. over samples too -- are carried from one sample's scales to the next
// when the run leaves a sample (an exact multiplication by a power of two; beyond 2^+-40 they are banked in the workgroup's slab at
// true scale instead), so that one accumulation only ever holds one pair of scales.
template <class AR>
__global__ __launch_bounds__(512, 2) void pw_bwd_x6_kernel(
    const float* __restrict__ x, const float* __restrict__ dT, const float* __restrict__ dOut,
    const uint4* __restrict__ w1f, const uint4* __restrict__ w2kf, const uint4* __restrict__ w1cf,
    const float* __restrict__ b1, float* __restrict__ dX, float* __restrict__ slabs, long nvox, int vps, int D, PwAmax am)
{
    constexpr int NP = AR::NP;
    constexpr int PB_TILE = NP * PB_IMG;                          // one staged tile: NP piece images
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char* XA = lds_raw;                                  // [2 buffers][NP pieces][32 voxels][80 B]
    unsigned char* DA = XA + 2 * PB_TILE;                         // same for dT (channels D..31 stay zero)
    float* TbAll = reinterpret_cast<float*>(DA + 2 * PB_TILE);   // [2 tile parities][8 waves][32][33] dX partials
    float* sB1 = TbAll + 16 * PB_TB;                              // 256 expand biases
    unsigned char* TiAll = reinterpret_cast<unsigned char*>(sB1 + 256);       // [8 waves][NP pieces][32 voxels][72 B] transpose images
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // H3 scales.  Tensors that exist only in registers are scaled from bounds: |H| <= 32 amax(x) amax(w1) + amax(b1) (as in the
    // forward kernel) and |dH| <= D amax(dT) amax(w2), per sample.  W1 as the operand of (c) is cut per cin row (its dX column).
    // A sample's scales are four exponents (scalar registers): ex, ed of X and dT, eh, eg of the hidden tile and its gradient.
    struct Sc { int ex, ed, eh, eg; };
    Sc cur = {0, 0, 0, 0}, nxt = cur;
    unsigned aw1 = 0u, aw2 = 0u, ab1 = 0u; int ew1 = 0, ew2 = 0, ew1c = 0;
    if constexpr (AR::SCALED) { aw1 = *am.w1; aw2 = *am.w2; ab1 = *am.b1; ew1 = h3_exp_w(aw1); ew2 = h3_exp_w(aw2); ew1c = h3_exp_w(am.w1r[col]); }
    auto sample_scales = [&](int n) -> Sc {
        Sc q = {0, 0, 0, 0};
        if constexpr (AR::SCALED) {
            const unsigned ax = am.x[n], ad = am.dt[n];
            q.ex = h3_exp(ax); q.ed = h3_exp(ad);
            q.eh = h3_exp(32.f * __uint_as_float(ax) * __uint_as_float(aw1) + __uint_as_float(ab1));
            q.eg = h3_exp((float)D * __uint_as_float(ad) * __uint_as_float(aw2));
        }
        return q;
    };
    const int c = wave;                                           // this wave's hidden chunk

    Frag w1[2][NP], w2[2][NP], w3[2][NP];                         // chunk-resident weight pieces
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            w1[kb][p].u = w1f[((c * 2 + kb) * NP + p) * 64 + lane];
            w2[kb][p].u = w2kf[((c * 2 + kb) * NP + p) * 64 + lane];
            w3[kb][p].u = w1cf[((c * 2 + kb) * NP + p) * 64 + lane];
        }
    // H3: two images per wave (dH' and H' pieces), so that both transposes are in LDS before (c) and the reads of (d) and (e) overlap;
    // the three-piece X6 images only fit once (the second store then waits for the reads of (d))
    constexpr int NTI = AR::SCALED ? 2 : 1;
    unsigned char* Ti = TiAll + wave * NTI * NP * PT_IMG;
    unsigned char* Th = Ti + (NTI - 1) * NP * PT_IMG;
    float omax = 0.f;
    f32x16 dW1, dW2t;
    float bs1v[16];                                               // db1 partial of (hidden rowmap(r, half), this lane's voxels)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW1[r] = 0.f; dW2t[r] = 0.f; bs1v[r] = 0.f; }

    for (int i = tid; i < 2 * PB_TILE / 16; i += 512) reinterpret_cast<uint4*>(DA)[i] = make_uint4(0u, 0u, 0u, 0u);

    const int tps = (vps + 31) >> 5;                              // tiles per sample
    const int ntiles = (int)(nvox / vps) * tps;
    const int tbeg = (int)((long)ntiles * blockIdx.x / gridDim.x), tend = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);   // this workgroup's run of tiles
    // (a contiguous run also measures 5 % faster than the round-1 order tile = block + k * grid)
    // Staging belongs to the first-dispatched half of the workgroup (waves 0..3); the dX reduction is shared by both halves.  With the
    // staging spread over all eight waves the second half (which loses the per-SIMD issue arbitration) was the pole wave of every tile
    // while the first half waited ~20 % of a tile at the barrier; with staging AND reduction on the first half the roles flipped.
    // Thread t < 256 moves one float4 of the X tile and the dT elements f = t + 256 k (k < 4, f < 32 D; element f = voxel * D + out);
    // clamped unconditional loads, zero selected afterwards.
    const bool stager = wave < 4;                                     // wave-uniform
    int dvk[4], dok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int f = tid + 256 * k; dvk[k] = f / D; dok[k] = f - dvk[k] * D; }
    float bs2[4] = {0.f, 0.f, 0.f, 0.f};
    // (raw loads only, clamped and unconditional: the elements beyond a partial tile are zeroed when the tile is stored -- a select right
    // behind a load makes the compiler wait for every load separately)
    auto stage_load = [&](int sn, int sj, float4& xv, float (&d)[4]) {
        if (!stager) return;
        const long v0 = (long)sn * vps + 32 * sj;
        const int nrem = vps - 32 * sj < 32 ? vps - 32 * sj : 32;
        const int vv = tid >> 3;
        xv = reinterpret_cast<const float4*>(x + (v0 + (vv < nrem ? vv : 0)) * 32)[tid & 7];
        const float* dt0 = dT + v0 * D;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = tid + 256 * k;
            d[k] = dt0[f < nrem * D ? f : 0];
        }
    };
    auto stage_store = [&](int buf, int sj, const float4& xv, const float (&d)[4], float sx, float sd) {
        if (!stager) return;
        const int nrem = vps - 32 * sj < 32 ? vps - 32 * sj : 32;
        {
            const bool xl = (tid >> 3) < nrem;
            unsigned a[NP], b[NP];
            cut_pair<AR>(xl ? xv.x : 0.f, xl ? xv.y : 0.f, sx, a);
            cut_pair<AR>(xl ? xv.z : 0.f, xl ? xv.w : 0.f, sx, b);
            unsigned char* dst = XA + buf * PB_TILE + (tid >> 3) * PB_ROW + (tid & 7) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(dst + p * PB_IMG) = make_uint2(a[p], b[p]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dv = tid + 256 * k < nrem * D ? d[k] : 0.f;
            if (tid + 256 * k < 32 * D) {
                unsigned short q[NP];
                cut_one<AR>(dv, sd, q);
                unsigned char* dst = DA + buf * PB_TILE + dvk[k] * PB_ROW + dok[k] * 2;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<unsigned short*>(dst + p * PB_IMG) = q[p];
            }
            bs2[k] += dv;
        }
    };
    // one slab per workgroup: [dW1 32x256 | dW2 256xD | db1 256 | db2 D]; the accumulators of (d), (e) and the db1 sums are added to it
    // (at true scale) whenever the run leaves a sample, and at the end; the first flush stores
    const long slab_floats = 8192 + 256 * (long)D + 256 + D;
    float* sl = slabs + (long)blockIdx.x * slab_floats;
    bool flushed = false;                                             // wave-uniform
    auto flush = [&](const Sc& q) {
        asm volatile("" ::: "memory");                                // a rare path: nothing of it may be speculated into the tile loop
        float a1[16], a2[16], a3[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a1[r] = dW1[r]; a2[r] = dW2t[r];
            if constexpr (AR::SCALED) { a1[r] = ldexpf(a1[r], -(q.ex + q.eg)); a2[r] = ldexpf(a2[r], -(q.ed + q.eh)); }
            float v = bs1v[r];                                        // db1[hidden] = sum over the voxel lanes: butterfly inside each 32-lane half (fixed order)
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
            if constexpr (AR::SCALED) v = ldexpf(v, -q.eg);
            a3[r] = v;
            dW1[r] = 0.f; dW2t[r] = 0.f; bs1v[r] = 0.f;
        }
        if (flushed) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = rowmap(r, half);
                a1[r] += sl[(long)rw * 256 + 32 * c + col];
                if (rw < D) a2[r] += sl[8192 + (long)(32 * c + col) * D + rw];
                if (col == 0) a3[r] += sl[8192 + 256 * (long)D + 32 * c + rw];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rw = rowmap(r, half);
            sl[(long)rw * 256 + 32 * c + col] = a1[r];                                        // [cin][hidden]
            if (rw < D) sl[8192 + (long)(32 * c + col) * D + rw] = a2[r];                     // [hidden][out]
            if (col == 0) sl[8192 + 256 * (long)D + 32 * c + rw] = a3[r];
        }
        flushed = true;
    };

    int tile = tbeg;
    int n = tbeg / tps, j = tbeg - n * tps;                           // (sample, tile inside it) of `tile`
    int buf = 0;
    XS_DECL;
#ifdef PROBAV_STAMP2
    unsigned long long xs2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xs2t = 0;
#endif
    __syncthreads();                                   // DA pads are zero
    if (tile < tend) cur = sample_scales(n);
    if (tid < 256) sB1[tid] = b1[tid] * pow2i(cur.eh);            // the expand biases at the hidden tile's scale of the run's first sample (rewritten when the run enters another sample)
    if (tile < tend) {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f); float d[4] = {0.f, 0.f, 0.f, 0.f};
        stage_load(n, j, xv, d);
        stage_store(0, j, xv, d, pow2i(cur.ex), pow2i(cur.ed));
    }
    // dX of a tile = dOut + the eight chunk partials.  The partials of tile t are reduced during iteration t+1 (after the one barrier
    // per tile), from the buffer of t's parity, while iteration t+1 fills the other one.  Waves w and w+4 reduce voxels 8(w&3) .. 8(w&3)+7.
    const int rk0 = wave < 4 ? 0 : 2;                                 // waves w and w+4 share the 8 voxels 8(w&3)..: two of the four lane slices each
    long pv0 = -1; int pnrem = 0, pn = -1, pkdx = 0;                  // previous tile: first voxel, live voxels, sample, exponent of its dX partials
    float pdo[4] = {0.f, 0.f, 0.f, 0.f};                              // ... and its dOut values
    auto reduce_prev = [&](int pb) {
        if (pv0 < 0) return;
        // All sixteen partials are requested before the first is used, and the sums are pinned in front of the store's branch: left alone,
        // the compiler sinks the reads into the branch and strings them out as read, wait, add, read, wait, add ... -- sixteen LDS round trips
        // in a row per tile and wave.
        float t[2][8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int rv = 8 * (wave & 3) + (lane >> 5) + 2 * (k + rk0);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t[k][jj] = TbAll[(pb * 8 + jj) * PB_TB + rv * 33 + col];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) asm volatile("" : "+v"(t[k][jj]));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int rv = 8 * (wave & 3) + (lane >> 5) + 2 * (k + rk0);
            float sacc;
            if constexpr (AR::SCALED) {
                sacc = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) sacc += t[k][jj];
                sacc = ldexpf(sacc, pkdx) + pdo[k];
            } else {
                sacc = pdo[k];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) sacc += t[k][jj];
            }
            asm volatile("" : "+v"(sacc));
            if (rv < pnrem) { dX[(pv0 + rv) * 32 + col] = sacc; omax = fmaxf(omax, fabsf(sacc)); }
        }
    };
    float omax_done = 0.f; int n_done = -1;             // a finished sample's largest |dX| (per lane) waiting to be committed
    auto commit_done = [&]() {
        if (n_done >= 0 && am.y) amax_commit(omax_done, am.y + n_done);
        n_done = -1;
    };
    // One tile.  Two instances: the INTERIOR one (the next tile exists and belongs to the same sample: no scale change, no bookkeeping) is
    // the hot loop; the BOUNDARY one (last tile of a sample inside the run, or of the run) carries everything that happens once per sample.
    auto do_tile = [&](auto boundary_tag) {
        constexpr bool BND = decltype(boundary_tag)::value;
        XS_ACC(5);
        __syncthreads();                               // tile `tile` is staged in buffer `buf`; the previous tile's partials are complete
        XS_ACC(1);
        float* Tb = TbAll + (buf * 8 + wave) * PB_TB;
        // the next tile of the run: (nn, nj); its sample's scales are needed for the staging at the end of this iteration
        int nn = n, nj = j + 1;
        bool has_next = true, leaves = false;
        if constexpr (BND) {
            if (nj == tps) { nj = 0; ++nn; }
            has_next = tile + 1 < tend;
            leaves = nn != n;                                          // this is the run's last tile of sample n (wave-uniform)
        }
        float4 nxv = make_float4(0.f, 0.f, 0.f, 0.f); float nd[4] = {0.f, 0.f, 0.f, 0.f};
        float ch = 1.f, cg = 1.f;                                      // accumulators (a), (b) -> hidden values / hidden gradients at their own scales
        if constexpr (AR::SCALED) {
            const int kh = cur.eh - cur.ex - ew1, kg = cur.eg - ew2 - cur.ed;   // (<= -17 always, see the bounds)
            ch = pow2i(kh < -126 ? -126 : kh); cg = pow2i(kg < -126 ? -126 : kg);
        }
        const long v0 = (long)n * vps + 32 * j;
        const int nrem = vps - 32 * j < 32 ? vps - 32 * j : 32;
        float cdo[4] = {0.f, 0.f, 0.f, 0.f};                           // dOut of this tile's voxels, consumed by the next iteration's reduction
        const unsigned char* Xb = XA + buf * PB_TILE;
        const unsigned char* Db = DA + buf * PB_TILE;
        Frag xf[2][NP], df[2][NP];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                xf[kb][p].u = *reinterpret_cast<const uint4*>(Xb + p * PB_IMG + col * PB_ROW + kb * 32 + half * 16);
                df[kb][p].u = *reinterpret_cast<const uint4*>(Db + p * PB_IMG + col * PB_ROW + kb * 32 + half * 16);
            }
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#ifdef PROBAV_STAMP2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        xs2t = __builtin_amdgcn_s_memtime();
#endif
        {
            f32x16 H = zero, dH = zero, dx = zero;
            H = mac<AR>(w1[0], xf[0], H); H = mac<AR>(w1[1], xf[1], H);               // (a)
            dH = mac<AR>(w2[0], df[0], dH); dH = mac<AR>(w2[1], df[1], dH);           // (b)
            __builtin_amdgcn_sched_barrier(0);
            reduce_prev(buf ^ 1);                          // LDS reads, adds and two stores in the shadow of the 24 MFMAs just issued
            if (pn >= 0 && pn != n) {                      // the previous tile was the last one of its sample: its dX is complete now; the
                omax_done = omax; n_done = pn; omax = 0.f; // wave-wide maximum and the atomic wait for the next boundary tile (rare code stays out of this loop)
            }
            __builtin_amdgcn_sched_barrier(0);
            // The next tile's X / dT rows and this tile's dOut values are requested HERE, behind the dX stores of the previous tile: issued
            // at the top of the tile they sat in front of those stores in the in-order vmcnt queue and the wave waited for them there.
            stage_load(has_next ? nn : n, has_next ? nj : j, nxv, nd);     // in flight during the rest of this tile (after the last tile: a harmless re-read)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rv = 8 * (wave & 3) + (lane >> 5) + 2 * (k + rk0);
                cdo[k] = dOut[(v0 + (rv < nrem ? rv : 0)) * 32 + col];
            }
            __builtin_amdgcn_sched_barrier(0);
            XS2(0);
            Frag gf[2][NP], hf[2][NP];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float gs[8], hs[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {                          // registers 4G .. 4G+3 <-> hidden 32c + 8G + 4h + (0..3)
                    const float4 bb = *reinterpret_cast<const float4*>(sB1 + 32 * c + 8 * (2 * kb + g) + 4 * half);
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 8 * kb + 4 * g + i;
                        const float hv = fmaf(H[r], ch, bv[i]), dv = dH[r] * cg;              // (sB1 holds b1 2^eh of the current sample; ch = cg = 1 without scaling)
                        gs[4 * g + i] = hv > 0.f ? dv : 0.f;
                        hs[4 * g + i] = fmaxf(hv, 0.f);
                        bs1v[r] += gs[4 * g + i];
                    }
                }
                cut8_scaled<AR>(gs, gf[kb]);
                cut8_scaled<AR>(hs, hf[kb]);
            }
            XS2(1);
            // dH' (and, with two images, H') pieces go to this wave's transpose images before (c), whose MFMAs cover the LDS writes.  LDS
            // operations of one wave execute in order; the empty asm statements only keep the COMPILER from moving reads above the
            // writes they depend on.
            store_pieces<NP>(Ti, col, half, gf);
            if constexpr (NTI == 2) store_pieces<NP>(Th, col, half, hf);
            asm volatile("" ::: "memory");
            dx = mac<AR>(w3[0], gf[0], dx); dx = mac<AR>(w3[1], gf[1], dx);           // (c)
            XS2(2);
#pragma unroll
            for (int r = 0; r < 16; ++r) Tb[col * 33 + rowmap(r, half)] = dx[r];
            if constexpr (NTI == 2) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {                               // (d) and (e) side by side: four transposed reads per piece, then both products
                    Frag at[NP], bt[NP], ae[NP], be[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        tr_frag<PB_ROW>(Xb + p * PB_IMG, lane, kb, at[p]); tr_frag<PT_ROW>(Ti + p * PT_IMG, lane, kb, bt[p]);
                        tr_frag<PB_ROW>(Db + p * PB_IMG, lane, kb, ae[p]); tr_frag<PT_ROW>(Th + p * PT_IMG, lane, kb, be[p]);
                    }
                    dW1 = mac<AR>(at, bt, dW1);                                // dW1c[cin][hidden] += X^T dH'
                    dW2t = mac<AR>(ae, be, dW2t);                              // dW2c^T[out][hidden] += dT^T H'
                }
                XS2(3);
            } else {
                Frag at[NP], bt[NP];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) { tr_frag<PB_ROW>(Xb + p * PB_IMG, lane, kb, at[p]); tr_frag<PT_ROW>(Ti + p * PT_IMG, lane, kb, bt[p]); }
                    dW1 = mac<AR>(at, bt, dW1);                                // dW1c[cin][hidden] += X^T dH'
                }
                XS2(3);
                asm volatile("" ::: "memory");
                store_pieces<NP>(Ti, col, half, hf);                            // (e): H' pieces, same image
                asm volatile("" ::: "memory");
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) { tr_frag<PB_ROW>(Db + p * PB_IMG, lane, kb, at[p]); tr_frag<PT_ROW>(Ti + p * PT_IMG, lane, kb, bt[p]); }
                    dW2t = mac<AR>(at, bt, dW2t);                              // dW2c^T[out][hidden] += dT^T H'
                }
            }
            asm volatile("" ::: "memory");
            XS2(4);
        }
        XS_ACC(2);
        nxt = cur;
        if constexpr (AR::SCALED && BND) {
            commit_done();
            if (leaves && has_next) {
                // (d), (e) of the next tile run at another sample's scales.  The running sums move to the new scales by an exact multiplication
                // with a power of two (fp32 keeps 2^+-40 around sums of ~2^45 without leaving its range); a jump beyond that -- a dead sample
                // next to a bright one -- banks the sums in the slab instead and starts over.
                nxt = sample_scales(nn);
                const int d1 = (nxt.ex + nxt.eg) - (cur.ex + cur.eg), d2 = (nxt.ed + nxt.eh) - (cur.ed + cur.eh), d3 = nxt.eg - cur.eg;
                const int big = max(max(d1 < 0 ? -d1 : d1, d2 < 0 ? -d2 : d2), d3 < 0 ? -d3 : d3);
                if (big > 40) flush(cur);
                else if (big != 0) {
                    const float f1 = pow2i(d1), f2 = pow2i(d2), f3 = pow2i(d3);
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dW1[r] *= f1; dW2t[r] *= f2; bs1v[r] *= f3; }
                }
                if (nxt.eh != cur.eh) {                    // the biases at the next sample's hidden scale: every wave is past its last read of
                    __syncthreads();                       // sB1 for this tile here, and the next tile's first barrier publishes the new values
                    if (tid < 256) sB1[tid] = b1[tid] * pow2i(nxt.eh);
                }
            }
        }
        if (has_next) stage_store(buf ^ 1, nj, nxv, nd, pow2i(nxt.ex), pow2i(nxt.ed));
        XS_ACC(3);
        pv0 = v0; pnrem = nrem; pn = n;
        if constexpr (AR::SCALED) pkdx = -(ew1c + cur.eg);              // (c) partials -> true values: W1's row `col` and this sample's dH scale
#pragma unroll
        for (int k = 0; k < 4; ++k) pdo[k] = cdo[k];
        cur = nxt; n = nn; j = nj;
    };
    while (tile < tend) {
        int seg_last = (n + 1) * tps - 1;                              // last tile of sample n inside this run
        if (seg_last > tend - 1) seg_last = tend - 1;
        for (; tile < seg_last; ++tile, buf ^= 1) do_tile(std::false_type());
        do_tile(std::true_type());
        ++tile; buf ^= 1;
    }
    __syncthreads();
    reduce_prev(buf ^ 1);
    commit_done();
    if (am.y && pn >= 0) amax_commit(omax, am.y + pn);
    flush(cur);                                            // the sums of the run's last sample(s), at that sample's scales (all exponents 0 without scaling)
    // db2[out] = sum of the staged dT values: thread t always staged out (t % D) and ((t + 512) % D); fixed-order sum
    __syncthreads();
    float* R = TbAll;                                                 // R[f] = column sum of staged element f (f < 1024)
    if (stager) {
#pragma unroll
        for (int k = 0; k < 4; ++k) R[tid + 256 * k] = (tid + 256 * k < 32 * D) ? bs2[k] : 0.f;
    }
    __syncthreads();
    if (tid < D) {
        float t = 0.f;
        for (int jj = tid; jj < 1024; jj += D) t += R[jj];
        sl[8192 + 256 * (long)D + 256 + tid] = t;
    }
    XS_ACC(6);
    XS_OUT;
#ifdef PROBAV_STAMP2
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) for (int k_ = 0; k_ < 5; ++k_) g_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 1 + k_] = xs2[k_];
#endif
}


int x6_pw_backward(const float* x, const float* dT, const float* dOut, const float* w1f, const float* w2kf, const float* w1cf,
                   const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2, float* slabs, long nvox, long vps, int D,
                   int arith, const PwAmax& am, hipStream_t s)
{
    static std::once_flag once;
    std::call_once(once, [] {
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_bwd_x6_kernel<X6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_bwd_x6_kernel<H3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    if (vps <= 0 || vps > nvox) vps = nvox;
    if (nvox % vps || vps > 0x7fffffffL) { set_error("x6_pw_backward: nvox must be a multiple of the voxels per sample", hipSuccess); return PROBAV_EINVAL; }
    if (arith == 2) {
        if (!am.x || !am.w1 || !am.w2 || !am.b1 || !am.dt || !am.w1r) { set_error("x6_pw_backward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
        // round 5: one wave per SIMD, 512 registers, no barrier in the tile loop (kernels_pw4.hip).  pw_bwd_x6_kernel<H3> below is the general form: any
        // batch, any D <= 32 (and PROBAV_GEN1=1 / pw4_set_enabled(0) for A/B runs); rounds 3 and 4's schedules of it (pw_bwd_h3s / h3t) left with round 5
        if (pw4_enabled() && pw4_backward_supported(nvox, vps, D))
            return pw4_backward(x, dT, dOut, w1f, w2kf, w1cf, b1, dX, dW1, dW2, db1, db2, slabs, nvox, vps, D, am, s);
        const size_t lds = (size_t)4 * H3::NP * PB_IMG + ((size_t)16 * PB_TB + 256) * sizeof(float) + (size_t)8 * 2 * H3::NP * PT_IMG;   // (two transpose images per wave)
        hipLaunchKernelGGL(pw_bwd_x6_kernel<H3>, dim3(mfma_pw_backward_grid()), dim3(512), lds, s, x, dT, dOut, (const uint4*)w1f,
                           (const uint4*)w2kf, (const uint4*)w1cf, b1, dX, slabs, nvox, (int)vps, D, am);
    } else {
        const size_t lds = (size_t)4 * X6::NP * PB_IMG + ((size_t)16 * PB_TB + 256) * sizeof(float) + (size_t)8 * X6::NP * PT_IMG;
        hipLaunchKernelGGL(pw_bwd_x6_kernel<X6>, dim3(mfma_pw_backward_grid()), dim3(512), lds, s, x, dT, dOut, (const uint4*)w1f,
                           (const uint4*)w2kf, (const uint4*)w1cf, b1, dX, slabs, nvox, (int)vps, D, am);
    }
    int rc = check_launch("pw_bwd_x6");
    if (rc) return rc;
    return mfma_pw_backward_reduce(slabs, D, dW1, dW2, db1, db2, s);
}

// ---------------------------------------------------------------------------------------------------
// backward-filter of the 'same' 3x3x3 convolution, x6 form:  dW[tap][ci][co] = sum_v X[v + tap][ci] dY[v][co]
// One GEMM per tap with M = ci (one 32-row tile; rows >= Cin are discarded), N = co, K = the voxels of one output
// row in blocks of 16.  Per workgroup a three-row ring of the input lives in LDS as bf16 PIECE IMAGES
// ([voxel][piece][channels], pads zero), cut once when a row is staged; ds_read_b64_tr_b16 then hands the
// A operand (K = voxel) straight to the MFMA -- four voxel rows per read, each row's address supplied by its own
// lanes, so the tap shift and the row wrap are address arithmetic.  The inner loop is 6 transposed reads + 6 MFMAs
// per tap and k-block and has no VALU work beyond two address adds.  dY comes from HBM/L2 as the B operand
// (8 coalesced loads per lane and k-block), cut into pieces once per k-block and reused by the wave's 7 taps.
// 8 waves: wave w owns taps (w & 3) + 4j (accumulators: 7 x 16 registers) and the k-blocks of parity w >> 2; the two
// parities are added in a fixed order at the end.  Workgroups walk contiguous (patch, row) runs, so consecutive rows
// re-stage one row.  One slab per workgroup, summed by slab_sum_batch_kernel (fp64, fixed order).
// ---------------------------------------------------------------------------------------------------
struct WgArgs {
    int N, H, W, T, Cout;       // OUTPUT extents (rows, columns, depth)
    int Hi, Wi, Ti;             // input extents
    int ph, pw, pt, reflect;    // pads (0 or 1 each); reflect: H/W pads mirror the input (tf.pad REFLECT) instead of zeros
    int Wp, Tp, nv, total_tiles;
    int nsplit, Wt;             // output rows are cut into nsplit column ranges of Wt columns when three full rows do not fit the LDS
    int tab;                    // byte offset of the staging table in LDS (nsplit == 1 and room for it), 0 = none
    unsigned mT, mTi;           // ceil(2^32 / T), ceil(2^32 / Ti): floor(v / d) = umulhi(v, m) for the small v used here (d >= 2)
};

__device__ __forceinline__ int wg_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <int CIN, bool GATE, class AR>
__global__ __launch_bounds__(512, 2) void conv3_wgrad_x6_kernel(WgArgs a, const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ gate, float* __restrict__ partial,
                                                               float* __restrict__ partial_b, Amax am)
{
    XS_ENTRY;
    constexpr int NPC = AR::NP;                    // pieces per value
    constexpr int CB = CIN <= 28 ? 56 : 64;        // bytes of one piece of one voxel (channels padded to 28 / 32)
    constexpr int VS = NPC * CB;                   // bytes per voxel
    constexpr int NP = (CIN + 1) / 2;              // channel pairs staged per voxel
    float sx = 1.f, sd = 1.f; int kun = 0;         // H3: scales of x (am.x) and dY (am.w), exponent that undoes both
    // (the contraction runs over the voxels of ALL samples, so both operands take ONE scale: that of their largest sample)
    if constexpr (AR::SCALED) { const int ex = h3_exp(amax_over_samples(am.x, a.N)), ed = h3_exp(amax_over_samples(am.w, a.N)); sx = pow2i(ex); sd = pow2i(ed); kun = -(ex + ed); }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int li = lane & 15, gcol = (lane >> 4) & 1;
    const int tg = wave & 3, ksel = wave >> 2;
    const int rowbytes = a.Wp * a.Tp * VS;
    // H3: the tile's dY row lives in LDS as a piece image too ([voxel][piece][32 channels], 128 B per voxel, rows beyond the tile zero), staged
    // and cut ONCE per tile by the whole workgroup; the B operand of a k-block is then four transposed reads.  (Before, each of the four
    // waves of a k-block parity loaded the same 16 x 32 block from memory and cut it for itself.)
    constexpr bool DYI = AR::SCALED;
    const int dyrows = (a.nv + 15) & ~15;                  // x6_wgrad_split(): dyrows * 8 <= 512 * NDY
    unsigned char* dyimg = lds_raw + ((3 * rowbytes + 16 + 15) & ~15);
    const int zbytes = DYI ? ((3 * rowbytes + 16 + 15) & ~15) + dyrows * 128 : 3 * rowbytes + 16;
    for (int i = tid; i < zbytes / 8; i += 512) reinterpret_cast<uint2*>(lds_raw)[i] = make_uint2(0u, 0u);

    // M tiles: the GEMM's rows are the (tap, input channel) pairs, RT rows per tap.  With 25 input channels the H3 form packs them (28
    // rows per tap = the image's padded channel count, a multiple of the 4-channel granule of a transposed read): 27 x 28 = 756 rows =
    // 24 tiles instead of 27 tap tiles of which 7 rows in 32 are padding -- 14 % fewer MFMAs and A-operand reads.  Wave (tg, ksel) owns tiles tg + 4 j.
    constexpr int RT = (AR::SCALED && CIN == 25) ? 28 : 32;
    constexpr int NT = (27 * RT + 31) / 32, NJ = (NT + 3) / 4;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    // staging of one input row, split in two so that the HBM/L2 latency of the NEXT tile's row hides under this tile's MFMAs:
    // stage_load leaves the row in registers (NST channel pairs per thread), stage_store cuts and stores it.
    constexpr int NST = CIN <= 28 ? 6 : 7;                 // x6_wgrad_supported(): (Wt + 2) * Ti * NP <= 512 * NST
    // a tile = (patch n, column range sp, output row ho).  Ring row `key` = ho + dh holds input row key - ph (mirrored / zero outside),
    // local column lw <-> input column ws0 + lw - pw, local depth lt <-> input depth lt - pt (depth pads stay zero from the init).
    // meta[k]: what stage_store needs of the item's index arithmetic, so that it is done once per item and not twice -- bits 0..23 the byte offset of the
    // item's pair inside a ring row, bit 29: the pair's second channel exists, bit 30: the item exists, bit 31: inside the patch (row and column)
    // With one column range per row (nsplit == 1) a thread's items are the same for every row: their index arithmetic is done ONCE, at the start of the
    // kernel, and kept in LDS ([NST][512] pairs: source offset inside an input row, meta without the row's validity) -- six 8-byte reads per row instead of
    // some 25 vector instructions per item.
    int2* stab = a.tab ? reinterpret_cast<int2*>(lds_raw + a.tab) : nullptr;
    if (stab) {
        const int items = (a.W + 2) * a.Ti * NP;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + 512 * k;
            const int ic = i < items ? i : 0;
            const int vox = ic / NP, cp = ic - vox * NP;
            const int lw = (int)__umulhi((unsigned)vox, a.mTi), t = vox - lw * a.Ti;
            int iw = lw - a.pw;
            if (a.reflect) iw = wg_reflect(iw, a.Wi);
            const bool cok = iw >= 0 && iw < a.Wi;
            const int c0 = 2 * cp;
            stab[k * 512 + tid] = make_int2(((cok ? iw : 0) * a.Ti + t) * CIN + c0,
                                            ((lw * a.Tp + t + a.pt) * VS + cp * 4) | (c0 + 1 < CIN ? 1 << 29 : 0) | (i < items ? 1 << 30 : 0) | (cok ? (int)0x80000000u : 0));
        }
    }
    auto stage_load = [&](int n, int key, int ws0, int Wts, float (&f)[NST][2], int (&meta)[NST]) {
        int ih = key - a.ph;
        if (a.reflect) ih = wg_reflect(ih, a.Hi);
        const bool rok = ih >= 0 && ih < a.Hi;
        const float* src = x + ((long)n * a.Hi + (rok ? ih : 0)) * (long)a.Wi * a.Ti * CIN;
        if (stab) {
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int2 c = stab[k * 512 + tid];
                meta[k] = rok ? c.y : c.y & 0x7fffffff;
                const float f0 = src[c.x], f1 = src[c.x + ((c.y >> 29) & 1)];
                if constexpr (CIN == 25) { f[k][0] = f0; f[k][1] = f1; }
                else { const bool ok = meta[k] < 0; f[k][0] = ok ? f0 : 0.f; f[k][1] = ok ? f1 : 0.f; }
            }
            return;
        }
        const int items = (Wts + 2) * a.Ti * NP;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = tid + 512 * k;
            const int ic = i < items ? i : 0;
            const int vox = ic / NP, cp = ic - vox * NP;
            const int lw = (int)__umulhi((unsigned)vox, a.mTi), t = vox - lw * a.Ti;
            int iw = ws0 + lw - a.pw;
            if (a.reflect) iw = wg_reflect(iw, a.Wi);
            const bool ok = rok && iw >= 0 && iw < a.Wi;
            const int c0 = 2 * cp, c1 = c0 + 1 < CIN ? c0 + 1 : c0;
            const long o = ((long)(ok ? iw : 0) * a.Ti + t) * CIN;
            meta[k] = ((lw * a.Tp + t + a.pt) * VS + cp * 4) | (c0 + 1 < CIN ? 1 << 29 : 0) | (i < items ? 1 << 30 : 0) | (ok ? (int)0x80000000u : 0);
            // raw values (clamped addresses): what must be zero is zeroed in stage_store -- a select right behind the load makes the wave
            // wait for the load where it is issued, i.e. at the top of the tile instead of under its MFMAs
            // (32 input channels -- the reducers -- keep the select here: one more item per thread, and the registers do not stretch to it)
            if constexpr (CIN == 25) { f[k][0] = src[o + c0]; f[k][1] = src[o + c1]; }
            else { const float f0 = src[o + c0], f1 = src[o + c1]; f[k][0] = ok ? f0 : 0.f; f[k][1] = ok ? f1 : 0.f; }
        }
    };
    auto stage_store = [&](int key, const float (&f)[NST][2], const int (&meta)[NST]) {
        unsigned char* slot = lds_raw + (key % 3) * rowbytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            if (meta[k] & (1 << 30)) {
                const bool ok = meta[k] < 0;
                unsigned q[NPC];
                if constexpr (CIN == 25) cut_pair<AR>(ok ? f[k][0] : 0.f, (ok && (meta[k] & (1 << 29))) ? f[k][1] : 0.f, sx, q);
                else cut_pair<AR>(f[k][0], f[k][1], sx, q);
                unsigned char* d = slot + (meta[k] & 0xffffff);
#pragma unroll
                for (int p = 0; p < NPC; ++p) *reinterpret_cast<unsigned*>(d + p * CB) = q[p];
            }
        }
    };
    auto stage_three = [&](int n, int ho, int ws0, int Wts) {
#pragma unroll 1
        for (int rr = 0; rr < 3; ++rr) {
            float f[NST][2];
            int mt[NST];
            stage_load(n, ho + rr, ws0, Wts, f, mt);
            stage_store(ho + rr, f, mt);
        }
    };
    auto decode = [&](int tile, int& n, int& ws0, int& Wts, int& ho) {     // tile = (n * nsplit + sp) * H + ho
        const int q = tile / a.H;
        ho = tile - q * a.H;
        n = q / a.nsplit;
        const int sp = q - n * a.nsplit;
        ws0 = sp * a.Wt;
        Wts = a.W - ws0 < a.Wt ? a.W - ws0 : a.Wt;
    };

    constexpr int NDY = 4;
    float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f);         // H3: bias gradient of channels 4 (tid & 7) .. + 3, summed as the rows are staged
    auto dy_load = [&](int n, int ho, int ws0, int Wts, float4 (&d)[NDY]) {
        const int ni = Wts * a.T * 8;                      // (voxel, channel quad) items of the row range
        const long ob = (((long)n * a.H + ho) * a.W + ws0) * a.T * 32;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int i = tid + 512 * k;
            const int ic = i < ni ? i : 0;
            d[k] = *reinterpret_cast<const float4*>(dy + ob + ic * 4);
            if constexpr (GATE) {
                const float4 m = *reinterpret_cast<const float4*>(gate + ob + ic * 4);       // the layer's own output: ReLU mask of dY
                d[k].x = m.x > 0.f ? d[k].x : 0.f; d[k].y = m.y > 0.f ? d[k].y : 0.f; d[k].z = m.z > 0.f ? d[k].z : 0.f; d[k].w = m.w > 0.f ? d[k].w : 0.f;
            }
        }
    };
    auto dy_store = [&](int Wts, const float4 (&d)[NDY]) {
        const int ni = Wts * a.T * 8;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int i = tid + 512 * k;
            if (i < dyrows * 8) {
                const bool live = i < ni;
                const float4 v = live ? d[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                bs4.x += v.x; bs4.y += v.y; bs4.z += v.z; bs4.w += v.w;
                unsigned q0[NPC], q1[NPC];
                cut_pair<AR>(v.x, v.y, sd, q0);
                cut_pair<AR>(v.z, v.w, sd, q1);
                unsigned char* dst = dyimg + (i >> 3) * 128 + (i & 7) * 8;
#pragma unroll
                for (int p = 0; p < NPC; ++p) *reinterpret_cast<uint2*>(dst + p * 64) = make_uint2(q0[p], q1[p]);
            }
        }
    };

    const int tbeg = (int)((long)blockIdx.x * a.total_tiles / gridDim.x), tend = (int)((long)(blockIdx.x + 1) * a.total_tiles / gridDim.x);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    XS_DECL;
    if (tbeg < tend) {
        // the first tile's three rows and its dY row are requested together and BEFORE the barrier that ends the zeroing of the ring (a thread reads
        // only its own entries of the staging table): one memory latency in the prologue instead of four in a row
        int n, ws0, Wts, ho; decode(tbeg, n, ws0, Wts, ho);
        float f3[3][NST][2];
        int mt3[3][NST];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) stage_load(n, ho + rr, ws0, Wts, f3[rr], mt3[rr]);
        float4 d0[NDY];
        if constexpr (DYI) dy_load(n, ho, ws0, Wts, d0);
        __syncthreads();                                   // ring zeroed
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) stage_store(ho + rr, f3[rr], mt3[rr]);
        if constexpr (DYI) dy_store(Wts, d0);
    } else {
        __syncthreads();
    }
    XS_ACC(1);
#pragma unroll 1
    for (int tile = tbeg; tile < tend; ++tile) {
        int n, ws0, Wts, ho;
        decode(tile, n, ws0, Wts, ho);
        const int nv = Wts * a.T, nkb = (nv + 15) >> 4;
        __syncthreads();                                   // this tile's three rows are staged
        XS_ACC(2);
        const bool has_next = tile + 1 < tend;
        int nn = n, nws0 = ws0, nWts = Wts, nho = ho;
        if (has_next) decode(tile + 1, nn, nws0, nWts, nho);
        const bool consecutive = has_next && nn == n && nws0 == ws0 && nho == ho + 1;    // next tile = next row of the same column range
        // Nothing of the previous tile is in flight here; saying so explicitly lets the compiler's wait-count model forget loads it still
        // counts as possibly pending from the loop's back edge (it otherwise puts a vmcnt(0) in front of the k-loop's first transposed
        // read, which makes the prefetches below synchronous).
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        float nf[NST][2];
        int nmeta[NST];
        if (consecutive) stage_load(n, ho + 3, ws0, Wts, nf, nmeta);   // ring row of the next tile's dh = 2, in flight during this tile's MFMAs
        float4 ndy[NDY];
        if constexpr (DYI) { if (has_next) dy_load(nn, nho, nws0, nWts, ndy); }       // the next tile's dY row likewise
        const long out_base = (((long)n * a.H + ho) * a.W + ws0) * a.T;
        int tapoff[NJ];                                                                // per lane: its four rows of tile j = channels ci0 .. ci0 + 3 of one tap
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int T = tg + 4 * j < NT ? tg + 4 * j : NT - 1;                      // (a slot beyond the last tile: harmless reread)
            const int R0 = 32 * T + 16 * gcol + 4 * (li & 3);
            int tap = R0 / RT;
            const int ci0 = R0 - tap * RT;
            tap = tap < 27 ? tap : 26;                                                 // (rows beyond the last tap are discarded at the end)
            const int dh = tap / 9, dw = (tap / 3) % 3, dt = tap % 3;
            tapoff[j] = ((ho + dh) % 3) * rowbytes + (dw * a.Tp + dt) * VS + ci0 * 2; // ring row ho + dh
        }
        // B operand: dY[voxel 16kb + 8h + j][co = col].  Three-stage pipeline over this wave's k-blocks: loads of block i+2 |
        // cutting block i+1 into pieces and its transposed-read addresses | MFMAs of block i.  The second stage is spread over
        // the seven taps of the third so that its VALU work issues in the shadow of the MFMAs.
        const float* dyrow = dy + out_base * 32 + col;
        const float* gtrow = GATE ? gate + out_base * 32 + col : nullptr;       // the layer's own output: ReLU mask of dY
        auto load_dy = [&](int kb, float (&r)[8]) {
            const int v0 = 16 * kb + 8 * h;
            if (16 * kb + 16 <= nv) {                                                   // wave-uniform fast path
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    r[j] = dyrow[(v0 + j) * 32];
                    if constexpr (GATE) r[j] = gtrow[(v0 + j) * 32] > 0.f ? r[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool live = v0 + j < nv;
                    float d = dyrow[live ? (v0 + j) * 32 : 0];
                    if constexpr (GATE) d = gtrow[live ? (v0 + j) * 32 : 0] > 0.f ? d : 0.f;
                    r[j] = live ? d : 0.f;
                }
            }
        };
        auto tr_addr = [&](int kb, int jj) -> int {                                     // block row li >> 2 of half h, columns 16 gcol + 4 (li & 3)
            int vi = 16 * kb + 8 * h + 4 * jj + (li >> 2);
            vi = vi < nv ? vi : nv - 1;
            const int w = (int)__umulhi((unsigned)vi, a.mT), t = vi - w * a.T;
            return (w * a.Tp + t) * VS;                                                  // (the lane's channel offset is part of tapoff[])
        };
        typedef const unsigned char* cptr;
        if constexpr (DYI) {
            // B operand of k-block kb: rows 16 kb + 8 h + 4 jj + (li >> 2) of the dY image, channels 16 gcol + 4 (li & 3) .. + 3 (the read transposes)
            auto load_b = [&](int kb, Frag (&f)[NPC]) {
                cptr p0 = dyimg + (16 * kb + 8 * h + (li >> 2)) * 128 + (16 * gcol + 4 * (li & 3)) * 2;
#pragma unroll
                for (int p = 0; p < NPC; ++p) {
                    f[p].hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + p * 64));
                    f[p].hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * 128 + p * 64));
                }
            };
            Frag bf[NPC], bfn[NPC];
            int va[2], van[2];
            load_b(ksel < nkb ? ksel : 0, bf);
            va[0] = tr_addr(ksel, 0); va[1] = tr_addr(ksel, 1);
#pragma unroll 1
            for (int kb = ksel; kb < nkb; kb += 2) {
                const int kn = kb + 2 < nkb ? kb + 2 : kb;
                load_b(kn, bfn);
                Frag af[2][NPC];
                auto load_a = [&](int j, Frag (&f)[NPC]) {
                    cptr p0 = lds_raw + tapoff[j] + va[0];
                    cptr p1 = lds_raw + tapoff[j] + va[1];
#pragma unroll
                    for (int p = 0; p < NPC; ++p) {
                        f[p].hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + p * CB));
                        f[p].hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p1 + p * CB));
                    }
                };
                load_a(0, af[0]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (j + 1 < NJ) load_a(j + 1, af[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tg + 4 * j < NT) acc[j] = mac<AR>(af[j & 1], bf, acc[j]);     // wave-uniform
                    if (j >= 4 && j < 6) {                                             // the next block's transposed-read addresses, in the shadow of these MFMAs
                        int kk = kn;
                        asm volatile("" : "+v"(kk));
                        int v = tr_addr(kk, j - 4);
                        asm volatile("" : "+v"(v));
                        van[j - 4] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int p = 0; p < NPC; ++p) bf[p] = bfn[p];
                va[0] = van[0]; va[1] = van[1];
            }
        } else {
            Frag bf[NPC], bfn[NPC];
            int va[2], van[2];
            float rawn[8], rawnn[8];
            {
                float raw0[8];
                load_dy(ksel, raw0);
                load_dy(ksel + 2 < nkb ? ksel + 2 : ksel, rawn);
                cut8<AR>(raw0, sd, bf);
                va[0] = tr_addr(ksel, 0); va[1] = tr_addr(ksel, 1);
                if (tg == 0 && ksel < nkb) {
    #pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += raw0[j];
                }
            }
    #pragma unroll 1
            for (int kb = ksel; kb < nkb; kb += 2) {
                const int kn = kb + 2 < nkb ? kb + 2 : kb, knn = kb + 4 < nkb ? kb + 4 : kb;
                load_dy(knn, rawnn);
                Frag af[2][NPC];
                auto load_a = [&](int j, Frag (&f)[NPC]) {
                    cptr p0 = lds_raw + tapoff[j] + va[0];
                    cptr p1 = lds_raw + tapoff[j] + va[1];
    #pragma unroll
                    for (int p = 0; p < NPC; ++p) {
                        f[p].hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + p * CB));
                        f[p].hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p1 + p * CB));
                    }
                };
                load_a(0, af[0]);
                    static_assert(AR::SCALED || NJ == 7, "the seven-step preparation pipeline below");
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (j + 1 < NJ) load_a(j + 1, af[(j + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tg + 4 * j < NT) acc[j] = mac<AR>(af[j & 1], bf, acc[j]);         // wave-uniform
                    // a seventh of the next block's preparation.  The empty volatile asm statements pin it between this tap's
                    // scheduling barriers (pure arithmetic would otherwise be sunk to the end of the loop body).
                    if (j < 4) {
                        float ra = rawn[2 * j], rb = rawn[2 * j + 1];
                        asm volatile("" : "+v"(ra), "+v"(rb));
                        unsigned q[NPC];
                        cut_pair<AR>(ra, rb, sd, q);
    #pragma unroll
                        for (int p = 0; p < NPC; ++p) {
                            asm volatile("" : "+v"(q[p]));
                            if (j == 0) bfn[p].u.x = q[p];
                            if (j == 1) bfn[p].u.y = q[p];
                            if (j == 2) bfn[p].u.z = q[p];
                            if (j == 3) bfn[p].u.w = q[p];
                        }
                    } else if (j < 6) {
                        int kk = kn;
                        asm volatile("" : "+v"(kk));
                        int v = tr_addr(kk, j - 4);
                        asm volatile("" : "+v"(v));
                        van[j - 4] = v;
                    } else if (tg == 0 && kb + 2 < nkb) {
    #pragma unroll
                        for (int i = 0; i < 8; ++i) bsum += rawn[i];
                    }
    #pragma unroll
                    for (int i = 0; i < (AR::SCALED ? 3 : 6); ++i) {                      // one MFMA, then two VALU in its shadow
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
    #pragma unroll
                for (int p = 0; p < NPC; ++p) bf[p] = bfn[p];
                va[0] = van[0]; va[1] = van[1];
    #pragma unroll
                for (int i = 0; i < 8; ++i) rawn[i] = rawnn[i];
            }
        }
        XS_ACC(3);
        __syncthreads();                                   // every wave is done with this tile's rows
        XS_ACC(4);
        if (consecutive) stage_store(ho + 3, nf, nmeta);     // replaces ring row ho
        else if (has_next) stage_three(nn, nho, nws0, nWts);
        if constexpr (DYI) { if (has_next) dy_store(nWts, ndy); }
        XS_ACC(5);
    }
    // slab of this workgroup: [27 * Cin][Cout] (+ bias sums).  The two k-block parities meet in LDS (the ring is dead now):
    // parity 1 leaves its accumulators there, parity 0 adds them and writes the slab once.
    float* pp = partial + (long)blockIdx.x * 27 * CIN * a.Cout;
    float* pb = partial_b + (long)blockIdx.x * a.Cout;
    float* xch = reinterpret_cast<float*>(lds_raw) + tg * (7 * 16 + 1) * 64;             // [tg][7 taps x 16 registers + bias][64 lanes]
    __syncthreads();
    if constexpr (DYI) {                                   // bias sums: 64 threads hold partial sums of each channel quad (fixed order)
        reinterpret_cast<float4*>(lds_raw)[tid] = bs4;
        __syncthreads();
        if (tid < a.Cout) {
            float b = 0.f;
            for (int m = 0; m < 64; ++m) b += reinterpret_cast<const float*>(lds_raw)[(8 * m + (tid >> 2)) * 4 + (tid & 3)];
            pb[tid] = b;
        }
        __syncthreads();
    }
    if (ksel == 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64 + lane] = acc[j][r];
        xch[7 * 16 * 64 + lane] = bsum;
    }
    __syncthreads();
    if (ksel == 0) {
        // The slab is written through a buffer descriptor: a GEMM row that is no (tap, input channel) pair gets an offset beyond the slab and the
        // store is dropped -- no branch per store.  Registers 4 q .. 4 q + 3 of a lane are four consecutive rows R0 .. R0 + 3 with R0 a multiple of 4,
        // and RT is one too: they belong to ONE tap, so the row -> (tap, channel) division is done once per four stores.
        unsigned plo = (unsigned)(unsigned long)pp, phi = (unsigned)((unsigned long)pp >> 32);
        plo = __builtin_amdgcn_readfirstlane(plo); phi = __builtin_amdgcn_readfirstlane(phi);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((unsigned long)phi << 32) | plo), 0, 27 * CIN * 32 * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int T = tg + 4 * j;
            if (T >= NT) continue;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[j][r] + xch[(j * 16 + r) * 64 + lane];
                if constexpr (AR::SCALED) v[r] = ldexpf(v[r], kun);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int R0 = 32 * T + 8 * q + 4 * h;                                 // = 32 T + rowmap(4 q, h): row of the GEMM -> (tap, input channel)
                const int tap = R0 / RT, ci0 = R0 - tap * RT;
                const int base = tap < 27 ? ((tap * CIN + ci0) * 32 + col) * 4 : (int)0x80000000;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int off = ci0 + i < CIN ? base + i * 128 : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[4 * q + i]), rs, off, 0, 0);
                }
            }
        }
        if (tg == 0 && !DYI) {
            float b = bsum + xch[7 * 16 * 64 + lane];
            b += __shfl_xor(b, 32, 64);
            if (h == 0) pb[col] = b;
        }
    }
    XS_ACC(6);
    XS_OUT;
}

static int x6_wgrad_split(const ConvGeom& g, int arith = 1)              // number of column ranges per row, 0 = unsupported
{
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.ph < 0 || g.ph > 1 || g.pw < 0 || g.pw > 1 || g.pt < 0 || g.pt > 1) return 0;
    if (g.reflect_hw && (g.ph != 1 || g.pw != 1 || g.Hi < 2 || g.Wi < 2)) return 0;
    if ((g.Cin != 25 && g.Cin != 32) || g.Cout != 32 || g.Ti < 2 || g.To < 2) return 0;
    if (g.Ho != g.Hi + 2 * g.ph - 2 || g.Wo != g.Wi + 2 * g.pw - 2 || g.To != g.Ti + 2 * g.pt - 2 || g.Ho < 1 || g.Wo < 1) return 0;
    const int np_ = arith == 2 ? 2 : 3;
    const int vs = (g.Cin == 25 ? 56 : 64) * np_, np = (g.Cin + 1) / 2, nst = g.Cin == 25 ? 6 : 7;
    for (int ns = 1; ns <= 4 && ns <= g.Wo; ++ns) {
        const int Wt = (g.Wo + ns - 1) / ns;
        size_t lds = (size_t)3 * (Wt + 2) * (g.Ti + 2 * g.pt) * vs + 16;
        const int dyrows = (Wt * g.To + 15) & ~15;
        if (arith == 2) { lds = ((lds + 15) & ~(size_t)15) + (size_t)dyrows * 128; if (dyrows * 8 > 512 * 4) continue; }   // H3: + the dY image
        if (lds <= 160 * 1024 && (Wt + 2) * g.Ti * np <= 512 * nst) return ns;
    }
    return 0;
}
bool x6_wgrad_supported(const ConvGeom& g) { return x6_wgrad_split(g) > 0; }     // (whatever X6 fits, H3 fits)

static int x6_wgrad_grid(const ConvGeom& g, int arith = 1)
{
    const int total = g.N * g.Ho * x6_wgrad_split(g, arith);
    return total < 256 ? total : 256;
}
// scratch for either arithmetic
size_t x6_wgrad_partial_floats(const ConvGeom& g)
{
    const int g1 = x6_wgrad_grid(g, 1), g2 = x6_wgrad_split(g, 2) ? x6_wgrad_grid(g, 2) : 0;
    return (size_t)(g1 > g2 ? g1 : g2) * ((size_t)27 * g.Cin * g.Cout + g.Cout);
}

int x6_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db, float* partial,
                  int arith, const Amax& am, hipStream_t s)
{
    if (!x6_wgrad_supported(g)) { set_error("x6_conv_wgrad: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    if (arith == 2 && (!am.x || !am.w)) { set_error("x6_conv_wgrad: H3 arithmetic needs the per-sample amax slots of x (am.x) and dY (am.w)", hipSuccess); return PROBAV_EINVAL; }
    // round 5: the residual blocks' and the reducers' layers as one-wave-per-SIMD kernel (kernels_wg4.hip); this kernel is the general form (any extent, pads, reflect, gate, 32 channels).
    // H3 on the general form would be H3 with PLAIN second pieces: one scale per operand tensor and a channel 2^-24 below its mates resolved to 7e-5 (the lifted pieces of
    // conv3_wgrad_w4_kernel need a second accumulator set, and this kernel's 112 accumulator registers live in a budget of 128).  A layer without an instance of that kernel
    // (depth 19, other extents, more than 256 samples) therefore runs the x6 arithmetic here -- bf16 pieces carry fp32's exponent, no operand is scaled, every slice is resolved;
    // the H3 instance stays reachable for A/B runs (PROBAV_GEN1 = wg | 1: every layer on this kernel, H3 as in rounds 2 - 4).
    if (arith == 2 && wg4_enabled()) {
        if (wg4_wgrad_supported(g, gate)) return wg4_conv_wgrad(g, x, dy, gate, dw, db, partial, am, s);
        arith = 1;
    }
    if (arith == 2 && x6_wgrad_split(g, 2) == 0) arith = 1;                  // (the H3 form also needs room for the dY image; the scale-free form serves the rest)
    WgArgs a;
    a.nsplit = x6_wgrad_split(g, arith); a.Wt = (g.Wo + a.nsplit - 1) / a.nsplit;
    a.N = g.N; a.H = g.Ho; a.W = g.Wo; a.T = g.To; a.Cout = g.Cout; a.Hi = g.Hi; a.Wi = g.Wi; a.Ti = g.Ti;
    a.ph = g.ph; a.pw = g.pw; a.pt = g.pt; a.reflect = g.reflect_hw;
    a.Wp = a.Wt + 2; a.Tp = g.Ti + 2 * g.pt;
    a.nv = a.Wt * g.To; a.total_tiles = g.N * g.Ho * a.nsplit;
    a.mT = (unsigned)((0x100000000ull + (unsigned)g.To - 1) / (unsigned)g.To);
    a.mTi = (unsigned)((0x100000000ull + (unsigned)g.Ti - 1) / (unsigned)g.Ti);
    const int grid = x6_wgrad_grid(g, arith);
    const long nw = (long)27 * g.Cin * g.Cout;
    float* partial_b = partial + (size_t)grid * nw;
    const int vs = (g.Cin == 25 ? 56 : 64) * (arith == 2 ? 2 : 3);
    size_t lds = (size_t)3 * a.Wp * a.Tp * vs + 16;
    if (arith == 2) lds = ((lds + 15) & ~(size_t)15) + (size_t)((a.nv + 15) & ~15) * 128;      // H3: + the dY image
    const size_t xch = (size_t)4 * (7 * 16 + 1) * 64 * sizeof(float);                 // exchange area of the epilogue
    if (lds < xch) lds = xch;
    a.tab = 0;
    {   // the staging table (kernel: stab): one column range per row, and room behind the images
        const size_t tb = (lds + 15) & ~(size_t)15, tsz = (size_t)(g.Cin == 25 ? 6 : 7) * 512 * 8;
        if (a.nsplit == 1 && tb + tsz <= 160 * 1024) { a.tab = (int)tb; lds = tb + tsz; }
    }
    static std::once_flag once;
    std::call_once(once, [] {
#define PROBAV_WGA(C, G, A) note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_x6_kernel<C, G, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        PROBAV_WGA(25, false, X6); PROBAV_WGA(25, true, X6); PROBAV_WGA(32, false, X6); PROBAV_WGA(32, true, X6);
        PROBAV_WGA(25, false, H3); PROBAV_WGA(25, true, H3); PROBAV_WGA(32, false, H3); PROBAV_WGA(32, true, H3);
#undef PROBAV_WGA
    });
#define PROBAV_WG6(C, G, A) hipLaunchKernelGGL((conv3_wgrad_x6_kernel<C, G, A>), dim3(grid), dim3(512), lds, s, a, x, dy, gate, partial, partial_b, am)
    if (arith == 2) {
        if (g.Cin == 25) { if (gate) PROBAV_WG6(25, true, H3); else PROBAV_WG6(25, false, H3); }
        else             { if (gate) PROBAV_WG6(32, true, H3); else PROBAV_WG6(32, false, H3); }
    } else {
        if (g.Cin == 25) { if (gate) PROBAV_WG6(25, true, X6); else PROBAV_WG6(25, false, X6); }
        else             { if (gate) PROBAV_WG6(32, true, X6); else PROBAV_WG6(32, false, X6); }
    }
#undef PROBAV_WG6
    int rc = check_launch("conv3_wgrad_x6");
    if (rc) return rc;
    return mfma_wgrad_reduce(partial, partial_b, dw, db, nw, g.Cout, grid, s);
}

}  // namespace probav
