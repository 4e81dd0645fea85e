// Fused expConv + ReLU + decConv forward (1x1x1, 32 -> 256 -> D <= 32), H3 arithmetic -- ONE WAVE PER SIMD, no LDS traffic in the tile loop (round 5).
// Reference semantics: models/modelsTF.py:179-183 (ResConv3D: expConv_i -> ReLU -> decConv_i).
//
// The arithmetic is pw_fwd_x6_kernel<H3>'s, product for product and in its order: H^T[hidden][voxel] = W1c X^T in two k-blocks of v_mfma_f32_32x32x16_f16
// (w1 x0 + w0 x1 + w0 x0 each), H' = max(fma(H, c, b), 0) at the sample's hidden scale, cut into two fp16 pieces, T^T[out][voxel] += W2c^T H' chunk by chunk -- the
// accumulator registers of the first product ARE the second one's B operand (its k-order is the accumulator's row order; PACK_H3_PW_W2 is packed to match).  That is
// also the order in which the reverse pass recomputes the hidden tile (pw_bwd_w4_kernel): forward and reverse pass decide every ReLU gate on the same bits.
// (pw_fwd_h3k_kernel, the forward kernel of rounds 3 and 4, sums the 32 input channels in one 16x16x32 instruction: gates of pre-activations at zero could differ.)
//
// What is different is where things live and who does what:
//   * the weight fragments sit in registers for the whole launch -- all of W1 and the first pieces of W2 in a[0:191] (an MFMA reads its A operand from there), the second
//     pieces of W2 in LDS (2 x ds_read_b128 per chunk); each wave requests a quarter of the 64 KB, they meet in LDS once (the prologue of kernels_cw4.hip);
//   * X comes from memory straight into registers (the lane's voxel, eight consecutive input channels per lane half and k-block) and is cut there, a tile ahead;
//   * four independent waves per workgroup, each with a contiguous run of tiles, no barrier in the loop;
//   * one hand-pipelined instruction stream per wave: iteration (tile, chunk c) issues the second product of chunk c - 1 and the first product of chunk c + 1 -- twelve MFMAs --
//     and in their gaps the vector work of chunk c (bias / ReLU, the cut: 56 instructions), plus its share of the tile's other work (the epilogue of the previous tile, the
//     cut of the next tile's X, requests).  The stream is bound by the vector instructions it issues (~6 per MFMA, v_cvt_pk / v_fma_mix at 8-9 cycles: docs/notebook_r1-r5.md 4.0).
#include "kernels_x6.h"
#include "x6_device.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace probav {
#ifdef PF4_DIAG                 // tools/pf4bench.hip may include this file as probav::diag
namespace diag {
#endif

namespace {
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
constexpr unsigned PF4_OOB = 0x40000000u;     // a buffer offset beyond every num_records: stores are dropped
constexpr int PF4_TAB = 64 * 1024;            // byte offset of the tables behind the exchange area / the second pieces of W2
}
#define PF4_SBAR() __builtin_amdgcn_sched_barrier(0)
#define PF4_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

template <bool DUMP, int DT>      // DUMP (tests only): the post-ReLU hidden tile also goes to hdump [nvox][256] (true values, as pw_fwd_x6_kernel's); DT: the layer's D where it is 25, 0 = any D <= 32
__global__ __launch_bounds__(256, 1) void pw_fwd_w4_kernel(const float* __restrict__ x, const uint4* __restrict__ w1frag, const uint4* __restrict__ w2frag,
                                                           const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ dec,
                                                           long nvox, int vps, int Drt, PwAmax am, float* __restrict__ hdump)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int D = DT ? DT : Drt;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const sB1 = reinterpret_cast<float*>(lds + PF4_TAB);                  // [256] expand biases
    float* const sB2 = sB1 + 256;                                                // [32] decay biases
    int* const sE2 = reinterpret_cast<int*>(sB2 + 32);                           // [32] H3 exponents of the decay filter's output columns
    float* const sBw = reinterpret_cast<float*>(sE2 + 32) + wave * 864;          // this wave's: [3][256] expand biases at a sample's hidden scale, [3][32] output multipliers 2^-(e2 + eh)
    float* const sMw = sBw + 768;                                                // (three copies: the previous tile's epilogue, the current tile and the next one may be three samples)

    // ---- the wave's run of tiles (tiles never straddle samples; the last tile of a sample may be partial) ----
    const int tps = (vps + 31) >> 5;
    const long ntiles = (nvox / vps) * tps;
    const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    const long tb = ntiles * gw / nw, te = ntiles * (gw + 1) / nw;
    // raw X of a tile: the lane's voxel, input channels 16 kb + 8 half .. + 7 (two 16-byte loads per k-block)
    // (through a buffer descriptor: 32-bit offsets, no 64-bit address arithmetic per request; sample and tile-in-sample of the tiles ahead are kept incrementally)
    u32x4p nx[2][2];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)(nvox * 128), 0x00020000);
    auto load_x = [&](int nn, int jj) {
        const int vl = 32 * jj + col;
        const int v = nn * vps + (vl < vps ? vl : vps - 1);
        const int off = v * 128 + 32 * half;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            nx[kb][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 64 * kb, 0, 0);
            nx[kb][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 64 * kb + 16, 0, 0);
        }
    };
    int n = (int)(tb / tps), j = (int)(tb - (long)n * tps);                      // sample / tile-in-sample of the tile being multiplied
    auto step = [&](int& nn, int& jj) { if (++jj == tps) { jj = 0; ++nn; } };
    int n1 = n, j1 = j; step(n1, j1);                                            // of the next tile, and of the one after it
    int n2 = n1, j2 = j1; step(n2, j2);
    if (tb < te) load_x(n, j);
    // ---- the weight fragments: 64 of 1 KB (W1 [8 chunks][2 kb][2 pieces], then W2 likewise); a quarter per wave, exchanged through LDS ----
    {
        const u32x4p* w14 = reinterpret_cast<const u32x4p*>(w1frag);
        const u32x4p* w24 = reinterpret_cast<const u32x4p*>(w2frag);
        u32x4p xq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int f = 4 * i + wave; xq[i] = f < 32 ? w14[f * 64 + lane] : w24[(f - 32) * 64 + lane]; }
        const float bq = b1[tid];
        float b2q = 0.f; int e2q = 0;
        if (tid < 32) { b2q = tid < D ? b2[tid] : 0.f; e2q = tid < D ? h3_exp_w(am.w2c[tid]) : 0; }
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4p*>(lds + (4 * i + wave) * 1024 + lane * 16) = xq[i];
        sB1[tid] = bq;
        if (tid < 32) { sB2[tid] = b2q; sE2[tid] = e2q; }
    }
    __syncthreads();
    f16x8 w1a[8][2][2], w2a[8][2];                                               // all of W1, the first pieces of W2: 192 registers of the accumulator half
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int p = 0; p < 2; ++p) { Frag f; f.u = *reinterpret_cast<const uint4*>(lds + (((c * 2 + kb) * 2 + p) * 64 + lane) * 16); w1a[c][kb][p] = f.h; }
            Frag f; f.u = *reinterpret_cast<const uint4*>(lds + ((32 + (c * 2 + kb) * 2) * 64 + lane) * 16); w2a[c][kb] = f.h;
        }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { asm volatile("" : "+a"(w1a[c][kb][0])); asm volatile("" : "+a"(w1a[c][kb][1])); asm volatile("" : "+a"(w2a[c][kb])); }
    const unsigned char* const w2l = lds + ((32 + 1) * 64 + lane) * 16;           // the lane's 16 bytes of W2's second piece (chunk 0, kb 0); + (c * 2 + kb) * 2048

    // ---- per-sample scales: the hidden tensor exists only in registers, so its scale comes from a bound: |h| <= 32 amax(x) amax(w1) + amax(b1) ----
    const unsigned aw1 = *am.w1, ab1 = *am.b1;
    const int ew1 = h3_exp_w(aw1);
    auto clampexp = [](int k) { return k < -126 ? -126 : k; };
    int flip = 0;                                                                // which copy of the wave's tables the current tile reads
    float sx = 1.f, ch = 1.f;                                                    // of the tile being multiplied: X scale (used when its rows were cut), accumulator -> hidden scale
    float dsc = 1.f, dscn = 1.f;                                                 // (DUMP) 2^-eh of the current / the next tile's sample
    auto sample_tables = [&](int n, int fl, float& sx_, float& ch_, float& dsc_) {      // scales of sample n; its bias / multiplier tables into copy fl
        const unsigned ax = am.x[n];
        const int ex = h3_exp(ax);
        const int eh = h3_exp(32.f * __uint_as_float(ax) * __uint_as_float(aw1) + __uint_as_float(ab1));
        sx_ = pow2i(ex);
        ch_ = pow2i(clampexp(eh - ex - ew1));
        dsc_ = pow2i(clampexp(-eh));
        const float sb = pow2i(eh);
#pragma unroll
        for (int q = 0; q < 4; ++q) sBw[fl * 256 + lane + 64 * q] = sB1[lane + 64 * q] * sb;
        if (lane < 32) sMw[fl * 32 + lane] = pow2i(clampexp(-(sE2[lane] + eh)));
    };

    // ---- the pipeline's registers ----
    f32x16 H[2], T;
    float tp[16];                                                                // the previous tile's sums while its epilogue runs
    Frag xb[2][2], xbn[2][2], hb[2][2][2];                                       // X pieces of the tile / the next tile [kb][piece]; H' pieces [chunk parity][kb][piece]
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) { zero[i] = 0.f; T[i] = 0.f; tp[i] = 0.f; H[0][i] = 0.f; H[1][i] = 0.f; }
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int c_ = 0; c_ < 2; ++c_) hb[a_][b_][c_].u = make_uint4(0u, 0u, 0u, 0u);
    float omax = 0.f;
    float omask[16];                                                             // 1 for the lane's output channels that exist (8 g + 4 half + i < D), 0 beyond
#pragma unroll
    for (int i = 0; i < 16; ++i) omask[i] = 8 * (i >> 2) + 4 * half + (i & 3) < D ? 1.f : 0.f;
    if (tb >= te) return;
    sample_tables(n, 0, sx, ch, dsc);
    // the first tile's X pieces; the second tile's rows are requested
    auto cut_x = [&](int kb, int p, float s, Frag (&dst)[2][2]) {                // pair p of k-block kb: the scaled pair, its first pieces, its second pieces
        const u32x4p q = nx[kb][p >> 1];
        unsigned qq[2];
        cut_pair<H3>(__uint_as_float(q[2 * (p & 1)]), __uint_as_float(q[2 * (p & 1) + 1]), s, qq);
        if (p == 0) { dst[kb][0].u.x = qq[0]; dst[kb][1].u.x = qq[1]; }
        if (p == 1) { dst[kb][0].u.y = qq[0]; dst[kb][1].u.y = qq[1]; }
        if (p == 2) { dst[kb][0].u.z = qq[0]; dst[kb][1].u.z = qq[1]; }
        if (p == 3) { dst[kb][0].u.w = qq[0]; dst[kb][1].u.w = qq[1]; }
    };
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < 4; ++p) cut_x(kb, p, sx, xb);
    if (tb + 1 < te) load_x(n1, j1);
    // the first tile's first product, chunk 0; its biases
    H[0] = PF4_MFMA(w1a[0][0][1], xb[0][0].h, zero); H[0] = PF4_MFMA(w1a[0][0][0], xb[0][1].h, H[0]); H[0] = PF4_MFMA(w1a[0][0][0], xb[0][0].h, H[0]);
    H[0] = PF4_MFMA(w1a[0][1][1], xb[1][0].h, H[0]); H[0] = PF4_MFMA(w1a[0][1][0], xb[1][1].h, H[0]); H[0] = PF4_MFMA(w1a[0][1][0], xb[1][0].h, H[0]);
    float4 bq[4];                                                                // biases of the chunk whose vector work comes next: hidden 32 c + 8 g + 4 half + (0..3)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(sBw + 8 * g + 4 * half);
    // output through a buffer descriptor: an invalid voxel's offset lies beyond num_records
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(dec, 0, (unsigned)(nvox * D) * 4u, 0x00020000);
    unsigned vo = PF4_OOB, vop = PF4_OOB;                                        // byte offset of the lane's voxel in dec: the tile being multiplied / the previous one
    int hv = 0;                                                                  // (DUMP) the lane's voxel
    {
        const int vl = 32 * j + col;
        vo = vl < vps ? (unsigned)((n * vps + vl) * D) * 4u : PF4_OOB;
        hv = n * vps + (vl < vps ? vl : vps - 1);
    }
    int np = n, flp = 0;                                                         // sample and table copy of the previous tile (its epilogue runs a tile later)
    float sxn = sx, chn = ch;                                                    // scales of the NEXT tile's sample
    int nn = n, fln = 0;

    // One iteration = chunk C of the current tile: the second product of chunk C - 1 (of the previous tile for C = 0), the first product of chunk C + 1 (of the next tile
    // for C = 7), and in their gaps chunk C's vector work -- and the tile's other work, dealt out over the eight iterations.
    auto iter = [&](auto c_tag, long tile) __attribute__((always_inline)) {
        constexpr int C = decltype(c_tag)::value, CP = (C + 7) & 7, CN = (C + 1) & 7, P = C & 1;
        Frag w2s[2];                                                              // second pieces of W2, chunk CP (requested behind the first MFMA, used from the seventh on)
        float hs[16];
        unsigned q0[4][2], q1[4][2];                                              // first / second pieces of the four groups (two dwords each)
#pragma unroll
        for (int G = 0; G < 12; ++G) {
            // ---- the MFMA: first the next chunk's first product (its result is read an iteration later), then the previous chunk's second product ----
            if (G < 6) {                                                          // H of chunk CN = W1c X^T: per k-block w1 x0 + w0 x1 + w0 x0
                const int kb = G / 3, r = G % 3;
                const f16x8 wa = r == 0 ? w1a[CN][kb][1] : w1a[CN][kb][0];
                const f16x8 xv = C == 7 ? (r == 1 ? xbn[kb][1].h : xbn[kb][0].h) : (r == 1 ? xb[kb][1].h : xb[kb][0].h);
                if (G == 0) H[P ^ 1] = PF4_MFMA(wa, xv, zero); else H[P ^ 1] = PF4_MFMA(wa, xv, H[P ^ 1]);
            } else {                                                              // T += W2c^T H' of chunk CP
                const int kb = (G - 6) / 3, r = (G - 6) % 3;
                const f16x8 wa = r == 0 ? w2s[kb].h : w2a[CP][kb];
                const f16x8 hbv = r == 1 ? hb[P ^ 1][kb][1].h : hb[P ^ 1][kb][0].h;
                if (CP == 0 && G == 6) T = PF4_MFMA(wa, hbv, zero); else T = PF4_MFMA(wa, hbv, T);
            }
            PF4_SBAR();
            // ---- chunk C's vector work: H' = max(fma(H, ch, b), 0), cut into pieces; group g = registers 4 g .. 4 g + 3 = hidden 32 C + 8 g + 4 half + (0..3) ----
            if (G < 8) {                                                          // two values per gap
                const float4 b4 = bq[G >> 1];
                // (a packed fma -- v_pk_fma_f32 through asm -- was tried: the compiler copies its operands into aligned pairs and pads hazards, 74 against 58 us)
                hs[2 * G] = fmaxf(fmaf(H[P][2 * G], ch, (G & 1) ? b4.z : b4.x), 0.f);
                hs[2 * G + 1] = fmaxf(fmaf(H[P][2 * G + 1], ch, (G & 1) ? b4.w : b4.y), 0.f);
            }
            // first pieces of group g behind its four values, the second pieces one and two gaps later
            if (G == 2 || G == 5 || G == 8 || G == 9) {
                const int g = G == 2 ? 0 : G == 5 ? 1 : G == 8 ? 2 : 3;
                q0[g][0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){hs[4 * g], hs[4 * g + 1]}, f16x2));
                q0[g][1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){hs[4 * g + 2], hs[4 * g + 3]}, f16x2));
            }
            if (G == 3 || G == 6 || G == 9 || G == 10) {
                const int g = G == 3 ? 0 : G == 6 ? 1 : G == 9 ? 2 : 3;
                asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
                    : "=&v"(q1[g][0]), "=&v"(q1[g][1]) : "v"(q0[g][0]), "v"(q0[g][1]), "v"(hs[4 * g]), "v"(hs[4 * g + 2]));
            }
            if (G == 4 || G == 7 || G == 10 || G == 11) {
                const int g = G == 4 ? 0 : G == 7 ? 1 : G == 10 ? 2 : 3;
                asm("v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "+v"(q1[g][0]), "+v"(q1[g][1]) : "v"(q0[g][0]), "v"(q0[g][1]), "v"(hs[4 * g + 1]), "v"(hs[4 * g + 3]));
                // the group's place in the fragments: dwords 2 (g & 1), + 1 of k-block g >> 1
                if (g & 1) { hb[P][g >> 1][0].u.z = q0[g][0]; hb[P][g >> 1][0].u.w = q0[g][1]; hb[P][g >> 1][1].u.z = q1[g][0]; hb[P][g >> 1][1].u.w = q1[g][1]; }
                else { hb[P][g >> 1][0].u.x = q0[g][0]; hb[P][g >> 1][0].u.y = q0[g][1]; hb[P][g >> 1][1].u.x = q1[g][0]; hb[P][g >> 1][1].u.y = q1[g][1]; }
                if constexpr (DUMP) {
                    if (vo != PF4_OOB) {
                        float* hp_ = hdump + (long)hv * 256 + 32 * C + 8 * g + 4 * half;
                        hp_[0] = hs[4 * g] * dsc; hp_[1] = hs[4 * g + 1] * dsc; hp_[2] = hs[4 * g + 2] * dsc; hp_[3] = hs[4 * g + 3] * dsc;
                    }
                }
            }
            // ---- requests: the second pieces of W2 of chunk CP (used six gaps on), the biases of chunk C + 1 (of the next tile's sample for C = 7) behind this chunk's last use of them ----
            if (G == 0) { w2s[0].u = *reinterpret_cast<const uint4*>(w2l + (CP * 2) * 2048); w2s[1].u = *reinterpret_cast<const uint4*>(w2l + (CP * 2 + 1) * 2048); }
            if (G == 8) {
                const float* bsrc = sBw + (C == 7 ? fln : flip) * 256 + 32 * CN + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(bsrc + 8 * g);
            }
            // ---- the tile's other work ----
            // C = 1: the previous tile's sums were complete behind the last MFMA of C = 0: copy them (its epilogue runs in C = 2 .. 5; T collects this tile's from this iteration's seventh MFMA on)
            if (C == 1 && G < 4) {
#pragma unroll
                for (int i = 4 * G; i < 4 * G + 4; ++i) tp[i] = T[i];
            }
            // C = 2 .. 5: the previous tile's epilogue, group g = C - 2: four values (one multiply-add each: the multiplier 2^-(e2 + eh) and the bias come from the wave's
            // table), their largest magnitude, a 16-byte store
            if (C >= 2 && C <= 5 && (G == 0 || G == 1 || G == 11)) {
                const int g = C - 2;
                if (G == 0) {
                    const float4 mm = *reinterpret_cast<const float4*>(sMw + flp * 32 + 8 * g + 4 * half), b4 = *reinterpret_cast<const float4*>(sB2 + 8 * g + 4 * half);
                    tp[4 * g] = fmaf(tp[4 * g], mm.x, b4.x); tp[4 * g + 1] = fmaf(tp[4 * g + 1], mm.y, b4.y);
                    tp[4 * g + 2] = fmaf(tp[4 * g + 2], mm.z, b4.z); tp[4 * g + 3] = fmaf(tp[4 * g + 3], mm.w, b4.w);
                }
                if (G == 1) {                                                     // (rows of the filter beyond D are whatever the packed fragments hold: not part of the tensor)
                    const bool whole = DT ? 8 * g + 8 <= DT : false;
                    const float t0 = whole ? tp[4 * g] : tp[4 * g] * omask[4 * g], t1 = whole ? tp[4 * g + 1] : tp[4 * g + 1] * omask[4 * g + 1];
                    const float t2 = whole ? tp[4 * g + 2] : tp[4 * g + 2] * omask[4 * g + 2], t3 = whole ? tp[4 * g + 3] : tp[4 * g + 3] * omask[4 * g + 3];
                    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(omax) : "v"(t0), "v"(t1));
                    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(omax) : "v"(t2), "v"(t3));
                }
                if (G == 11) {
                    const int c0 = 8 * g + 4 * half;
                    const u32x4p o = {__float_as_uint(tp[4 * g]), __float_as_uint(tp[4 * g + 1]), __float_as_uint(tp[4 * g + 2]), __float_as_uint(tp[4 * g + 3])};
                    if (c0 + 4 <= D) __builtin_amdgcn_raw_buffer_store_b128(o, yrs, vop + 4u * c0, 0, 0);
                    else {
                        if (c0 < D) __builtin_amdgcn_raw_buffer_store_b32(o[0], yrs, vop + 4u * c0, 0, 0);
                        if (c0 + 1 < D) __builtin_amdgcn_raw_buffer_store_b32(o[1], yrs, vop + 4u * c0 + 4u, 0, 0);
                        if (c0 + 2 < D) __builtin_amdgcn_raw_buffer_store_b32(o[2], yrs, vop + 4u * c0 + 8u, 0, 0);
                    }
                }
            }
            // C = 6: the previous tile is finished: its sample's largest magnitude leaves when the run has left the sample (or this is the run's first tile: nothing was finished)
            if (C == 6 && G == 0) {
                if (tile == tb) omax = 0.f;
                else if (np != n) { if (am.y) amax_commit(omax, am.y + np); omax = 0.f; }
            }
            // C = 0 .. 3: the cut of the NEXT tile's X (requested a tile ago), one pair per gap 10 / 11
            if (C < 4 && (G == 10 || G == 11)) {
                const int pi = 2 * C + (G - 10);                                  // 0 .. 7: k-block pi >> 2, pair pi & 3
                cut_x(pi >> 2, pi & 3, sxn, xbn);
            }
            // C = 4: the rows of the tile after the next one are requested (the registers are free: the cut above has read them)
            if (C == 4 && G == 10) { if (tile + 2 < te) load_x(n2, j2); }
            // C = 6, behind the last use of xb (the first product of chunk 7 was issued in this iteration's first half): xb = the next tile's pieces
            if (C == 6 && G == 7) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) xb[kb][p].u = xbn[kb][p].u;
            }
            PF4_SBAR();
        }
    };
#pragma unroll 1
    for (long tile = tb; tile < te; ++tile) {
        // the next tile's sample: its scales (for the cut of its X in C = 0 .. 3) and, if it is another sample, its tables in the other copy
        {
            nn = tile + 1 < te ? n1 : n;
            if (nn != n) { fln = flip == 2 ? 0 : flip + 1; sample_tables(nn, fln, sxn, chn, dscn); } else { fln = flip; sxn = sx; chn = ch; dscn = dsc; }
        }
        iter(std::integral_constant<int, 0>(), tile); iter(std::integral_constant<int, 1>(), tile); iter(std::integral_constant<int, 2>(), tile); iter(std::integral_constant<int, 3>(), tile);
        iter(std::integral_constant<int, 4>(), tile); iter(std::integral_constant<int, 5>(), tile); iter(std::integral_constant<int, 6>(), tile); iter(std::integral_constant<int, 7>(), tile);
        // the tile becomes the previous one
        vop = vo; np = n; flp = flip;
        n = nn; flip = fln; sx = sxn; ch = chn; dsc = dscn;
        {
            const int vl = 32 * j1 + col;
            vo = (tile + 1 < te && vl < vps) ? (unsigned)((n1 * vps + vl) * D) * 4u : PF4_OOB;
            hv = n1 * vps + (vl < vps ? vl : vps - 1);
        }
        j = j1; n1 = n2; j1 = j2; step(n2, j2);
    }
    // ---- drain: the last tile's second product of chunk 7, its epilogue ----
    {
        Frag w2s[2];
        w2s[0].u = *reinterpret_cast<const uint4*>(w2l + 14 * 2048); w2s[1].u = *reinterpret_cast<const uint4*>(w2l + 15 * 2048);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            T = PF4_MFMA(w2s[kb].h, hb[1][kb][0].h, T); T = PF4_MFMA(w2a[7][kb], hb[1][kb][1].h, T); T = PF4_MFMA(w2a[7][kb], hb[1][kb][0].h, T);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 mm = *reinterpret_cast<const float4*>(sMw + flp * 32 + 8 * g + 4 * half), b4 = *reinterpret_cast<const float4*>(sB2 + 8 * g + 4 * half);
            const float t0 = fmaf(T[4 * g], mm.x, b4.x), t1 = fmaf(T[4 * g + 1], mm.y, b4.y), t2 = fmaf(T[4 * g + 2], mm.z, b4.z), t3 = fmaf(T[4 * g + 3], mm.w, b4.w);
            const int c0 = 8 * g + 4 * half;
            const u32x4p o = {__float_as_uint(t0), __float_as_uint(t1), __float_as_uint(t2), __float_as_uint(t3)};
            if (vop != PF4_OOB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (c0 + i < D) omax = fmaxf(omax, fabsf(__uint_as_float(o[i])));
            }
            if (c0 + 4 <= D) __builtin_amdgcn_raw_buffer_store_b128(o, yrs, vop + 4u * c0, 0, 0);
            else {
                if (c0 < D) __builtin_amdgcn_raw_buffer_store_b32(o[0], yrs, vop + 4u * c0, 0, 0);
                if (c0 + 1 < D) __builtin_amdgcn_raw_buffer_store_b32(o[1], yrs, vop + 4u * c0 + 4u, 0, 0);
                if (c0 + 2 < D) __builtin_amdgcn_raw_buffer_store_b32(o[2], yrs, vop + 4u * c0 + 8u, 0, 0);
            }
        }
        if (am.y) amax_commit(omax, am.y + np);
    }
}

// ---- host side ----
#ifndef PF4_DIAG
static int g_pf4_enabled = -1;
bool pf4_enabled()
{
    if (g_pf4_enabled < 0) { const char* e = getenv("PROBAV_GEN1"); g_pf4_enabled = !(e && (e[0] == '1' || (e[0] == 'p' && e[1] == 'w' && e[2] != 'b'))); }      // PROBAV_GEN1 = 1 | pw | pwf | pwb | conv
    return g_pf4_enabled != 0;
}
void pf4_set_enabled(int on) { g_pf4_enabled = on ? 1 : 0; }
#endif

bool pf4_forward_supported(long nvox, long vps, int D)
{
    return D >= 1 && D <= 32 && vps >= 1 && nvox % vps == 0 && nvox * (long)D * 4 < 0x40000000L && nvox * 128L < 0xffffffffL;      // (32-bit buffer offsets)
}

int pf4_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec, long nvox, long vps, int D,
                const PwAmax& am, hipStream_t s, float* hdump)
{
    if (!pf4_forward_supported(nvox, vps, D)) { set_error("pf4_forward: unsupported shape", hipSuccess); return PROBAV_EINVAL; }
    if (!am.x || !am.w1 || !am.w2c || !am.b1) { set_error("pf4_forward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)pw_fwd_w4_kernel<false, 25>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)pw_fwd_w4_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)pw_fwd_w4_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    const size_t lds = (size_t)PF4_TAB + (256 + 32 + 32 + 4 * 864) * sizeof(float);
    if (hdump) hipLaunchKernelGGL((pw_fwd_w4_kernel<true, 0>), dim3(256), dim3(256), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag, b1, b2, dec, nvox, (int)vps, D, am, hdump);
    else if (D == 25) hipLaunchKernelGGL((pw_fwd_w4_kernel<false, 25>), dim3(256), dim3(256), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag, b1, b2, dec, nvox, (int)vps, D, am, nullptr);
    else hipLaunchKernelGGL((pw_fwd_w4_kernel<false, 0>), dim3(256), dim3(256), lds, s, x, (const uint4*)w1frag, (const uint4*)w2frag, b1, b2, dec, nvox, (int)vps, D, am, nullptr);
    return check_launch("pw_fwd_w4");
}

#ifdef PF4_DIAG
}  // namespace diag
#endif
}  // namespace probav
