// Fused backward of expConv + ReLU + decConv (1x1x1), H3 arithmetic -- ONE WAVE PER SIMD, the whole 512-register file (round 5).
// Reference semantics: tape.gradient through models/modelsTF.py:179-183 (ResConv3D: expConv_i -> ReLU -> decConv_i).
//
// What round 4 measured on pw_bwd_h3t_kernel (docs/notebook_r1-r5.md 4.0): two waves per SIMD, 228 VGPRs each, every wave one 32-channel hidden chunk of a tile
// shared by the eight waves of the workgroup; matrix pipe 41 % busy, the rest dependent chains through LDS hand-offs (both H' and dH' cross the LDS
// for their transposes, the dX partials of the eight chunks meet there, the two halves of the workgroup alternate phases behind s_barrier).
// This kernel removes the hand-offs instead of tuning them:
//   * A WAVE OWNS A TILE.  Four waves per workgroup (one per SIMD), each with its own run of 32-voxel tiles of ONE sample and all eight hidden
//     chunks of it: the dX contributions of the chunks add up in the MFMA accumulator (no partials in LDS), a wave stages its own rows (no
//     staging hand-off), and the tile loop holds no s_barrier at all.
//   * THE HIDDEN CHANNEL SITS ON THE LANE.  (a) and (b) are issued with the data as the A operand: H^T[voxel][hidden] = X W1c, dH^T = dT W2c^T.
//     Bias, ReLU and gate are elementwise; the cut registers of H'^T and dH'^T ARE the B operands of (e) dW2c^T += dT^T H' and (d) dW1c += X^T dH'
//     (both contract over the voxel = the accumulator's row index: cdna_hip_programming.md section 3, 'An accumulator tile as the next MFMA's
//     operand'), and the bias is one value per lane.  Only dH' crosses the LDS, once, for (c) dX^T += W1c dH'^T.
//   * 256 ACCUMULATOR REGISTERS: dW1c and dW2c^T of all eight chunks stay in a[0:255] for the whole run (asm MFMAs with "+a" operands); the
//     tile-local accumulators (H, dH, dX) are VGPR-form builtins (this unit is compiled with -mllvm -amdgpu-mfma-vgpr-form: build flags
//     in __graft_entry__.py), so the vector instructions read them without v_accvgpr moves.
//   * ONE INSTRUCTION STREAM, software-pipelined by hand over the chunks: chunk c's vector work (bias/ReLU, gate, the two cuts) runs in the
//     gaps of (b) of chunk c, (e), (d), (c) of chunk c-1 and (a) of chunk c+1.  One set of H / dH registers: (a) of the next chunk is issued
//     when the cut of H' is done, (b) when the cut of dH' is.
// Weights: the three fragment sets (96 KB) live in LDS for the whole launch, read per chunk.  Per wave: X and dT piece images of the current
// tile (row reads for (a), (b); transposed reads for (d), (e)), one dH' image, the expand biases at the sample's hidden scale.
// Products and their order inside a tile are those of pw_bwd_h3t_kernel (w1 x0 + w0 x1 + w0 x0 per k-block); what differs is the summation
// order ACROSS tiles (a wave's run instead of a workgroup's) and across chunks for dX (one accumulator chain instead of eight partials).
#include "kernels_x6.h"
#include "x6_device.h"
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace probav {
#ifdef PW4_DIAG                 // tools/pw4bench.hip includes this file as probav::diag (stamped / ablated builds beside the product's copy in the library)
namespace diag {
#endif
#ifdef PW4_STAMP                // diagnostic build only: cycles per phase, summed over a wave's run -- [wave][slot]: 0 (b), 1 (e), 2 (d), 3 (c), 4 (a) of the chunk iterations, 5 / 6 / 7 thirds of the boundary, 8 whole run, 9 100-MHz ticks of the run
__device__ unsigned long long g_pw4_stamps[1024 * 16];
#define PW4_ST(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define PW4_ST(k) do { } while (0)
#endif

// timing-only ablations of the diagnostic build (tools/pw4diag.hip; every one of them computes wrong results): PW4_ABL_NOMFMA no matrix instruction,
// PW4_ABL_NOVALU no bias / ReLU / gate / cut, PW4_ABL_NOLDS no LDS read inside the chunk iterations
#ifdef PW4_ABL_NOMFMA
#define PW4_MFMA_V(a, b, c) (c)
#else
#define PW4_MFMA_V(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a).h, (b).h, (c), 0, 0, 0)
#endif
// accumulator in the AGPR half of the file.  The A / B operands are never written by the instruction in front of the statement: they come out of
// LDS reads (the compiler's s_waitcnt stands in front) or were cut one chunk earlier.
// (hipcc pads nothing inside an asm statement: a VALU write of an operand needs two wait states in front of the MFMA that reads it.  In the tile loop the operands
// are LDS reads or cuts of the chunk before; tools/pw4_audit.py -- run by tests/test_pw4_audit.py on every build -- holds the emitted code to it.  PW4_MFMA_AS carries
// the wait states itself: the boundary in FRONT of a run takes zero fragments that the compiler materialises right where they are used.)
#ifdef PW4_ABL_NOMFMA
#define PW4_MFMA_A(acc, a, b) asm volatile("" : "+a"(acc) : "v"((a).h), "v"((b).h))
#define PW4_MFMA_AS(acc, a, b) asm volatile("" : "+a"(acc) : "v"((a).h), "v"((b).h))
#else
#define PW4_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"((a).h), "v"((b).h))
#define PW4_MFMA_AS(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"((a).h), "v"((b).h))
#endif
#define PW4_MFMA_B(acc, a, b) do { if constexpr (FIRST) PW4_MFMA_AS(acc, a, b); else PW4_MFMA_A(acc, a, b); } while (0)      // inside the boundary: FIRST = the one in front of a run

constexpr int PW4_WSET = 8 * 2 * 2 * 64 * 16;                 // bytes of one fragment set [8 chunks][2 k-blocks][2 pieces][64 lanes][16 B]
constexpr int PW4_XI = 0, PW4_DI = 2 * PB_IMG, PW4_GI = 4 * PB_IMG, PW4_SB = PW4_GI + 2 * PS_IMG, PW4_EX = PW4_SB + 1024, PW4_WAVE = PW4_EX + 128;
constexpr int PW4_LDS = 3 * PW4_WSET + 4 * PW4_WAVE;

__global__ __launch_bounds__(256, 1) void pw_bwd_w4_kernel(
    const float* __restrict__ x, const float* __restrict__ dT, const float* __restrict__ dOut,
    const uint4* __restrict__ w1f, const uint4* __restrict__ w2kf, const uint4* __restrict__ w1cf,
    const float* __restrict__ b1, float* __restrict__ dX, float* __restrict__ slabs, int nsamp, int vps, int D, int wps, PwAmax am)
{
#ifdef PW4_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const PW = lds + 3 * PW4_WSET + wave * PW4_WAVE;
    unsigned char* const XI = PW + PW4_XI;                         // [2 pieces][32 voxels][80 B]
    unsigned char* const DI = PW + PW4_DI;                         // same for dT (channels D..31 stay zero)
    unsigned char* const GI = PW + PW4_GI;                         // [2 pieces][32 hidden][64 B] dH' of one chunk (swizzled)
    float* const SB = reinterpret_cast<float*>(PW + PW4_SB);      // 256 expand biases at the sample's hidden-tile scale
    int* const EX = reinterpret_cast<int*>(PW + PW4_EX);           // 32: exponent that brings (c)'s accumulator row cin back to true scale

    // ---- this wave's run: tiles [tb, te) of sample n ----
    const int gw = blockIdx.x * 4 + wave;
    const int n = gw / wps, jw = gw - n * wps;
    const bool active = n < nsamp;
    const int tps = (vps + 31) >> 5;
    // The sample's tiles over its wps waves: tps / wps each, and the remainder one apiece to the waves of EVEN workgroups first.  Workgroup b runs on XCD b % 8, the odd
    // XCDs hold ~2.5 % less clock than the even ones in every kernel looked at (docs/notebook_r1-r5.md 4.0, 'The XCDs do not run at one speed'), and the launch ends with its
    // slowest wave: 137 tiles over 8 waves leave one wave with 18, which the plain split tps * jw / wps puts into an odd workgroup.  (Speed only: any split is correct.)
    int tb = 0, te = 0;
    if (active) {
        const int q = tps / wps, r = tps - q * wps;
        auto extra_before = [&](int j) {                            // how many of the waves 0 .. j-1 carry an extra tile
            if (wps & 7) return j < r ? j : r;                        // (waves per sample not a multiple of 8: workgroups are not aligned with samples -- plain order)
            // rank of wave j in the order [waves of even workgroups, then waves of odd ones]: even workgroups hold j with (j >> 2) even
            const int ne = wps / 2;                                   // waves in even workgroups
            const int re = r < ne ? r : ne, ro = r - re;              // extras that go to even / odd workgroups
            const int je = (j >> 3) * 4 + (((j >> 2) & 1) ? 4 : (j & 3)), jo = (j >> 3) * 4 + (((j >> 2) & 1) ? (j & 3) : 0);      // waves of even / odd workgroups among 0 .. j-1
            return (je < re ? je : re) + (jo < ro ? jo : ro);
        };
        tb = q * jw + extra_before(jw);
        te = q * (jw + 1) + extra_before(jw + 1);
    }
    const unsigned aw1 = *am.w1, aw2 = *am.w2, ab1 = *am.b1;
    const int ew1 = h3_exp_w(aw1), ew2 = h3_exp_w(aw2);
    int ex = 0, ed = 0, eh = 0, eg = 0;
    if (active) {
        const unsigned ax = am.x[n], ad = am.dt[n];
        ex = __builtin_amdgcn_readfirstlane(h3_exp(ax)); ed = __builtin_amdgcn_readfirstlane(h3_exp(ad));
        eh = __builtin_amdgcn_readfirstlane(h3_exp(32.f * __uint_as_float(ax) * __uint_as_float(aw1) + __uint_as_float(ab1)));
        eg = __builtin_amdgcn_readfirstlane(h3_exp((float)D * __uint_as_float(ad) * __uint_as_float(aw2)));
    }
    auto clampexp = [](int k) { return k < -126 ? -126 : k; };
    const float sx = pow2i(ex), sd = pow2i(ed), ch = pow2i(clampexp(eh - ex - ew1)), cg = pow2i(clampexp(eg - ew2 - ed));

    // ---- prologue: weights, tables, zeroed images ----
    {   // the three fragment sets, 96 KB: all of a thread's 24 requests first, then its stores -- one round trip (a copy loop waits for every request in turn: 9.6 us of prologue)
        uint4 wq[24];
#pragma unroll
        for (int k = 0; k < 8; ++k) { wq[k] = w1f[tid + 256 * k]; wq[8 + k] = w2kf[tid + 256 * k]; wq[16 + k] = w1cf[tid + 256 * k]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 24; ++k) reinterpret_cast<uint4*>(lds)[(k >> 3) * (PW4_WSET / 16) + tid + 256 * (k & 7)] = wq[k];
    }
    for (int i = lane; i < PW4_SB / 16; i += 64) reinterpret_cast<uint4*>(PW)[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) SB[lane + 64 * k] = b1[lane + 64 * k] * pow2i(eh);
    if (lane < 32) EX[lane] = -(h3_exp_w(am.w1r[lane]) + eg);
    __syncthreads();

    // ---- per-lane addresses ----
    const int li = lane & 15, gcol = (lane >> 4) & 1;
    const int rowo = col * PB_ROW + half * 16;                                       // row reads of a piece image: + kb * 32 + p * PB_IMG
    const int tro = (4 * half + (li >> 2)) * PB_ROW + (16 * gcol + 4 * (li & 3)) * 2;    // transposed reads (tr_frag<PB_ROW>): + 16 kb * PB_ROW (+ 8 * PB_ROW) + p * PB_IMG
    const int s0 = col * 64 + ((half ^ ps_key(col)) << 3);                           // dH' image, stores: s0 ^ (G << 4)
    int t0;
    { const int r0 = 4 * half + (li >> 2); t0 = r0 * 64 + (((4 * gcol + (li & 3)) ^ r0) << 3); }
    auto toff = [&](int kb, int q) { return (t0 ^ ((2 * kb + q) << 3)) + 1024 * kb + 512 * q; };
    const int wfo = lane * 16;                                                       // a weight fragment: + ((c * 2 + kb) * 2 + p) * 1024

    // ---- global rows through per-tile buffer descriptors (pw_bwd_h3t_kernel): rows beyond a tile load zeros / store nothing; a ghost tile has no rows ----
    // rows of tile t: none in front of the run (the ghost whose dX rows the first boundary would store) and none beyond the sample; a tile behind the run's end that
    // still belongs to the sample is the next wave's first one -- staging it into the last boundary is harmless (nothing consumes it) and keeps this branch-free
    auto t_rows = [&](int t) { int r = vps - 32 * t; r = r < 32 ? r : 32; r = r < 0 ? 0 : r; return t < tb ? 0 : r; };
    auto tile_rsrc = [&](const float* base, int t, int row_bytes) {
        const int tc = t < 0 ? 0 : t;
        const unsigned long p = reinterpret_cast<unsigned long>(base) + (unsigned long)(((long)n * vps + 32L * tc) * row_bytes);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(((unsigned long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(t_rows(t) * row_bytes), 0x00020000);
    };

    // ---- state ----
    f32x16 dW1[8], dW2t[8];                                        // a[0:255]
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dW1[c][r] = 0.f; dW2t[c][r] = 0.f; }
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    f32x16 H = zero, dH = zero, dx = zero;                         // VGPRs
    Frag hp[2][2], gp[2][2];                                       // [k-block][piece]: cut H'^T / dH'^T of the chunk before the current one
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) { hp[kb][p].u = make_uint4(0u, 0u, 0u, 0u); gp[kb][p].u = make_uint4(0u, 0u, 0u, 0u); }
    Frag w2c[2][2];                                                // W2 fragments of (b) of the coming chunk
    Frag xf[2][2], df[2][2];                                       // this lane's rows of the tile's X and dT images (A operands of (a) and (b)): read once per tile, in the boundary
    Frag at[2][2], ae[2][2];                                       // the same images transposed (A operands of (d) and (e)): likewise
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) { at[kb][p].u = make_uint4(0u, 0u, 0u, 0u); ae[kb][p].u = make_uint4(0u, 0u, 0u, 0u); }
    float bc = 0.f;                                                // its bias (this lane's hidden channel)
    float bs1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // db1 partials: hidden 32 c + col, this lane's 16 voxel rows of every tile
    float bs2 = 0.f;                                               // db2 partial: out channel col (rows of dT^T), this lane's voxel slots
    float omax = 0.f;
    u32x4b xr[4]; unsigned dr[16]; u32x4b dor[4];                  // raw rows in flight: X / dT of the next tile, dOut of the current one
#pragma unroll
    for (int k = 0; k < 4; ++k) { xr[k] = u32x4b{0u, 0u, 0u, 0u}; dor[k] = u32x4b{0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int k = 0; k < 16; ++k) dr[k] = 0u;

#ifdef PW4_STAMP
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long st_c0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto SBAR = [] { __builtin_amdgcn_sched_barrier(0); };
    // raw rows in flight, requested in pieces from the gaps of chunks 3 .. 5 (a gap takes two or three requests beside its vector work)
    __amdgpu_buffer_rsrc_t rsx = tile_rsrc(x, tb, 128), rsd = tile_rsrc(dT, tb, 4 * D), rso = tile_rsrc(dOut, tb - 1, 128);
    auto load_x = [&](int k) __attribute__((always_inline)) {         // X rows of the next tile: this lane's cin 16 kb + 8 half + (0..7) of voxel col, k = 2 kb + (0 | 1)
        xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsx, col * 128 + half * 32, (k >> 1) * 64 + (k & 1) * 16, 0);
    };
    // dT rows of the next tile: like X, this lane's out channels 16 kb + 8 half + (0..7) of voxel col, k = 8 kb + (0..7) -- single dwords (a row is 4 D bytes: no wider
    // alignment).  Channels >= D of the last slice are the NEXT row's first values: they are multiplied by zero when the slice is cut (sdz)
    auto load_d = [&](int k) __attribute__((always_inline)) {
        dr[k] = __builtin_amdgcn_raw_buffer_load_b32(rsd, col * 4 * D + half * 32, (k >> 3) * 64 + (k & 7) * 4, 0);
    };
    // The rows are REQUESTED late (chunk 7: their registers are those of the tile's X / dT fragments, dead by then) and TOUCHED early (chunk 2: one byte of every
    // 128-byte line of the three row blocks, so that the requests of chunk 7 find them in the L2 instead of waiting for HBM in front of the boundary)
    unsigned touched[3] = {0u, 0u, 0u};                             // (their destinations: read by nobody but an empty asm statement at the boundary -- a use right behind the loads would wait for them)
    auto touch_next = [&]() __attribute__((always_inline)) {
        touched[0] = __builtin_amdgcn_raw_buffer_load_b32(rsx, lane * 64, 0, 0);
        touched[1] = __builtin_amdgcn_raw_buffer_load_b32(rsd, lane * 64, 0, 0);
        touched[2] = __builtin_amdgcn_raw_buffer_load_b32(rso, lane * 64, 0, 0);
    };
    auto load_o = [&](int G) __attribute__((always_inline)) {         // dOut of the current tile: cin 8 G + 4 half + (0..3) of voxel col
        dor[G] = __builtin_amdgcn_raw_buffer_load_b128(rso, col * 128 + half * 16, 32 * G, 0);
    };
    // LDS reads
#if defined(PW4_ABL_NOLDS) || defined(PW4_ABL_NOLDS_W) || defined(PW4_ABL_NOLDS_R) || defined(PW4_ABL_NOLDS_T) || defined(PW4_ABL_NOLDS_G)
    auto undef = [](Frag& f) { asm volatile("" : "=v"(f.u.x), "=v"(f.u.y), "=v"(f.u.z), "=v"(f.u.w)); };
#endif
#ifdef PW4_ABL_NOLDS
    auto rd_w = [&](const unsigned char*, int, int, int, Frag& f) { undef(f); };
    auto rd_row = [&](const unsigned char*, int, int, Frag& f) { undef(f); };
    auto rd_tr = [&](const unsigned char*, int, int, Frag& f) { undef(f); };
    auto rd_g = [&](int, int, Frag& f) { undef(f); };
#else
#ifdef PW4_ABL_NOLDS_W
    auto rd_w = [&](const unsigned char*, int, int, int, Frag& f) { undef(f); };
#else
    auto rd_w = [&](const unsigned char* set, int c, int kb, int p, Frag& f) { f.u = *reinterpret_cast<const uint4*>(set + wfo + ((c * 2 + kb) * 2 + p) * 1024); };
#endif
#ifdef PW4_ABL_NOLDS_R
    auto rd_row = [&](const unsigned char*, int, int, Frag& f) { undef(f); };
#else
    auto rd_row = [&](const unsigned char* img, int kb, int p, Frag& f) { f.u = *reinterpret_cast<const uint4*>(img + rowo + kb * 32 + p * PB_IMG); };
#endif
#ifdef PW4_ABL_NOLDS_T
    auto rd_tr = [&](const unsigned char*, int, int, Frag& f) { undef(f); };
#else
    auto rd_tr = [&](const unsigned char* img, int kb, int p, Frag& f) {
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const unsigned char* q = img + tro + 16 * kb * PB_ROW + p * PB_IMG;
        f.hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q));
        f.hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q + 8 * PB_ROW));
    };
#endif
#ifdef PW4_ABL_NOLDS_G
    auto rd_g = [&](int, int, Frag& f) { undef(f); };
#else
    auto rd_g = [&](int kb, int p, Frag& f) { tr_frag_sw(GI + p * PS_IMG, toff(kb, 0), toff(kb, 1), f); };
#endif
#endif
    // ---- the vector work of one chunk, in pieces sized for the gaps between MFMAs.  One wave per SIMD issues one vector instruction per ~4.9 cycles (v_cvt_pk 8,
    // v_fma_mix 8.8; MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): the stream is bound by what it ISSUES (~880 cycles of vector work per chunk beside 960 of
    // matrix pipe), so every gap carries ~30 cycles of it and nothing that waits: no v_cmp -> v_cndmask pair (a write of VCC and two wait states: 21 cycles per element).
    // The gate is arithmetic.  H holds relu(hv) >= +0, and a positive H is never subnormal (it is the rounded sum of two fp32 numbers of ordinary magnitude: zero, or
    // at least 2^-24 of the larger one), so H * 2^127 is 0 or >= 2^3 > cg (cg <= 2^-17, see the bounds): min(H * 2^127, cg) = (H > 0 ? cg : 0) exactly, and dH' = dH * that.
#ifdef PW4_ABL_NOVALU
    auto relu1 = [&](int e) { asm volatile("" : "+v"(H[e])); };
    auto gate1 = [&](int e) { asm volatile("" : "+v"(dH[e]) : "v"(H[e])); };
    auto db1add = [&](int e, float& bs) { asm volatile("" : "+v"(bs) : "v"(dH[e])); };
    auto cutHA = [&](int g, uint2& q0, uint2& q1) { asm volatile("" : "=v"(q0.x), "=v"(q0.y), "=v"(q1.x), "=v"(q1.y) : "v"(H[4 * g]), "v"(H[4 * g + 3])); };
    auto cutHB = [&](int g, const uint2& q0, uint2& q1) { asm volatile("" : "+v"(hp[g >> 1][0].u.x), "+v"(hp[g >> 1][1].u.x) : "v"(q0.x), "v"(q0.y), "v"(q1.x)); };
    auto cutGA = [&](int g, uint2& q0, uint2& q1) { asm volatile("" : "=v"(q0.x), "=v"(q0.y), "=v"(q1.x), "=v"(q1.y) : "v"(dH[4 * g]), "v"(dH[4 * g + 3])); };
    auto cutGB = [&](int g, const uint2& q0, uint2& q1) { asm volatile("" : "+v"(gp[g >> 1][0].u.x), "+v"(gp[g >> 1][1].u.x) : "v"(q0.x), "v"(q0.y), "v"(q1.x)); };
#else
    const float big = __uint_as_float(0x7f000000u);               // 2^127
    auto relu1 = [&](int e) { H[e] = fmaxf(fmaf(H[e], ch, bc), 0.f); };
    auto gate1 = [&](int e) { dH[e] = dH[e] * fminf(H[e] * big, cg); };
    auto db1add = [&](int e, float& bs) { bs += dH[e]; };
    // the cut of registers 4g .. 4g+3 (voxel rows 8g + 4 half + (0..3)) in two parts: A = the first pieces (two v_cvt_pk), B = the second ones (four v_fma_mix) and their
    // place in the fragments: dwords (g & 1) * 2, + 1 of k-block g >> 1
    // the second pieces of a group in two halves (two v_fma_mix each; the partial writes of one register stay one instruction apart, as in h3_second_pieces2)
    auto mixlo2 = [](unsigned h0a, float a0, unsigned h0b, float b0, unsigned& ra, unsigned& rb) {
        asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=&v"(ra), "=&v"(rb) : "v"(h0a), "v"(h0b), "v"(a0), "v"(b0));
    };
    auto mixhi2 = [](unsigned h0a, float a1, unsigned h0b, float b1, unsigned& ra, unsigned& rb) {
        asm("v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %1, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ra), "+v"(rb) : "v"(h0a), "v"(h0b), "v"(a1), "v"(b1));
    };
    auto cutHA = [&](int g, uint2& q0, uint2& q1) {                // first pieces + the low halves of the second ones
        const f32x2 va = {H[4 * g], H[4 * g + 1]}, vb = {H[4 * g + 2], H[4 * g + 3]};
        q0.x = __builtin_bit_cast(unsigned, __builtin_convertvector(va, f16x2)); q0.y = __builtin_bit_cast(unsigned, __builtin_convertvector(vb, f16x2));
        mixlo2(q0.x, H[4 * g], q0.y, H[4 * g + 2], q1.x, q1.y);
    };
    auto cutHB = [&](int g, const uint2& q0, uint2& q1) {          // the high halves, and the group's place in the fragments
        mixhi2(q0.x, H[4 * g + 1], q0.y, H[4 * g + 3], q1.x, q1.y);
        if (g & 1) { hp[g >> 1][0].u.z = q0.x; hp[g >> 1][0].u.w = q0.y; hp[g >> 1][1].u.z = q1.x; hp[g >> 1][1].u.w = q1.y; }
        else { hp[g >> 1][0].u.x = q0.x; hp[g >> 1][0].u.y = q0.y; hp[g >> 1][1].u.x = q1.x; hp[g >> 1][1].u.y = q1.y; }
    };
    auto cutGA = [&](int g, uint2& q0, uint2& q1) {
        const f32x2 va = {dH[4 * g], dH[4 * g + 1]}, vb = {dH[4 * g + 2], dH[4 * g + 3]};
        q0.x = __builtin_bit_cast(unsigned, __builtin_convertvector(va, f16x2)); q0.y = __builtin_bit_cast(unsigned, __builtin_convertvector(vb, f16x2));
        mixlo2(q0.x, dH[4 * g], q0.y, dH[4 * g + 2], q1.x, q1.y);
    };
    auto cutGB = [&](int g, const uint2& q0, uint2& q1) {
        mixhi2(q0.x, dH[4 * g + 1], q0.y, dH[4 * g + 3], q1.x, q1.y);
        if (g & 1) { gp[g >> 1][0].u.z = q0.x; gp[g >> 1][0].u.w = q0.y; gp[g >> 1][1].u.z = q1.x; gp[g >> 1][1].u.w = q1.y; }
        else { gp[g >> 1][0].u.x = q0.x; gp[g >> 1][0].u.y = q0.y; gp[g >> 1][1].u.x = q1.x; gp[g >> 1][1].u.y = q1.y; }
        // the same dwords to the dH' image [hidden = col][voxel]: (c) reads it transposed
        *reinterpret_cast<uint2*>(GI + (s0 ^ (g << 4))) = q0;
        *reinterpret_cast<uint2*>(GI + PS_IMG + (s0 ^ (g << 4))) = q1;
    };
#endif

    // ---- one chunk iteration: 30 gaps.  C: the chunk whose vector work runs here; PREV: (e), (d), (c) of chunk C-1 are issued; NEXT: (a) of chunk C+1 and the
    //      operands of its (b).  Order of the matrix work: (b) C | (e) C-1 | (d) C-1 | (c) C-1 | (a) C+1.  Order of the vector work, and why it may stand where it does:
    //        gaps  2-7   bias + ReLU of H (gap 1 stays empty: H is complete two MFMAs behind the previous iteration's last one, and a read before that is padded with s_nop)
    //        gaps  8-14  the gate, two or three elements per gap (dH is complete one MFMA behind gap 6)
    //        gaps 15-22  the cut of H' (hp is free: (e) C-1 has been issued) and the db1 sums        -- H is dead behind gap 22: (a) C+1 starts in gap 25
    //        gaps 23-30  the cut of dH' (gp and the image are free: (d) C-1 has been issued, the image's reads stand in front of these stores in the wave's LDS order)
    //      LDS reads are requested four to six gaps ahead of the MFMA that takes them. ----
    auto iter = [&](auto c_tag, auto prev_tag, auto next_tag, int t) __attribute__((always_inline)) {
        constexpr int C = decltype(c_tag)::value;
        constexpr bool PREV = decltype(prev_tag)::value, NEXT = decltype(next_tag)::value;
        constexpr int P = C - 1, N = C + 1;
        Frag gq[2][2], w3[2][2], w1n[2][2];
        uint2 qa, qb;
        float& bs = bs1[C];
        PW4_ST(4);
        SBAR();
        dH = PW4_MFMA_V(df[0][0], w2c[0][1], zero);                                    // gap 1
        SBAR();
        SBAR();
        dH = PW4_MFMA_V(df[0][1], w2c[0][0], dH);                                      // 2
        SBAR();
        relu1(0); relu1(1); relu1(2);
        SBAR();
        dH = PW4_MFMA_V(df[0][0], w2c[0][0], dH);                                      // 3
        SBAR();
        relu1(3); relu1(4); relu1(5);
        SBAR();
        dH = PW4_MFMA_V(df[1][0], w2c[1][1], dH);                                      // 4
        SBAR();
        relu1(6); relu1(7); relu1(8);
        SBAR();
        dH = PW4_MFMA_V(df[1][1], w2c[1][0], dH);                                      // 5
        SBAR();
        relu1(9); relu1(10); relu1(11);
        SBAR();
        dH = PW4_MFMA_V(df[1][0], w2c[1][0], dH);                                      // 6
        SBAR();
        relu1(12); relu1(13);
        if constexpr (C == 2) { rsx = tile_rsrc(x, t + 1, 128); rsd = tile_rsrc(dT, t + 1, 4 * D); rso = tile_rsrc(dOut, t, 128); }      // (scalar work; xr / dr were consumed by the staging in front of chunk 0)
        if constexpr (C == 2) touch_next();
        SBAR();
        PW4_ST(0);
        // (e) of chunk C-1: dW2c^T[out][hidden] += dT^T H'
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[0][1], hp[0][0]);                   // 7
        SBAR();
        relu1(14); relu1(15);
        if constexpr (C == 7) { load_x(0); load_x(1); }
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[0][0], hp[0][1]);                   // 8
        SBAR();
        if constexpr (C == 7) { load_x(2); load_x(3); }
        gate1(0); gate1(1); gate1(2);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[0][0], hp[0][0]);                   // 9
        SBAR();
        if constexpr (PREV) { rd_g(0, 1, gq[0][1]); rd_g(0, 0, gq[0][0]); }
        if constexpr (C == 7) { load_d(0); load_d(1); load_d(2); }
        gate1(3); gate1(4);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[1][1], hp[1][0]);                   // 10
        SBAR();
        if constexpr (PREV) { rd_g(1, 1, gq[1][1]); rd_g(1, 0, gq[1][0]); }
        if constexpr (C == 7) { load_d(3); load_d(4); load_d(5); }
        gate1(5); gate1(6);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[1][0], hp[1][1]);                   // 11
        SBAR();
        if constexpr (C == 7) { load_d(6); load_d(7); load_d(8); }
        gate1(7); gate1(8);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW2t[P], ae[1][0], hp[1][0]);                   // 12
        SBAR();
        if constexpr (C == 7) { load_d(9); load_d(10); load_d(11); }
        gate1(9); gate1(10);
        SBAR();
        PW4_ST(1);
        // (d) of chunk C-1: dW1c[cin][hidden] += X^T dH'
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[0][1], gp[0][0]);                    // 13
        SBAR();
        if constexpr (C == 7) { load_d(12); load_d(13); load_d(14); load_d(15); }
        if constexpr (PREV) { rd_w(lds + 2 * PW4_WSET, P, 0, 1, w3[0][1]); rd_w(lds + 2 * PW4_WSET, P, 0, 0, w3[0][0]); }
        gate1(11); gate1(12);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[0][0], gp[0][1]);                    // 14
        SBAR();
        if constexpr (C == 7) { load_o(0); load_o(1); load_o(2); load_o(3); }
        if constexpr (PREV) { rd_w(lds + 2 * PW4_WSET, P, 1, 1, w3[1][1]); rd_w(lds + 2 * PW4_WSET, P, 1, 0, w3[1][0]); }
        gate1(13); gate1(14); gate1(15);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[0][0], gp[0][0]);                    // 15
        SBAR();
        if constexpr (NEXT) { rd_w(lds, N, 0, 1, w1n[0][1]); rd_w(lds, N, 0, 0, w1n[0][0]); }
        cutHA(0, qa, qb);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[1][1], gp[1][0]);                    // 16
        SBAR();
        cutHB(0, qa, qb); db1add(0, bs); db1add(1, bs); db1add(2, bs); db1add(3, bs);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[1][0], gp[1][1]);                    // 17
        SBAR();
        if constexpr (NEXT) { rd_w(lds, N, 1, 1, w1n[1][1]); rd_w(lds, N, 1, 0, w1n[1][0]); }
        cutHA(1, qa, qb);
        SBAR();
        if constexpr (PREV) PW4_MFMA_A(dW1[P], at[1][0], gp[1][0]);                    // 18
        SBAR();
        cutHB(1, qa, qb); db1add(4, bs); db1add(5, bs); db1add(6, bs); db1add(7, bs);
        SBAR();
        PW4_ST(2);
        // (c) of chunk C-1: dX^T[cin][voxel] += W1c dH'^T (B: the dH' image of chunk C-1)
        if constexpr (PREV) { if constexpr (P == 0) dx = PW4_MFMA_V(w3[0][1], gq[0][0], zero); else dx = PW4_MFMA_V(w3[0][1], gq[0][0], dx); }      // 19
        SBAR();
        cutHA(2, qa, qb);
        SBAR();
        if constexpr (PREV) dx = PW4_MFMA_V(w3[0][0], gq[0][1], dx);                   // 20
        SBAR();
        if constexpr (NEXT) { rd_w(lds + PW4_WSET, N, 0, 1, w2c[0][1]); rd_w(lds + PW4_WSET, N, 0, 0, w2c[0][0]); }
        cutHB(2, qa, qb); db1add(8, bs); db1add(9, bs); db1add(10, bs); db1add(11, bs);
        SBAR();
        if constexpr (PREV) dx = PW4_MFMA_V(w3[0][0], gq[0][0], dx);                   // 21
        SBAR();
        cutHA(3, qa, qb);
        SBAR();
        if constexpr (PREV) dx = PW4_MFMA_V(w3[1][1], gq[1][0], dx);                   // 22
        SBAR();
        if constexpr (NEXT) { rd_w(lds + PW4_WSET, N, 1, 1, w2c[1][1]); rd_w(lds + PW4_WSET, N, 1, 0, w2c[1][0]); }
        cutHB(3, qa, qb); db1add(12, bs); db1add(13, bs); db1add(14, bs); db1add(15, bs);
        SBAR();
        if constexpr (PREV) dx = PW4_MFMA_V(w3[1][0], gq[1][1], dx);                   // 23
        SBAR();
        cutGA(0, qa, qb);
        SBAR();
        if constexpr (PREV) dx = PW4_MFMA_V(w3[1][0], gq[1][0], dx);                   // 24
        SBAR();
        if constexpr (NEXT) bc = SB[32 * N + col];
        cutGB(0, qa, qb);
        SBAR();
        PW4_ST(3);
        // (a) of chunk C+1: H^T[voxel][hidden] = X W1c (H is free: its cut is done)
        if constexpr (NEXT) H = PW4_MFMA_V(xf[0][0], w1n[0][1], zero);                 // 25
        SBAR();
        cutGA(1, qa, qb);
        SBAR();
        if constexpr (NEXT) H = PW4_MFMA_V(xf[0][1], w1n[0][0], H);                    // 26
        SBAR();
        cutGB(1, qa, qb);
        SBAR();
        if constexpr (NEXT) H = PW4_MFMA_V(xf[0][0], w1n[0][0], H);                    // 27
        SBAR();
        cutGA(2, qa, qb);
        SBAR();
        if constexpr (NEXT) H = PW4_MFMA_V(xf[1][0], w1n[1][1], H);                    // 28
        SBAR();
        cutGB(2, qa, qb);
        SBAR();
        if constexpr (NEXT) H = PW4_MFMA_V(xf[1][1], w1n[1][0], H);                    // 29
        SBAR();
        cutGA(3, qa, qb);
        SBAR();
        if constexpr (NEXT) H = PW4_MFMA_V(xf[1][0], w1n[1][0], H);                    // 30
        SBAR();
        cutGB(3, qa, qb);
        SBAR();
    };

    // ---- between two tiles (24 gaps): (e), (d), (c) of the last chunk of tile t, its dX rows, the staging of tile t+1, (a) of that tile's chunk 0 and the operands of its
    //      (b).  The images' last reads for tile t are requested first: they stand in front of the staging stores in the wave's LDS order. ----
    auto boundary = [&](auto first_tag, int t) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        Frag gq[2][2], w3[2][2], w1n[2][2];
        float xs[8];
        uint2 q0, q1, r0, r1;
        // this lane's X values: cut into its fragment of the row image (cin 16 kb + 8 half + (0..7) of voxel col)
        auto xmul = [&](int kb) {
            const u32x4b a = xr[2 * kb], b = xr[2 * kb + 1];
            xs[0] = __uint_as_float(a.x) * sx; xs[1] = __uint_as_float(a.y) * sx; xs[2] = __uint_as_float(a.z) * sx; xs[3] = __uint_as_float(a.w) * sx;
            xs[4] = __uint_as_float(b.x) * sx; xs[5] = __uint_as_float(b.y) * sx; xs[6] = __uint_as_float(b.z) * sx; xs[7] = __uint_as_float(b.w) * sx;
        };
        auto xput = [&](int kb) {
            *reinterpret_cast<uint4*>(XI + rowo + kb * 32) = make_uint4(q0.x, q0.y, r0.x, r0.y);
            *reinterpret_cast<uint4*>(XI + PB_IMG + rowo + kb * 32) = make_uint4(q1.x, q1.y, r1.x, r1.y);
        };
        // element f = lane + 64 k of the next tile's dT block: both pieces to its place in the image (the place comes from a table: f / D and f % D per element
        // would be a dozen vector instructions)
        // this lane's dT values: the same, into the dT image (channels 16 kb + 8 half + (0..7) of voxel col; those >= D are zeros: the last slice of the upper lane half
        // holds channel 24 and seven values of the next row, which its scale sdz = 0 removes)
        const float sdz = half ? 0.f : sd;                             // (D = 25: slice kb = 1 of the upper half holds channel 24, then the next row)
        auto dmul = [&](int kb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[j] = __uint_as_float(dr[8 * kb + j]) * ((kb == 0 || j == 0) ? sd : sdz);
        };
        auto dputv = [&](int kb) {
            *reinterpret_cast<uint4*>(DI + rowo + kb * 32) = make_uint4(q0.x, q0.y, r0.x, r0.y);
            *reinterpret_cast<uint4*>(DI + PB_IMG + rowo + kb * 32) = make_uint4(q1.x, q1.y, r1.x, r1.y);
        };
        const unsigned ones = 0x3c003c00u;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        // db2[out] = sum over the voxels of dT: the rows of dT^T are on the lanes (out = col), eight voxel slots per lane and k-block: pieces summed by v_dot2 (fp32 accumulate)
        auto dsum = [&](const Frag& f) {
            bs2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f.u.x), __builtin_bit_cast(h2, ones), bs2, false);
            bs2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f.u.y), __builtin_bit_cast(h2, ones), bs2, false);
            bs2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f.u.z), __builtin_bit_cast(h2, ones), bs2, false);
            bs2 = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, f.u.w), __builtin_bit_cast(h2, ones), bs2, false);
        };
        const __amdgpu_buffer_rsrc_t ry = tile_rsrc(dX, t, 128);
        // the dX rows of tile t = dOut + the accumulator at true scale (registers 4G .. 4G+3 = cin 8G + 4 half + (0..3) of voxel col)
        int4 ex4[4];                                                   // (requested in gap 16, used from gap 20 on: a read right in front of its use waits out the LDS latency)
        auto xout = [&](int G) {
            const int4 e4 = ex4[G];
            float o[4];
            o[0] = ldexpf(dx[4 * G], e4.x) + __uint_as_float(dor[G].x); o[1] = ldexpf(dx[4 * G + 1], e4.y) + __uint_as_float(dor[G].y);
            o[2] = ldexpf(dx[4 * G + 2], e4.z) + __uint_as_float(dor[G].z); o[3] = ldexpf(dx[4 * G + 3], e4.w) + __uint_as_float(dor[G].w);
            omax = fmaxf(omax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));      // (rows beyond the tile: zeros -- their X, dT and dOut rows were)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4b{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])}, ry, col * 128 + half * 16, 32 * G, 0);
        };
        PW4_ST(4);
        asm volatile("" :: "v"(touched[0]), "v"(touched[1]), "v"(touched[2]));      // (the touch loads have destinations: this is their only reader)
        SBAR();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int p = 0; p < 2; ++p) { rd_g(kb, 1 - p, gq[kb][1 - p]); rd_w(lds + 2 * PW4_WSET, 7, kb, 1 - p, w3[kb][1 - p]); }
        xmul(0);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[0][1], hp[0][0]);                                     // gap 1
        SBAR();
        h3_cut4_scaled(xs[0], xs[1], xs[2], xs[3], q0, q1);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[0][0], hp[0][1]);                                     // 2
        SBAR();
        h3_cut4_scaled(xs[4], xs[5], xs[6], xs[7], r0, r1);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[0][0], hp[0][0]);                                     // 3
        SBAR();
        xput(0); xf[0][0].u = make_uint4(q0.x, q0.y, r0.x, r0.y); xf[0][1].u = make_uint4(q1.x, q1.y, r1.x, r1.y); xmul(1);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[1][1], hp[1][0]);                                     // 4
        SBAR();
        h3_cut4_scaled(xs[0], xs[1], xs[2], xs[3], q0, q1);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[1][0], hp[1][1]);                                     // 5
        SBAR();
        h3_cut4_scaled(xs[4], xs[5], xs[6], xs[7], r0, r1);
        SBAR();
        PW4_MFMA_B(dW2t[7], ae[1][0], hp[1][0]);                                     // 6
        SBAR();
        xput(1); xf[1][0].u = make_uint4(q0.x, q0.y, r0.x, r0.y); xf[1][1].u = make_uint4(q1.x, q1.y, r1.x, r1.y); dsum(ae[0][0]);
        SBAR();
        PW4_ST(5);
        PW4_MFMA_B(dW1[7], at[0][1], gp[0][0]);                                     // 7
        SBAR();
        dsum(ae[0][1]); dmul(0);
        SBAR();
        PW4_MFMA_B(dW1[7], at[0][0], gp[0][1]);                                     // 8
        SBAR();
        dsum(ae[1][0]); h3_cut4_scaled(xs[0], xs[1], xs[2], xs[3], q0, q1);
        SBAR();
        PW4_MFMA_B(dW1[7], at[0][0], gp[0][0]);                                     // 9
        SBAR();
        dsum(ae[1][1]); h3_cut4_scaled(xs[4], xs[5], xs[6], xs[7], r0, r1);
        SBAR();
        PW4_MFMA_B(dW1[7], at[1][1], gp[1][0]);                                     // 10
        SBAR();
        dputv(0); df[0][0].u = make_uint4(q0.x, q0.y, r0.x, r0.y); df[0][1].u = make_uint4(q1.x, q1.y, r1.x, r1.y); dmul(1);
        SBAR();
        PW4_MFMA_B(dW1[7], at[1][0], gp[1][1]);                                     // 11
        SBAR();
        h3_cut4_scaled(xs[0], xs[1], xs[2], xs[3], q0, q1);
        SBAR();
        PW4_MFMA_B(dW1[7], at[1][0], gp[1][0]);                                     // 12
        SBAR();
        h3_cut4_scaled(xs[4], xs[5], xs[6], xs[7], r0, r1);
        SBAR();
        dx = PW4_MFMA_V(w3[0][1], gq[0][0], dx);                                        // 13
        SBAR();
        dputv(1); df[1][0].u = make_uint4(q0.x, q0.y, r0.x, r0.y); df[1][1].u = make_uint4(q1.x, q1.y, r1.x, r1.y);
        rd_w(lds, 0, 0, 1, w1n[0][1]);
        SBAR();
        dx = PW4_MFMA_V(w3[0][0], gq[0][1], dx);                                        // 14
        SBAR();
        SBAR();
        dx = PW4_MFMA_V(w3[0][0], gq[0][0], dx);                                        // 15
        SBAR();
        rd_w(lds, 0, 0, 0, w1n[0][0]);
        rd_tr(XI, 0, 1, at[0][1]); rd_tr(XI, 0, 0, at[0][0]);
        SBAR();
        dx = PW4_MFMA_V(w3[1][1], gq[1][0], dx);                                        // 16
        SBAR();
        rd_w(lds, 0, 1, 1, w1n[1][1]);
        rd_tr(XI, 1, 1, at[1][1]); rd_tr(XI, 1, 0, at[1][0]);
#pragma unroll
        for (int G = 0; G < 4; ++G) ex4[G] = *reinterpret_cast<const int4*>(EX + 8 * G + 4 * half);
        SBAR();
        dx = PW4_MFMA_V(w3[1][0], gq[1][1], dx);                                        // 17
        SBAR();
        rd_w(lds, 0, 1, 0, w1n[1][0]);
        rd_tr(DI, 0, 1, ae[0][1]); rd_tr(DI, 0, 0, ae[0][0]);
        SBAR();
        dx = PW4_MFMA_V(w3[1][0], gq[1][0], dx);                                        // 18
        SBAR();
        rd_w(lds + PW4_WSET, 0, 0, 1, w2c[0][1]);
        rd_tr(DI, 1, 1, ae[1][1]); rd_tr(DI, 1, 0, ae[1][0]);
        SBAR();
        PW4_ST(6);
        // (a) of chunk 0 of tile t+1
        H = PW4_MFMA_V(xf[0][0], w1n[0][1], zero);                                      // 19
        SBAR();
        rd_w(lds + PW4_WSET, 0, 0, 0, w2c[0][0]);
        SBAR();
        H = PW4_MFMA_V(xf[0][1], w1n[0][0], H);                                         // 20
        SBAR();
        rd_w(lds + PW4_WSET, 0, 1, 1, w2c[1][1]);
        xout(0);
        SBAR();
        H = PW4_MFMA_V(xf[0][0], w1n[0][0], H);                                         // 21
        SBAR();
        rd_w(lds + PW4_WSET, 0, 1, 0, w2c[1][0]);
        xout(1);
        SBAR();
        H = PW4_MFMA_V(xf[1][0], w1n[1][1], H);                                         // 22
        SBAR();
        bc = SB[col];
        xout(2);
        SBAR();
        H = PW4_MFMA_V(xf[1][1], w1n[1][0], H);                                         // 23
        SBAR();
        xout(3);
        SBAR();
        H = PW4_MFMA_V(xf[1][0], w1n[1][0], H);                                         // 24
        SBAR();
        SBAR();
        PW4_ST(7);
    };

    // ---- the run ----
    if (te > tb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_x(k);
#pragma unroll
        for (int k = 0; k < 16; ++k) load_d(k);
        boundary(std::true_type(), tb - 1);                                          // (a ghost in front: zero pieces, zero image, no rows to store)
        for (int t = tb; t < te; ++t) {
            iter(std::integral_constant<int, 0>(), std::false_type(), std::true_type(), t);
            iter(std::integral_constant<int, 1>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 2>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 3>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 4>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 5>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 6>(), std::true_type(), std::true_type(), t);
            iter(std::integral_constant<int, 7>(), std::true_type(), std::false_type(), t);
            boundary(std::false_type(), t);
        }
        if (am.y) amax_commit(omax, am.y + n);
    }
#ifdef PW4_STAMP
    st_acc[8] = __builtin_amdgcn_s_memtime() - st_c0; st_acc[9] = __builtin_amdgcn_s_memrealtime() - st_r0;
    if (lane == 0 && gw < 1024) { for (int k = 0; k < 10; ++k) g_pw4_stamps[gw * 16 + k] = st_acc[k]; g_pw4_stamps[gw * 16 + 10] = st_r0 - st_entry; g_pw4_stamps[gw * 16 + 11] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- the four waves' sums at true scale meet in LDS (fixed order), one slab per workgroup: [dW1 32x256 | dW2 256xD | db1 256 | db2 D] ----
    const long slab_floats = 8192 + 256 * (long)D + 256 + D;
    float* sl = slabs + (long)blockIdx.x * slab_floats;
    // (the lane / thread index is taken afresh: a value kept across the tile loop for this epilogue would cost the loop a register -- or a spill)
    const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), tid2 = wave * 64 + lane2;
    float* R = reinterpret_cast<float*>(lds);                      // [4 waves][8 chunks][16 registers][64 lanes]
    __syncthreads();                                               // every wave is out of its loop: the weights and images are dead
#ifdef PW4_STAMP
    if (lane2 == 0 && gw < 1024) g_pw4_stamps[gw * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) R[((wave * 8 + c) * 16 + r) * 64 + lane2] = ldexpf(dW1[c][r], -(ex + eg));
    __syncthreads();
    for (int e4 = tid2; e4 < 2048; e4 += 256) {                    // four consecutive lanes = four consecutive hidden channels of one cin row: 16-byte LDS reads
        const float4 a = reinterpret_cast<const float4*>(R)[e4], b = reinterpret_cast<const float4*>(R + 8192)[e4],
                     c4 = reinterpret_cast<const float4*>(R + 2 * 8192)[e4], d = reinterpret_cast<const float4*>(R + 3 * 8192)[e4];
        const int e = 4 * e4, l = e & 63, r = (e >> 6) & 15, c = e >> 10;
        float* o = sl + (long)rowmap(r, l >> 5) * 256 + 32 * c + (l & 31);                      // [cin][hidden]  (a slab is an odd number of floats long: no 16-byte stores)
        o[0] = ((a.x + b.x) + c4.x) + d.x; o[1] = ((a.y + b.y) + c4.y) + d.y; o[2] = ((a.z + b.z) + c4.z) + d.z; o[3] = ((a.w + b.w) + c4.w) + d.w;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) R[((wave * 8 + c) * 16 + r) * 64 + lane2] = ldexpf(dW2t[c][r], -(ed + eh));
    float* R1 = R + 4 * 8192;                                      // [4 waves][8 chunks][64 lanes] db1 partials, then [4 waves][64 lanes] db2 partials
#pragma unroll
    for (int c = 0; c < 8; ++c) R1[(wave * 8 + c) * 64 + lane2] = ldexpf(bs1[c], -eg);
    R1[4 * 8 * 64 + wave * 64 + lane2] = ldexpf(bs2, -ed);
    __syncthreads();
    for (int o = tid2; o < 256 * D; o += 256) {                    // [hidden][out], walked in the slab's own order: consecutive threads store consecutive floats
        const int hid = o / D, out = o - hid * D;
        const int e = (((hid >> 5) * 16 + (out & 3) + 4 * (out >> 3)) * 64) + (hid & 31) + 32 * ((out >> 2) & 1);      // register r, lane half with rowmap(r, half) = out
        sl[8192 + o] = ((R[e] + R[8192 + e]) + R[2 * 8192 + e]) + R[3 * 8192 + e];
    }
    {
        const int c = tid2 >> 5, hcol = tid2 & 31;                   // db1[hidden 32 c + hcol]: both lane halves of the four waves
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += R1[(w * 8 + c) * 64 + hcol] + R1[(w * 8 + c) * 64 + 32 + hcol];
        sl[8192 + 256 * (long)D + tid2] = v;
        if (tid2 < D) {
            float u = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) u += R1[4 * 8 * 64 + w * 64 + tid2] + R1[4 * 8 * 64 + w * 64 + 32 + tid2];
            sl[8192 + 256 * (long)D + 256 + tid2] = u;
        }
    }
#ifdef PW4_STAMP
    if (lane2 == 0 && gw < 1024) g_pw4_stamps[gw * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
}

static int g_pw4_enabled = -1;
bool pw4_enabled()
{
    if (g_pw4_enabled < 0) { const char* e = getenv("PROBAV_GEN1"); g_pw4_enabled = !(e && (e[0] == '1' || (e[0] == 'p' && e[1] == 'w' && e[2] != 'f'))); }      // PROBAV_GEN1 = 1 (every general form) | pw (both pointwise kernels) | pwf | pwb (forward / backward only) | conv
    return g_pw4_enabled != 0;
}
void pw4_set_enabled(int on) { g_pw4_enabled = on ? 1 : 0; }

bool pw4_backward_supported(long nvox, long vps, int D)
{
    if (vps <= 0 || nvox % vps || D != 25) return false;        // (the dT staging knows where channel D - 1 sits in a lane's slices; every network of the reference has D = int(32 * 0.8) = 25)
    const long nsamp = nvox / vps;
    return nsamp >= 1 && nsamp <= 4L * mfma_pw_backward_grid() && vps < (1L << 26);
}

int pw4_backward(const float* x, const float* dT, const float* dOut, const float* w1f, const float* w2kf, const float* w1cf,
                 const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2, float* slabs, long nvox, long vps, int D,
                 const PwAmax& am, hipStream_t s)
{
    static std::once_flag once;
    std::call_once(once, [] { note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_bwd_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    if (!pw4_backward_supported(nvox, vps, D)) { set_error("pw4_backward: unsupported shape", hipSuccess); return PROBAV_EINVAL; }
    if (!am.x || !am.w1 || !am.w2 || !am.b1 || !am.dt || !am.w1r) { set_error("pw4_backward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
    const int grid = mfma_pw_backward_grid(), nsamp = (int)(nvox / vps);
    const int wps = 4 * grid / nsamp;                              // waves per sample (>= 1)
    hipLaunchKernelGGL(pw_bwd_w4_kernel, dim3(grid), dim3(256), PW4_LDS, s, x, dT, dOut, (const uint4*)w1f, (const uint4*)w2kf, (const uint4*)w1cf,
                       b1, dX, slabs, nsamp, (int)vps, D, wps, am);
    int rc = check_launch("pw_bwd_w4");
    if (rc) return rc;
    return mfma_pw_backward_reduce(slabs, D, dW1, dW2, db1, db2, s);
}

#ifdef PW4_DIAG
}  // namespace diag
#endif
}  // namespace probav
