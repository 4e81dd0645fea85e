// 3x3x3 convolution of the residual blocks (normConv forward: 25 -> 32 channels + skip; its backward-data: 32 -> 25), H3 arithmetic --
// ONE WAVE PER SIMD, the filter's first pieces in the accumulator half of the register file (round 5).
// Reference semantics: models/modelsTF.py:179-186 (ResConv3D: normConv_i = weight-normalised Conv3D, 'same' padding, + the block's skip) and
// tape.gradient through it (backward-data = the same convolution with the flipped, transposed filter: engine.hip packs it that way).
//
// What rounds 3 / 4 measured on conv3_pp_kernel (docs/notebook_r1-r5.md appendix A.4): 88 100 cycles of matrix work per SIMD, 183 000-187 000 cycles per wave.  The tap loop
// streams 90-108 KB of filter fragments per 32-voxel tile from the vector L1 / L2 (they do not fit the LDS beside the ring) -- 64 B/clk/CU is what a tap costs,
// not its MFMAs -- and the finishing half of the workgroup (epilogue, staging) adds another 30 000 cycles beside it.
// This kernel removes the stream instead of tuning it:
//   * THE FILTER STAYS IN REGISTERS.  A wave keeps the FIRST piece of every filter fragment (45 / 54 k-blocks x 4 registers = 180 / 216) in a[0:...] for the
//     whole launch -- an MFMA reads its A operand from there directly -- and the second pieces (one product of three) in LDS (45 / 46 KB; the 32-channel form
//     keeps eight of them in registers too, so that the six-row ring still fits).  No vector-memory instruction is left in the tap loop.
//   * ONE INSTRUCTION STREAM.  Four waves per workgroup (one per SIMD); a round = four 32-voxel tiles (one per wave) of the strip's flattened voxel stream.
//     A wave's round is 135 / 162 MFMAs on one accumulator; the epilogue of its PREVIOUS tile (scale, bias, ReLU, skip, amax, 16-byte stores), the staging
//     of the rows the NEXT round needs (loads early, cut + ds_write late), the skip loads and the next tile's addresses are dealt out over the MFMA gaps,
//     a few vector instructions per gap (an MFMA holds the SIMD's issue for 8 of its 32 cycles: MI355X_MICROARCH.md).  Two accumulator sets alternate
//     (the loop body is two rounds), so no register copy stands between a tile and the next.
//   * PLANAR RING, COMPILE-TIME GEOMETRY.  A ring row is eight planes (piece, 8-channel chunk) of 16-byte entries; depth, plane pitch and the MFMA k-order
//     are template constants, so a tap is an IMMEDIATE offset of the ds_read_b128 from the lane's entry: one v_add per k-block (25 channels: the lane halves
//     read different chunks) or none (32 channels).
//   * ONE BARRIER PER ROUND, in the middle of the tile: the k-blocks in front of it read row dh = 0 of their voxels only, which the previous round's
//     readers had already; what a round stages is first read behind the next barrier.  The matrix pipe does not drain at it.
// Products per k-block and their order are conv3_pp_kernel's (w0 x1 + w1 x0 + w0 x0); the sum over the k-blocks is ONE chain here (there: two waves'
// partial sums added), so results agree to rounding, not bit for bit.  Filter fragments: PACK_H3_CONVP (25 channels) / PACK_H3_CONV (32), unchanged.
#include "kernels_x6.h"
#include "x6_device.h"
#include <cstdlib>
#include <mutex>
#include <algorithm>

namespace probav {
#ifdef CW4_DIAG                 // tools/cw4bench.hip includes this file as probav::diag (stamped / ablated builds beside the product's copy in the library)
namespace diag {
#endif

struct Cw4Args {
    ConvGeom g;
    int Wp, Wt, nsplit, SR, nstrips, nslot;
    unsigned mTo, mNvr, mTi, mNslot;
};

namespace {
__device__ __forceinline__ int qdiv(int v, int d, unsigned m) { return (int)__umulhi((unsigned)v, m) + (d == 1 ? v : 0); }
unsigned qmagic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
constexpr int CW4_PSE = 151;                  // entries per plane: Wp * To records and 34 zero entries behind them (conv3_w4_kernel, tile_step); ODD multiple of 16 bytes, so that the
                                              // planes of one voxel -- the lanes of a staging store -- start in different banks
constexpr int CW4_PS = 16 * CW4_PSE;          // plane pitch, bytes
constexpr int CW4_ROW = 8 * CW4_PS;           // ring row, bytes
constexpr unsigned CW4_OOB = 0x40000000u;     // a buffer offset beyond every num_records: loads return 0, stores are dropped
}

#ifdef CW4_STAMP
__device__ unsigned long long g_cw4_stamps[1024 * 8];
#endif

#define CW4_SBAR() __builtin_amdgcn_sched_barrier(0)
#ifdef CW4_STAMP                // diagnostic build only: cycles per phase summed over a wave's rounds -- [wave][slot]: 0 whole kernel, 1 its 100-MHz ticks, 2 prologue, 3 round start -> barrier, 4 wait at the barrier, 5 barrier -> round end
#define CW4_ST(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#ifdef CW4_STAMP_ROUNDS
#define CW4_STR(k) CW4_ST(k)
#else
#define CW4_STR(k) do { } while (0)
#endif
#else
#define CW4_STR(k) do { } while (0)
#define CW4_ST(k) do { } while (0)
#endif

template <int CIN, int COUT, int TP, bool RELU, bool SKIP, bool BIAS>      // RELU / SKIP / BIAS false: the layer has none, and no instruction of it is in the tile loop; true: g.relu / skip / bias decide at run time
__global__ __launch_bounds__(256, 1) void conv3_w4_kernel(Cw4Args a, const float* __restrict__ x, const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                         const float* __restrict__ skip, float* __restrict__ y, Amax am)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NST = CIN == 25 ? 5 : 6, KB = 9 * NST, G = 3 * KB;      // k-blocks of 16 per (dh, dw) group / per tile; MFMAs (= gaps) per tile
    constexpr int NW1R = CIN == 25 ? 12 : 8;                              // second-piece fragments kept in registers (the others: LDS)
    constexpr int NCH = CIN == 25 ? 3 : 4;                               // 8-channel chunks a staged voxel is cut into (25 channels: + the gathered channel-24 chunk)
    constexpr int PS = CW4_PS, ROW = CW4_ROW;
    constexpr int PD = 2;                                                // k-blocks the operand reads run ahead
    constexpr int TO = TP - 2;                                           // depth of the layer = entries per column of a ring row (DENSE: no depth pads, see tile_step)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = a.nstrips * a.nsplit;
    const int n = blockIdx.x / per, srem = blockIdx.x - n * per;
    const int strip = srem / a.nsplit, sp = srem - strip * a.nsplit;
    const int ws0 = sp * a.Wt, hb = strip * a.SR;
    const int SRr = g.Ho - hb < a.SR ? g.Ho - hb : a.SR;
    const int nvr = a.Wt * g.To, NV = SRr * nvr, NTL = (NV + 31) >> 5, nround = (NTL + 3) >> 2;
    const int NS = a.nslot;
    float omax = 0.f;
#ifdef CW4_STAMP
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_prev = st_t0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // ---- staging constants (all 256 threads; a thread's items are the same for every row) ----
    // item i <-> (local voxel lv = i / NCH, chunk cc = i % NCH); 25 channels: channels 0..23 in three chunks, the voxel's fourth chunk GATHERS channel 24 of padded
    // depths t', t' + 1, t' + 2 and is one extra item per voxel (thread lv).  Planar ring: entry (lw * TP + t') of plane (piece, chunk).  A thread beyond the
    // last item repeats the last one (the same bytes to the same entry): no predicate anywhere in the staging.
    const int lw0 = a.nsplit == 1 ? g.pw : 0, Wl = a.nsplit == 1 ? g.Wi : a.Wt + 2;
    const int nvs = Wl * g.Ti, items = nvs * NCH;                        // plan: items <= 512, nvs <= 256
    const int ea = h3_exp(am.x[n]);
    const float sa = pow2i(ea);
    int s_srcb[2], s_offb[2];                                            // byte offset of the item inside an input row / inside a ring row
    float s_sc[2];                                                       // the sample's scale, 0 for a column outside the patch
    int s3_o[3], s3_offb = 0;                                            // gathered chunk: byte offsets of channel 24 at depths t - 1, t, t + 1 (clamped); its entry
    float s3_sc[3] = {0.f, 0.f, 0.f};
    {
        auto locate = [&](int lv, int& lw, int& t, int& iwc, int& colok) {
            const int lwr = qdiv(lv, g.Ti, a.mTi);
            t = lv - lwr * g.Ti; lw = lw0 + lwr;
            const int iw = ws0 + lw - g.pw;
            colok = (iw >= 0 ? 1 : 0) & (iw < g.Wi ? 1 : 0);
            iwc = iw < 0 ? 0 : (iw < g.Wi ? iw : g.Wi - 1);
        };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int it = tid + 256 * k;
            const int ic = it < items ? it : items - 1;
            const int lv = CIN == 25 ? (int)(((unsigned)ic * 43691u) >> 17) : ic >> 2;      // ic / 3 (ic < 2^16)
            const int cc = ic - lv * NCH;
            int lw, t, iwc, colok;
            locate(lv, lw, t, iwc, colok);
            s_srcb[k] = ((iwc * g.Ti + t) * CIN + 8 * cc) * 4;
            s_offb[k] = (lw * TO + t) * 16 + cc * PS;
            s_sc[k] = colok ? sa : 0.f;
        }
        if constexpr (CIN == 25) {
            const int lv = tid < nvs ? tid : nvs - 1;
            int lw, t, iwc, colok;
            locate(lv, lw, t, iwc, colok);
            const int base = ((iwc * g.Ti + t) * CIN + 24) * 4;
            s3_offb = (lw * TO + t) * 16 + 3 * PS;                         // (the entry of depth t holds input depths t - 1, t, t + 1)
            const int ok3[3] = {colok && t - 1 >= 0, colok, colok && t + 1 < g.Ti};
#pragma unroll
            for (int j = 0; j < 3; ++j) { s3_o[j] = base + (ok3[j] ? (j - 1) * CIN * 4 : 0); s3_sc[j] = ok3[j] ? sa : 0.f; }
        }
    }
    // the sample's input through a buffer descriptor: scalar row offset + the thread's constant byte offset, no 64-bit address arithmetic per load
    const long xsample = (long)g.Hi * g.Wi * g.Ti * CIN;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long)n * xsample), 0, (unsigned)xsample * 4u, 0x00020000);
    const int xrowb = g.Wi * g.Ti * CIN * 4;
    struct Staged { u32x4b v[2][2]; unsigned g3[3]; };
    auto row_of = [&](int q, int& rowoff, float& rokf) {                  // input row of ring row q (clamped), 1.0 / 0.0 = inside / outside the patch
        const int ih = hb - g.ph + q;
        const bool rok = ih >= 0 && ih < g.Hi;
        rowoff = (rok ? ih : 0) * xrowb;
        rokf = rok ? 1.f : 0.f;
    };
    auto stage_load_items = [&](int rowoff, Staged& sv) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sv.v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_srcb[k], rowoff, 0);
            sv.v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_srcb[k] + 16, rowoff, 0);
        }
    };
    auto stage_load_g3 = [&](int rowoff, Staged& sv) {
#ifdef CW4_ABL_NOG3             // (timing-only ablation: the gathered chunk's three 4-byte requests per item -- 64 cache lines per wave-instruction -- are not made)
        sv.g3[0] = sv.g3[1] = sv.g3[2] = 0u; return;
#endif
        if constexpr (CIN == 25) {
#pragma unroll
            for (int j = 0; j < 3; ++j) sv.g3[j] = __builtin_amdgcn_raw_buffer_load_b32(xrs, s3_o[j], rowoff, 0);
        }
    };
    auto slot_of = [&](int q) -> int { return (q - qdiv(q, NS, a.mNslot) * NS) * ROW; };
    // One row's cut and stores as a list of micro-operations -- about one instruction each, so that they can be dealt out over MFMA gaps:
    //   0..4 the row's scales; then per item k (18): per pair p the scaled pair, its first pieces (convert), its second pieces (two mixed fmas); two 16-byte stores;
    //   then the gathered chunk (11): three products, two converts, three mixed fmas, a zero, two 8-byte stores (words 2 and 3 of its entries stay zero from the clear)
    struct Cut { float sc[2], sg[3]; f32x2 t[4]; unsigned h0[4], h1[4]; float u[3]; };
    constexpr int ROW_MOPS = CIN == 25 ? 52 : 41;
    // (reload: the registers of an item are requested again, for the row the NEXT round cuts, as soon as its last value has been read -- one set of staging registers,
    //  a whole round between request and use; rel = that row's byte offset)
    auto stage_mop = [&](int m, int slot, float rokf, Staged& sv, Cut& c, bool reload, int rel) {
        if (m < 5) {
            if (m < 2) c.sc[m] = s_sc[m] * rokf;
            else if (CIN == 25) c.sg[m - 2] = s3_sc[m - 2] * rokf;
            return;
        }
        m -= 5;
        if (m < 36) {
            const int k = m / 18, i = m % 18;
            if (i < 16) {
                const int p = i >> 2, op = i & 3;
                if (op == 0) {
                    const u32x4b q = sv.v[k][p >> 1];
                    const f32x2 v = {__uint_as_float(q[2 * (p & 1)]), __uint_as_float(q[2 * (p & 1) + 1])};
                    c.t[p] = v * (f32x2){c.sc[k], c.sc[k]};
                    if (p == 3 && reload) {
                        sv.v[k][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_srcb[k], rel, 0);
                        sv.v[k][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_srcb[k] + 16, rel, 0);
                    }
                } else if (op == 1) {
                    c.h0[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(c.t[p], f16x2));
                } else if (op == 2) {
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(c.h1[p]) : "v"(c.h0[p]), "v"(c.t[p][0]));
                } else {
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c.h1[p]) : "v"(c.h0[p]), "v"(c.t[p][1]));
                }
            } else {
#ifdef CW4_ABL_NOWRITE
                asm volatile("" :: "v"(c.h0[0]), "v"(c.h0[1]), "v"(c.h0[2]), "v"(c.h0[3]), "v"(c.h1[0]), "v"(c.h1[1]), "v"(c.h1[2]), "v"(c.h1[3]));
                return;
#endif
                unsigned char* ent = lds + slot + s_offb[k];
                if (i == 16) *reinterpret_cast<uint4*>(ent) = make_uint4(c.h0[0], c.h0[1], c.h0[2], c.h0[3]);
                else *reinterpret_cast<uint4*>(ent + 4 * PS) = make_uint4(c.h1[0], c.h1[1], c.h1[2], c.h1[3]);
            }
            return;
        }
        if constexpr (CIN == 25) {
            m -= 36;
            if (m < 3) { c.u[m] = __uint_as_float(sv.g3[m]) * c.sg[m]; if (m == 2 && reload) stage_load_g3(rel, sv); }
            else if (m == 3) c.h0[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){c.u[0], c.u[1]}, f16x2));
            else if (m == 4) c.h0[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){c.u[2], 0.f}, f16x2));
            else if (m == 5) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(c.h1[0]) : "v"(c.h0[0]), "v"(c.u[0]));
            else if (m == 6) asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c.h1[0]) : "v"(c.h0[0]), "v"(c.u[1]));
            else if (m == 7) c.h1[1] = 0u;
            else if (m == 8) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(c.h1[1]) : "v"(c.h0[1]), "v"(c.u[2]));
            else if (m == 9) *reinterpret_cast<uint2*>(lds + slot + s3_offb) = make_uint2(c.h0[0], c.h0[1]);
            else *reinterpret_cast<uint2*>(lds + slot + s3_offb + 4 * PS) = make_uint2(c.h1[0], c.h1[1]);
        }
    };
    // last ring row (relative to the strip's first input row) that round r reads
    auto need = [&](int r) -> int {
        const int vlast = (r + 1) * 128 - 1 < NV - 1 ? (r + 1) * 128 - 1 : NV - 1;
        return qdiv(vlast, nvr, a.mNvr) + 2;
    };

    CW4_ST(3);
    // ---- prologue: every request first -- the rows of round 0 (at most four: nvr >= 64), the second pieces bound for LDS, the first pieces, the tables ----
    int hiq = nround > 0 ? need(0) : -1;
    Staged p0[4];
    int p0_slot[4];
    float p0_rok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int rowoff;
        row_of(q <= hiq ? q : 0, rowoff, p0_rok[q]);
        p0_slot[q] = slot_of(q <= hiq ? q : 0);
        stage_load_items(rowoff, p0[q]);
        stage_load_g3(rowoff, p0[q]);
    }
    constexpr int NQ = ((KB - NW1R) * 64 + 255) / 256;
    u32x4b w1q[NQ];
    const u32x4b* wfrag4 = reinterpret_cast<const u32x4b*>(wfrag);
    // The register-resident fragments (KB first pieces, NW1R second pieces) are the same for the four waves: four private copies are 180-216 KB through the CU's
    // one 64 B/clk vector L1 -- 7 700 cycles of request issue at kernel entry (tools/cw4diag.hip).  Each wave requests a QUARTER of them; they meet in LDS (the ring's
    // bytes, before the ring exists) and every wave reads all of them from there, straight into a[...].
    const int Z0c = a.Wp * (TP - 2);                                      // records per plane of a ring row (the zero zone starts there)
    constexpr int NF = KB + NW1R, NF4 = (NF + 3) / 4;
    u32x4b xq[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        int f = 4 * i + wave;
        f = f < NF ? f : NF - 1;                                         // (beyond the end: the last fragment again, the same bytes to the same place)
        xq[i] = wfrag4[(f < KB ? 2 * f : 2 * (f - KB) + 1) * 64 + lane];
    }
    // the output channels' exponents and biases: lane c of the first wave works out channel c's, two 32-entry tables behind the second pieces in LDS hand them to
    // every lane in the accumulator's layout (register 4 jj + i of a lane = channel 8 jj + 4 half + i: four 16-byte reads per table)
    const int tab = NS * ROW + (KB - NW1R) * 1024;
    const int tc = tid < g.Cout ? tid : 0;
    const unsigned tew = am.w[tc];
    const float tbv = (bias ? bias : reinterpret_cast<const float*>(am.w))[tc];
    // (the LDS-bound second pieces are requested LAST: every CU of the launch is in its prologue at once and gets ~11 bytes per cycle -- the first bytes should be the ones the exchange waits for)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int e = tid + 256 * i;
        e = e < (KB - NW1R) * 64 ? e : (KB - NW1R) * 64 - 1;               // (beyond the end: the last element again)
        w1q[i] = wfrag4[(2 * (NW1R + (e >> 6)) + 1) * 64 + (e & 63)];
    }
    CW4_ST(4);
    if (tid < 32) {
        const int e = -(ea + h3_exp_w(tew));
        reinterpret_cast<int*>(lds + tab)[tid] = e;
        reinterpret_cast<float*>(lds + tab + 128)[tid] = (BIAS && bias && tid < g.Cout) ? ldexpf(tbv, -e) : 0.f;      // the bias at the accumulator's scale (a power-of-two factor, exact)
    }
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        int f = 4 * i + wave;
        f = f < NF ? f : NF - 1;
        *reinterpret_cast<u32x4b*>(lds + f * 1024 + lane * 16) = xq[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (LDS only: the other requests stay in flight)
    f16x8 w0[KB], w1r[NW1R > 0 ? NW1R : 1];
#pragma unroll
    for (int K = 0; K < KB; ++K) { Frag f; f.u = *reinterpret_cast<const uint4*>(lds + K * 1024 + lane * 16); w0[K] = f.h; }
#pragma unroll
    for (int K = 0; K < NW1R; ++K) { Frag f; f.u = *reinterpret_cast<const uint4*>(lds + (KB + K) * 1024 + lane * 16); w1r[K] = f.h; }
    // pin them in the accumulator half of the file (an MFMA reads its A operand from there): from here on the allocator holds them in a[...]
#pragma unroll
    for (int K = 0; K < KB; ++K) asm volatile("" : "+a"(w0[K]));
#pragma unroll
    for (int K = 0; K < NW1R; ++K) asm volatile("" : "+a"(w1r[K]));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave has its copy: the bytes become the ring
    // What of the ring must be zero: the zero zones behind the records of every plane, the upper halves of the gathered planes' entries (their stores are 8 bytes) --
    // and, where the row is not cut into column ranges, the halo columns nobody stages: then all of it.
    if (a.nsplit == 1) {
        uint4* z = reinterpret_cast<uint4*>(lds);
        for (int i = tid; i < NS * ROW / 16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        const int nzone = CW4_PSE - Z0c, nplanes = NS * 8;
        for (int i = tid; i < nzone * nplanes; i += 256) {
            const int pl = i / nzone, e = i - pl * nzone;
            *reinterpret_cast<uint4*>(lds + pl * PS + (Z0c + e) * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
        if constexpr (CIN == 25) {
            for (int i = tid; i < Z0c * NS * 2; i += 256) {
                const int pl = i / Z0c, e = i - pl * Z0c;                  // pl = 2 slot + piece
                *reinterpret_cast<uint4*>(lds + (pl >> 1) * ROW + (3 + 4 * (pl & 1)) * PS + e * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    CW4_ST(5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                         // (a row beyond the round's need repeats row 0)
        Cut c;
#pragma unroll
        for (int m = 0; m < ROW_MOPS; ++m) stage_mop(m, p0_slot[q], p0_rok[q], p0[q], c, false, 0);
    }
    CW4_ST(6);
    const int w1a = NS * ROW + lane * 16;                                // LDS address of the lane's 16 bytes of second-piece fragment NW1R
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int e = tid + 256 * i;
        e = e < (KB - NW1R) * 64 ? e : (KB - NW1R) * 64 - 1;
        *reinterpret_cast<u32x4b*>(lds + NS * ROW + e * 16) = w1q[i];
    }
    int eun[16];
    // The bias enters as the accumulator's initial value, at the accumulator's scale -- the C operand of a tile's first MFMA, kept in the accumulator half of the
    // file like the filter: sixteen vector registers and sixteen additions per tile less.
    f32x16 bias16;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int4 e4 = *reinterpret_cast<const int4*>(lds + tab + (8 * jj + 4 * half) * 4);
        const float4 b4 = *reinterpret_cast<const float4*>(lds + tab + 128 + (8 * jj + 4 * half) * 4);
        eun[4 * jj] = e4.x; eun[4 * jj + 1] = e4.y; eun[4 * jj + 2] = e4.z; eun[4 * jj + 3] = e4.w;
        bias16[4 * jj] = b4.x; bias16[4 * jj + 1] = b4.y; bias16[4 * jj + 2] = b4.z; bias16[4 * jj + 3] = b4.w;
    }
    const float lo = g.relu ? 0.f : -__builtin_inff();                    // (RELU instances: the layer's flag decides at run time)
    (void)lo;
    if (BIAS) asm volatile("" : "+a"(bias16));
    CW4_ST(7);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- output / skip through buffer descriptors of the strip: an invalid voxel's offset lies beyond num_records (no branch around a load or a store) ----
    const long out_base = ((long)n * g.Ho + hb) * g.Wo * g.To;
    const unsigned ybytes = (unsigned)(SRr * g.Wo * g.To * g.Cout) * 4u;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y + out_base * g.Cout, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(skip ? skip + out_base * g.Cout : y), 0, skip ? ybytes : 0u, 0x00020000);
    unsigned cofs[4];                                                     // byte offset of the lane's channel group inside a voxel; beyond every num_records for a group the layer does not have
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) cofs[jj] = 8 * jj + 4 * half < COUT ? 4u * (8 * jj + 4 * half) : CW4_OOB;

    // Ring rows are DENSE: entry x = column * TO + depth of each of the eight planes, no depth pads -- consecutive voxels of a tile are consecutive entries across
    // column changes, so the sixteen lanes of a ds_read_b128 group (MI355X_MICROARCH.md, LDS) read sixteen consecutive 16-byte units: no bank conflict.  (With padded
    // columns, entry = column * (TO + 2) + depth, the entry jumps by 3 at a column change and a tile's 32 lanes cost ten extra LDS cycles per half: rocprofv3 showed
    // SQ_LDS_BANK_CONFLICT at 55 % of SQ_LDS_IDX_ACTIVE and the LDS 67-80 % busy.)  A tap is still an immediate offset -- dw: TO entries, depth step: one entry -- from
    // one of THREE bases per ring row: depth step -1, 0, +1; the base of a step that leaves the patch (depth 0 stepping down, depth TO - 1 stepping up) points into a
    // zone of zero entries behind the row's records (entries Wp * TO .. of every plane: cleared once, never written), wide enough for every immediate from
    // any of sixteen consecutive entries (tile_step picks the one in the lane's own bank).
    // 25 channels: chunk c < 9 of a (dh, dw) group = plane c % 3 at depth step c / 3 - 1, c = 9 = the gathered plane; lanes 0-31 read chunk 2 st, lanes 32-63 chunk
    // 2 st + 1 of k-block st: per-lane plane offsets here, the depth step picks the base.  32 channels: the lane half's plane is part of the bases.
    int offH[5];
    {
        const int pA[5] = {0, 2, 1, 0, 2}, pB[5] = {1, 0, 2, 1, 3};
#pragma unroll
        for (int st = 0; st < 5; ++st) offH[st] = (half ? pB[st] : pA[st]) * PS;
    }
    const int Z0 = a.Wp * TO;                                            // first zero entry of a plane
    // A tile's lane constants -- ring addresses of the lane's voxel at (dw, dt) = (0, 0) in rows hrel + dh; byte offsets of its four channel groups in the strip's
    // output -- in ten steps of a few instructions each (dealt out over MFMA gaps inside the rounds)
    struct TileTmp { int vi, hrel, rem, w, t, x, e0, s0, s1, s2, b0, b1, b2, dlo, dhi; unsigned yo; };
    constexpr int TILE_STEPS = 14;
    auto tile_step = [&](int step, int tile, TileTmp& q, int (&bs)[9], unsigned (&eo)[4]) {
        // (the empty asm statements keep a step's instructions in the step: without them the optimiser sinks the arithmetic to its first use, the end of the round)
        // A depth step that leaves the patch reads the zero entry whose index is congruent (mod 16) to the entry the step would have read: the lane keeps the bank
        // nobody else of its ds_read_b128 group has.
#define CW4_PIN(v) asm volatile("" : "+v"(v))
        if (step == 0) { const int v = tile * 32 + col; q.vi = v < NV ? v : NV - 1; q.yo = v < NV ? 0u : CW4_OOB; CW4_PIN(q.vi); CW4_PIN(q.yo); }
        if (step == 1) { q.hrel = (int)__umulhi((unsigned)q.vi, a.mNvr); q.rem = q.vi - q.hrel * nvr; CW4_PIN(q.hrel); CW4_PIN(q.rem); }
        if (step == 2) { q.w = (int)__umulhi((unsigned)q.rem, a.mTo); q.t = q.rem - q.w * g.To; CW4_PIN(q.w); CW4_PIN(q.t); }
        if (step == 3) { q.x = q.w * TO + q.t; q.s0 = q.hrel - (int)__umulhi((unsigned)q.hrel, a.mNslot) * NS; CW4_PIN(q.x); CW4_PIN(q.s0); }
        if (step == 4) { q.s1 = q.s0 + 1 < NS ? q.s0 + 1 : q.s0 + 1 - NS; q.e0 = q.x * 16 + (CIN == 32 ? half * PS : 0); CW4_PIN(q.s1); CW4_PIN(q.e0); }
        if (step == 5) { q.s2 = q.s1 + 1 < NS ? q.s1 + 1 : q.s1 + 1 - NS; CW4_PIN(q.s2); }
        if (step == 6) { q.b0 = q.s0 * ROW + q.e0; q.b1 = q.s1 * ROW + q.e0; q.b2 = q.s2 * ROW + q.e0; bs[1] = q.b0; bs[4] = q.b1; bs[7] = q.b2; CW4_PIN(bs[1]); CW4_PIN(bs[4]); CW4_PIN(bs[7]); }
        if (step == 7) { const int z = ((q.x - 1 - Z0) & 15) + Z0 - q.x; q.dlo = q.t == 0 ? z * 16 : -16; CW4_PIN(q.dlo); }
        if (step == 8) { const int z = ((q.x + 1 - Z0) & 15) + Z0 - q.x; q.dhi = q.t == TO - 1 ? z * 16 : 16; CW4_PIN(q.dhi); }
        if (step == 9) { bs[0] = q.b0 + q.dlo; bs[3] = q.b1 + q.dlo; bs[6] = q.b2 + q.dlo; CW4_PIN(bs[0]); CW4_PIN(bs[3]); CW4_PIN(bs[6]); }
        if (step == 10) { bs[2] = q.b0 + q.dhi; bs[5] = q.b1 + q.dhi; bs[8] = q.b2 + q.dhi; CW4_PIN(bs[2]); CW4_PIN(bs[5]); CW4_PIN(bs[8]); }
        if (step == 11) { q.yo += (unsigned)(((q.hrel * g.Wo + ws0 + q.w) * g.To + q.t) * COUT) * 4u; CW4_PIN(q.yo); }      // (an invalid voxel: beyond num_records with or without the sum)
        if (step == 12) { eo[0] = q.yo + cofs[0]; eo[1] = q.yo + cofs[1]; CW4_PIN(eo[0]); CW4_PIN(eo[1]); }
        if (step == 13) { eo[2] = q.yo + cofs[2]; eo[3] = q.yo + cofs[3]; CW4_PIN(eo[2]); CW4_PIN(eo[3]); }
#undef CW4_PIN
    };
    int adr5[5] = {0, 0, 0, 0, 0};
    // operand reads of k-block K of a tile with bases bs
    auto read_A = [&](const int (&bs)[9], int K, Frag (&af)[2]) {
        const int gg = K / NST, st = K % NST, dh = gg / 3, dw = gg % 3;
        if constexpr (CIN == 25) {
            if (dw == 0) {                                               // (one sum per (dh, st): the three dw of a row share it)
                const int bm = bs[3 * dh], bz = bs[3 * dh + 1], bp = bs[3 * dh + 2];      // (values first: `half ? bs[i] : bs[j]` is a conditional ADDRESS to the compiler, and bs[] then lives in scratch)
                const int b = st == 0 ? bm : st == 1 ? (half ? bz : bm) : st == 2 ? bz : st == 3 ? bp : (half ? bz : bp);
                adr5[st] = b + offH[st];
            }
            const unsigned char* p = lds + adr5[st];
            af[0].u = *reinterpret_cast<const uint4*>(p + dw * TO * 16);
            af[1].u = *reinterpret_cast<const uint4*>(p + dw * TO * 16 + 4 * PS);
        } else {
            const int d = st >> 1, kb = st & 1;
            const unsigned char* p = lds + bs[3 * dh + d];
            af[0].u = *reinterpret_cast<const uint4*>(p + dw * TO * 16 + 2 * kb * PS);
            af[1].u = *reinterpret_cast<const uint4*>(p + dw * TO * 16 + 2 * kb * PS + 4 * PS);
        }
    };
    auto read_W1 = [&](int K, Frag& wf) {
        if (K >= NW1R) wf.u = *reinterpret_cast<const uint4*>(lds + w1a + (K - NW1R) * 1024);
    };

    CW4_ST(2);                                                           // (diagnostic build: 3 constants, 4 requests issued, 5 ring cleared + barrier, 6 first rows cut, 7 second pieces stored + tables, 2 the rest of the prologue)
    // ---- the rounds ----
    f32x16 accA, accB;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accA[i] = 0.f; accB[i] = 0.f; }
#ifdef CW4_ABL_2ACC
    f32x16 acc2;
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#endif
    f32x4u skq[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) skq[jj] = (f32x4u){0.f, 0.f, 0.f, 0.f};
    int bs[9];
    unsigned eo[4], eo_prev[4] = {CW4_OOB, CW4_OOB, CW4_OOB, CW4_OOB};
    {
        TileTmp q;
#pragma unroll
        for (int st = 0; st < TILE_STEPS; ++st) tile_step(st, wave, q, bs, eo);
    }
    Frag A[PD + 1][2], W1[PD + 1];
#pragma unroll
    for (int K = 0; K < PD; ++K) { read_A(bs, K, A[K]); read_W1(K, W1[K]); }

    Staged sv[2];                                                         // the two rows the current round cuts (requested a round earlier)
    // One round: the tile's MFMAs into `acc`.  In their gaps a few instructions each (a single wave issues one instruction per 4-5 cycles, whatever its kind, and an
    // MFMA leaves 24 of its 32 cycles; the operand requests take their share): the epilogue of the previous tile (`pacc`, skq, eo_prev), the skip loads of this
    // tile, the next tile's constants, and the staging of TWO ring rows for round r + 1 -- a round needs one or two new rows; when it needs one the second is that row
    // again (the same bytes twice: no branch -- skipping the second row's micro-operations behind one uniform branch per gap was measured: the branches cost what the
    // skipped cuts save).  The rows' values were requested a whole round earlier, into the registers the cut of that round had just read.
    // Placement, by k-block K and MFMA j of it: epilogue value e behind (K = e, j = 2), the group stores behind (4, 8, 12, 16; 1), tile constants (1..10; 1), the
    // barrier (11; 1), skip loads (17, 18; 1, 2), cuts and stores (with the next round's requests among them) three behind each of (KS0..; 1, 2).
    auto round = [&](int r, f32x16& acc, f32x16& pacc) __attribute__((always_inline)) {
        const int hi_next = r + 1 < nround ? need(r + 1) : hiq;
        const int hi_nn = r + 2 < nround ? need(r + 2) : hi_next;
        const int q0 = hiq + 1, q1 = hi_next - hiq >= 2 ? hiq + 2 : hiq + 1;                   // the rows cut and stored now
        const int n0 = hi_next + 1, n1 = hi_nn - hi_next >= 2 ? hi_next + 2 : hi_next + 1;     // the rows requested now
        int rowoff[2], slot[2];
        float rokf[2], rokn;
        int dummy;
        row_of(q0, dummy, rokf[0]); row_of(q1, dummy, rokf[1]);
        row_of(n0, rowoff[0], rokn); row_of(n1, rowoff[1], rokn);
        slot[0] = slot_of(q0); slot[1] = slot_of(q1);
        Cut cut;
        TileTmp tq;
        int nbs[9];
        unsigned neo[4];
        float m2 = 0.f;
#pragma unroll
        for (int K = 0; K < KB; ++K) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int gp = 3 * K + j;                                 // the gap behind this MFMA
                const int sl = K % (PD + 1);
                // ---- the MFMA ----
                {
                    f16x8 wa;
                    if (j == 1) { if (K < NW1R) wa = w1r[K < NW1R ? K : 0]; else wa = W1[sl].h; }
                    else wa = w0[K];
                    const f16x8 xb = (j == 0) ? A[sl][1].h : A[sl][0].h;
                    if (gp == 0) {
                        if (BIAS) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, bias16, 0, 0, 0);
                        else {
                            f32x16 zero;
#pragma unroll
                            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, zero, 0, 0, 0);
                        }
                    } else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, acc, 0, 0, 0);
                }
                CW4_SBAR();
                // ---- operand requests PD k-blocks ahead (the last PD of a tile: the next tile's first) ----
#ifndef CW4_ABL_NOREAD          // (timing-only ablation of the diagnostic build: the operand registers are never reloaded)
                // (the filter's second piece FIRST: LDS reads return in order, so the wait for the k-block's A operand covers it -- one s_waitcnt per k-block instead of two:
                //  244 -> 178 in the forward instance, -1.2 % / -1.6 % of the kernel's cycles forward / backward-data; CW4_ABL_W1LATE restores the old place in diagnostic builds)
                if (j == 0) {
                    const int KN = K + PD, sn = KN % (PD + 1);
#ifndef CW4_ABL_W1LATE
                    read_W1(KN < KB ? KN : KN - KB, W1[sn]);
#endif
                    if (KN < KB) read_A(bs, KN, A[sn]); else read_A(nbs, KN - KB, A[sn]);
                }
#ifdef CW4_ABL_W1LATE
                if (j == 1) {
                    const int KN = K + PD, sn = KN % (PD + 1);
                    read_W1(KN < KB ? KN : KN - KB, W1[sn]);
                }
#endif
#endif
                // ---- the fillers ----
                // the next tile's constants (its first operand requests follow in the last PD k-blocks)
                if (K >= 1 && K <= 10 && j == 1) tile_step(K - 1, 4 * (r + 1) + wave, tq, nbs, neo);
                if (K >= 12 && K <= 15 && j == 1) tile_step(K - 2, 4 * (r + 1) + wave, tq, nbs, neo);
                // The round's barrier.  Behind it the rows staged in round r - 1 may be read (the k-blocks requested so far are dh = 0 ones: rows the previous round's
                // readers had already), and the slots of rows below this round's first may be overwritten.
                if (K == 11 && j == 1) { CW4_STR(3); asm volatile("s_barrier" ::: "memory"); CW4_STR(4); }
#ifndef CW4_ABL_NOEPI            // (timing-only ablations: no epilogue / no staging)
                // the previous tile's epilogue, one value per k-block; a group's 16-byte store behind its fourth value
                if (K < 16 && j == 2) {
                    const int e = K, jj = e >> 2, i = e & 3;
                    float v = ldexpf(pacc[e], eun[e]);
                    if (RELU) v = fmaxf(v, lo);
                    const float o = SKIP ? v + skq[jj][i] : v;
                    skq[jj][i] = o;
                    if ((e & 1) == 0) m2 = o; else asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(omax) : "v"(m2), "v"(o));
                }
                if (j == 1 && (K == 4 || K == 8 || K == 12 || K == 16)) {
                    const int jj = K / 4 - 1;
                    const u32x4b o = {__float_as_uint(skq[jj][0]), __float_as_uint(skq[jj][1]), __float_as_uint(skq[jj][2]), __float_as_uint(skq[jj][3])};
                    if (COUT == 32 || jj < 3) __builtin_amdgcn_raw_buffer_store_b128(o, yrs, eo_prev[jj], 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(o[0], yrs, eo_prev[jj], 0, 0);      // 25 output channels: the last group is channel 24 alone
                    if (K == 16) omax = r == 0 ? 0.f : omax;              // (round 0 has no previous tile: what its epilogue gaps computed is dropped)
                }
                // this tile's skip values (consumed by the next round's epilogue)
                if (SKIP && (K == 17 || K == 18) && j > 0) {
                    const int jj = 2 * (K - 17) + j - 1;
                    if (COUT == 32 || jj < 3) {
                        const u32x4b q = __builtin_amdgcn_raw_buffer_load_b128(srs, eo[jj], 0, 0);
                        skq[jj] = (f32x4u){__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
                    } else {
                        const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(srs, eo[jj], 0, 0);
                        skq[jj] = (f32x4u){__uint_as_float(q), 0.f, 0.f, 0.f};
                    }
                }
#endif
#ifndef CW4_ABL_NOSTAGE
                // cuts and stores of the two rows requested a round ago, six micro-operations per k-block
                {
                    constexpr int NKS = (2 * ROW_MOPS + 5) / 6, KS0 = KB - PD - NKS;
                    if (K >= KS0 && K < KS0 + NKS && j > 0) {
                        const int m0 = (K - KS0) * 6 + 3 * (j - 1);
#pragma unroll
                        for (int m = m0; m < m0 + 3; ++m)
                            if (m < 2 * ROW_MOPS) {
#ifdef CW4_ABL_NOCUT            // (timing-only ablations: the requests alone / cuts without their LDS stores)
                                if (m == 0) asm volatile("" :: "v"(sv[0].v[0][0]), "v"(sv[0].v[1][1]), "v"(sv[1].v[0][0]), "v"(sv[1].v[1][1]));
                                continue;
#endif
                                stage_mop(m % ROW_MOPS, slot[m / ROW_MOPS], rokf[m / ROW_MOPS], sv[m / ROW_MOPS], cut, true, rowoff[m / ROW_MOPS]);
                            }
                    }
                }
#endif
                CW4_SBAR();
            }
        }
        CW4_STR(5);
#pragma unroll
        for (int d = 0; d < 9; ++d) bs[d] = nbs[d];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { eo_prev[jj] = eo[jj]; eo[jj] = neo[jj]; }
        hiq = hi_next;
    };
    {   // the rows round 0 cuts (requested here; the prologue's own rows are in the ring already)
        const int hi1 = 1 < nround ? need(1) : hiq;
        const int n0 = hiq + 1, n1 = hi1 - hiq >= 2 ? hiq + 2 : hiq + 1;
        int ro;
        float rk;
        row_of(n0, ro, rk); stage_load_items(ro, sv[0]); stage_load_g3(ro, sv[0]);
        row_of(n1, ro, rk); stage_load_items(ro, sv[1]); stage_load_g3(ro, sv[1]);
    }
    // (The strip's 69th tile -- two voxels -- costs a whole round in which three SIMDs idle.  Splitting its k-blocks over the four waves was built and measured: -3 % per
    // launch.  It is not here because its four partial chains round differently from the one chain every other tile sums in, and which voxels sit in a lone tile depends
    // on how the batch size cuts the strips: a sample's result would depend on its batch in the last bits -- tests/test_gpu_parity.py, test_full_size_batch128_properties.)
    // (rounds in pairs, the second one unconditional -- an odd count runs one round on tiles beyond the strip, whose stores are dropped: a conditional second
    // round is its own basic block, and the optimiser sinks the first round's requests for it into that block)
#pragma unroll 1
    for (int r = 0; r < nround; r += 2) {
        round(r, accA, accB);
        round(r + 1, accB, accA);
    }
    // ---- the last tile's epilogue ----
    {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int jj = e >> 2, i = e & 3;
            float v = ldexpf(accB[e], eun[e]);                             // (the second round of a pair accumulates into accB)
            if (RELU) v = fmaxf(v, lo);
            const float o = SKIP ? v + skq[jj][i] : v;
            skq[jj][i] = o;
            omax = fmaxf(omax, fabsf(o));
        }
        if (nround == 0) omax = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const u32x4b o = {__float_as_uint(skq[jj][0]), __float_as_uint(skq[jj][1]), __float_as_uint(skq[jj][2]), __float_as_uint(skq[jj][3])};
            if (COUT == 32 || jj < 3) __builtin_amdgcn_raw_buffer_store_b128(o, yrs, eo_prev[jj], 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(o[0], yrs, eo_prev[jj], 0, 0);
        }
    }
#ifdef CW4_ABL_2ACC
    for (int i = 0; i < 16; ++i) omax += acc2[i];
#endif
    if (am.y) amax_commit(omax, am.y + n);
#ifdef CW4_STAMP
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_cw4_stamps + (blockIdx.x * 4 + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        for (int k = 2; k < 8; ++k) o[k] = st_acc[k];
    }
#endif
}

// ---- host side ----
static bool cw4_plan(const ConvGeom& g, Cw4Args& p, size_t& lds_bytes, int& grid)
{
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.reflect_hw || g.reflect_t) return false;
    if (!((g.Cin == 25 && g.Cout == 32) || (g.Cin == 32 && (g.Cout == 25 || g.Cout == 32)))) return false;
    if (g.ph != 1 || g.pw != 1 || g.pt != 1) return false;                                    // 'same' padding (normConv and its backward-data)
    if (g.Ho != g.Hi || g.Wo != g.Wi || g.To != g.Ti) return false;
    if (g.To != 9 && g.To != 7) return false;                                                  // the instantiated depths (TP = To + 2)
    if (g.Ho < 3) return false;
    const int Tp = g.To + 2;
    for (int ns = 1; ns <= 4; ++ns) {
        if (g.Wo % ns) continue;
        const int wt = g.Wo / ns, nvr = wt * g.To;
        const int wp = wt + 2;
        const int nvs = (ns == 1 ? g.Wi : wt + 2) * g.Ti, items = nvs * (g.Cin == 25 ? 3 : 4);
        if (items > 512 || nvs > 256 || nvr < 64 || wp * g.To + 16 + 2 * g.To > CW4_PSE) continue;
        int nstrips = (256 + g.N * ns - 1) / (g.N * ns);
        if (nstrips < 1) nstrips = 1;
        if (nstrips > g.Ho / 4) nstrips = g.Ho / 4 > 0 ? g.Ho / 4 : 1;
        const int SR = (g.Ho + nstrips - 1) / nstrips;
        nstrips = (g.Ho + SR - 1) / SR;
        // ring depth: while round r's taps read rows lo(r) .. need(r), the rows up to need(r + 1) are written
        const int NV = SR * nvr, nround = ((NV + 31) / 32 + 3) / 4;
        auto needf = [&](int r) { const int vl = std::min(NV - 1, (r + 1) * 128 - 1); return vl / nvr + 2; };
        int nslot = needf(0) + 1;
        for (int r = 0; r + 1 < nround; ++r) nslot = std::max(nslot, needf(r + 1) - (r * 128) / nvr + 1);
        const int kb = g.Cin == 25 ? 45 : 54, nw1r = g.Cin == 25 ? 12 : 8;
        const size_t need = (size_t)nslot * CW4_ROW + (size_t)(kb - nw1r) * 1024 + 256;      // ring, second pieces, the two channel tables
        if (need > 163840) continue;
        p.g = g; p.Wp = wp; p.Wt = wt; p.nsplit = ns; p.SR = SR; p.nstrips = nstrips; p.nslot = nslot;
        p.mTo = qmagic(g.To); p.mNvr = qmagic(nvr); p.mTi = qmagic(g.Ti); p.mNslot = qmagic(nslot);
        lds_bytes = need; grid = g.N * nstrips * ns;
        return true;
    }
    return false;
}

#ifndef CW4_DIAG
static int g_cw4_enabled = -1;
bool cw4_enabled()
{
    if (g_cw4_enabled < 0) { const char* e = getenv("PROBAV_GEN1"); g_cw4_enabled = !(e && (e[0] == '1' || e[0] == 'c')); }      // PROBAV_GEN1 = 1 (every general form) | conv | pw | pwf | pwb
    return g_cw4_enabled != 0;
}
void cw4_set_enabled(int on) { g_cw4_enabled = on ? 1 : 0; }
#endif

bool cw4_conv_supported(const ConvGeom& g, const float* gate)
{
    Cw4Args p;
    size_t lds_bytes;
    int grid;
    return gate == nullptr && cw4_plan(g, p, lds_bytes, grid);
}

int cw4_conv_forward(const ConvGeom& g, const float* x, const float* wfrag, const float* bias, const float* skip, float* y, const Amax& am, hipStream_t s)
{
    Cw4Args p;
    size_t lds_bytes;
    int grid;
    if (!cw4_plan(g, p, lds_bytes, grid)) { set_error("cw4_conv_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    if (!am.x || !am.w) { set_error("cw4_conv_forward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] {
    // instances: every (channels, depth) with all three of ReLU / skip / bias compiled in (a layer without one passes 0 / no tensor), and the two layers of the
    // residual blocks without what they do not have: normConv forward (skip + bias, no ReLU), its backward-data (none of the three)
#define CW4_BIG(C, O, T, R, S, B) (void)hipFuncSetAttribute((const void*)conv3_w4_kernel<C, O, T, R, S, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
        CW4_BIG(25, 32, 11, true, true, true); CW4_BIG(32, 25, 11, true, true, true); CW4_BIG(32, 32, 11, true, true, true);
        CW4_BIG(25, 32, 9, true, true, true); CW4_BIG(32, 25, 9, true, true, true); CW4_BIG(32, 32, 9, true, true, true);
        CW4_BIG(25, 32, 11, false, true, true); CW4_BIG(32, 25, 11, false, false, false); });
#undef CW4_BIG
#define CW4_LAUNCH(C, O, T, R, S, B) hipLaunchKernelGGL((conv3_w4_kernel<C, O, T, R, S, B>), dim3(grid), dim3(256), lds_bytes, s, p, x, (const uint4*)wfrag, bias, skip, y, am)
#define CW4_LAUNCH_T(C, O) do { if (g.To == 9) CW4_LAUNCH(C, O, 11, true, true, true); else CW4_LAUNCH(C, O, 9, true, true, true); } while (0)
    if (g.Cin == 25 && g.To == 9 && !g.relu && skip && bias) CW4_LAUNCH(25, 32, 11, false, true, true);
    else if (g.Cin == 32 && g.Cout == 25 && g.To == 9 && !g.relu && !skip && !bias) CW4_LAUNCH(32, 25, 11, false, false, false);
    else if (g.Cin == 25) CW4_LAUNCH_T(25, 32);
    else if (g.Cout == 25) CW4_LAUNCH_T(32, 25);
    else CW4_LAUNCH_T(32, 32);
#undef CW4_LAUNCH_T
#undef CW4_LAUNCH
    return check_launch("conv3_w4");
}

#ifdef CW4_DIAG
}  // namespace diag
#endif
}  // namespace probav
