// Backward-filter of the 3x3x3 convolutions of the residual blocks (normConv: 25 -> 32 channels) and of the reducers (32 -> 32), H3 arithmetic -- ONE WAVE PER SIMD (round 5).
// Reference semantics: tape.gradient (models/trainClass.py:131) through models/modelsTF.py:185-186 (normConv_i of ResConv3D) and :123-150 (convReducer_i behind tf.pad REFLECT):
//   dW[tap][ci][co] = sum over the voxels v of all samples of X[v + tap][ci] dY[v][co],   db[co] = sum_v dY[v][co].
//
// What conv3_wgrad_x6_kernel (kernels_x6.hip: the general form -- any extent, pads, depth 13, batches above 256) measured in the step: 108 us per launch, matrix pipe 37 %
// busy, 47 % of its LDS cycles bank conflicts, two waves per SIMD that each wait for the four transposed reads of the NEXT tap tile only.  The contraction has no operand
// reuse to speak of (N = 32 output channels is all there is: every (tap, channel, voxel) element of the A operand feeds ONE MFMA triple), so the A operand's LDS
// reads are the kernel, and their latency and their conflicts were what it waited for.  This kernel keeps the products (per k-block x1 d0, x0 d1, x0 d0; one scale
// per operand tensor -- since the end of round 5 with LIFTED second pieces and the two cross products in an accumulator of their own: WG4_LIFT below) and changes the shape:
//   * ONE WAVE PER SIMD, ONE INSTRUCTION STREAM per output row: the M tiles (25 channels: the taps' 25 + 3 rows packed into 24 tiles, six per wave; 32 channels: 27 tap tiles
//     dealt 7 / 7 / 7 / 6) stay with their wave for ALL k-blocks (no parity exchange at the end); a row = NKB k-blocks x 6 | 7 tiles x 3 MFMAs, fully unrolled, the transposed reads
//     two tiles ahead in a register ring; the accumulators live in a[...] and are read once, at the kernel's end.
//   * 64-BYTE PIECE PLANES: the input ring is [slot][piece][column][depth + 2][32 channels] fp16 -- the four voxel rows x eight channel quads of a 32-lane half of a transposed
//     read are 256 consecutive bytes (all 64 banks once) unless they straddle a tap (the packed 25-channel tiles always do: two-way conflicts, and still 9 % faster than 27
//     conflict-free tiles with 14 % more MFMAs: -DWG4_RT=32) or a depth wrap.  (The general form interleaves the pieces at 112 bytes per voxel: every read two-way conflicted.)
//   * dY IS CUT ONCE PER WORKGROUP: a k-block's two B fragments are a UNIT, a quarter per wave (two 4-byte requests through the row's buffer descriptor -- voxels behind the row's
//     end read as zero --, four mixed fmas, two bias additions, two 4-byte LDS stores per lane), requested LA k-blocks ahead of its cut by a compile-time schedule; every wave reads a
//     k-block's fragments with two ds_read_b128.  (Cut by every wave for itself it was a third of all instructions between the MFMAs.)
//   * FOUR RING SLOTS, TWO BARRIERS PER ROW: while row r reads input rows r - 1 .. r + 1, row r + 2 is cut and stored into the fourth slot (its values were requested a whole
//     row earlier, into the registers the previous cut had just read); one fragment buffer serves the row being read and the row being cut.  A wave arrives at a barrier with
//     s_waitcnt lgkmcnt(n), n = the LDS operations it has issued behind its last store (compile time): the reads in flight stay in flight.
// Workgroup = (sample, strip of rows); one slab per workgroup, summed by mfma_wgrad_reduce (fp64, fixed order) like the general form's.  Instances: normConv (25 -> 32 channels,
// 'same' zero padding, depth 9 | 7) and the reducers (32 -> 32, tf.pad(REFLECT) rows / columns, no depth pads, output depth 7 | 5 | 3, dY masked by the layer's output).
#include "kernels_x6.h"
#include "x6_device.h"
#include <cstdlib>
#include <mutex>

// The kernel's ten instances are one unrolled instruction stream each and 6 min 40 s of hipcc in one translation unit: kernels_wg4b.hip / kernels_wg4c.hip include THIS file with
// WG4_PART = 1 / 2 and instantiate their share of them (WG4_PART1 / WG4_PART2 below); part 0 -- this file compiled for itself -- declares those `extern template`, instantiates the
// residual blocks' three by use and holds the host side.  (tools/wg4*.hip include the file as probav::diag: everything in one unit.)
#ifndef WG4_PART
#define WG4_PART 0
#endif

namespace probav {
#ifdef WG4_DIAG                 // tools/wg4bench.hip includes this file as probav::diag (stamped / ablated builds beside the product's copy in the library)
namespace diag {
#endif

struct Wg4Args { int N, H, SR, nstrips, nsplit; };

// LIFTED SECOND PIECES (round 5).  The contraction runs over the voxels of all samples, so an operand takes ONE scale per tensor, and fp16's five exponent bits
// used to bound what a channel far below its tensor mates kept: a value's second piece rn16(v s - h0) is 2^-11 of the first and left the normal range 18 binades below the
// tensor's maximum (a channel at 2^-24 of it got a 1e-2 gradient slice).  Here the second piece is stored LIFTED, h1' = rn16((v s - h0) 2^11) -- v s - h0 is exact in fp32,
// |h1'| <= |h0| -- so both pieces of a value are normal 29 binades below the maximum, and the two cross products x1' d0 + x0 d1' (2^11 too large, both) go to an accumulator
// of their own: dW = (sum x0 d0 + 2^-11 sum (x1' d0 + x0 d1')) 2^-(ex + ed).  Same three MFMAs per k-block and tile; one more vector instruction per cut value; 96 / 112 more
// accumulator registers (a[...]: the kernel has them).  -DWG4_LIFT=0: the plain second pieces and one accumulator (rounds 2 - 4; for A/B builds).
#ifndef WG4_LIFT
#define WG4_LIFT 1
#endif
constexpr int WG4_IM = WG4_LIFT ? 16 : 12;      // micro-operations of one staging item (stage_mop); the last two are its LDS stores

namespace {
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
}

#ifdef WG4_STAMP
__device__ unsigned long long g_wg4_stamps[1024 * 8];
#endif

// ---- the compile-time schedule of a row's k-blocks (conv3_wgrad_w4_kernel: "the rows") ----
// k-block in which unit u (the dY fragments of k-block u) is cut and stored: units MB + 1 .. of a row in that row's k-blocks 0 .., units 0 .. MB of the NEXT row in k-blocks MB .. NKB - 2
template <int NKB, int MB>
__host__ __device__ constexpr int wg4_cutK(int u)
{
    constexpr int nB = MB + 1, kB = NKB - 1 - MB, extra = nB - kB;
    if (u <= MB && kB == 1) return MB;                                   // (a row of four k-blocks: one k-block between the two barriers takes all of them)
    return u > MB ? u - (MB + 1) : (u < 2 * extra ? MB + u / 2 : MB + u - extra);
}
// fillers of k-block K: the units it cuts (COPS micro-operations each), the units it requests (LOPS each; LA k-blocks ahead of their cut), the staging micro-operations (k-blocks MB .. SE)
template <int NKB, int MB>
__host__ __device__ constexpr int wg4_ncut(int K) { int c = 0; for (int u = 0; u < NKB; ++u) c += wg4_cutK<NKB, MB>(u) == K ? 1 : 0; return c; }
template <int NKB, int MB, int LA>
__host__ __device__ constexpr int wg4_nld(int K) { int c = 0; for (int u = 0; u < NKB; ++u) c += (wg4_cutK<NKB, MB>(u) + NKB - LA) % NKB == K ? 1 : 0; return c; }
// first filler of filler gap gi (0 .. 2 NJ - 1) when NF fillers are dealt over a k-block: NF / 2 NJ per gap, the remainder one each to the FIRST gaps (a k-block's stores come early)
__host__ __device__ constexpr int wg4_deal(int gi, int NF, int NG) { return gi * (NF / NG) + (gi < NF % NG ? gi : NF % NG); }
// LDS operations a wave issues in k-block K BEHIND its last LDS store of that k-block (mirrors the dealing in the kernel: per tile j the gaps m = 0, 1 carry two transposed
// reads each, gap (1, 2) the two fragment reads of the next k-block, the fillers [wg4_deal(gi), wg4_deal(gi + 1)) sit behind the reads of gap gi = 2 j + m - 1): what
// s_waitcnt lgkmcnt(...) in front of the barrier that follows the k-block may leave outstanding
template <int NKB, int MB, int NJ, int STG, int ROW_MOPS, int LA, int LOPS, int COPS, int SE>
__host__ __device__ constexpr int wg4_lds_behind_last_store(int K)
{
    const int ncut = wg4_ncut<NKB, MB>(K), nld = wg4_nld<NKB, MB, LA>(K);
    const int nstg = (K >= MB && K <= SE) ? STG : 0;
    const int NF = COPS * ncut + LOPS * nld + nstg;
    int last = -1;
    for (int f = 0; f < NF; ++f) {
        bool st = false;
        if (f < COPS * ncut) st = (f % COPS == COPS - 3) || (f % COPS == COPS - 1);
        else if (f >= COPS * ncut + LOPS * nld) { const int ms = (K - MB) * STG + f - COPS * ncut - LOPS * nld; st = ms < ROW_MOPS && ms % WG4_IM >= WG4_IM - 2; }
        if (st) last = f;
    }
    if (last < 0) return 15;
    int gl = 0;
    for (int gi = 0; gi < 2 * NJ; ++gi) if (wg4_deal(gi, NF, 2 * NJ) <= last && last < wg4_deal(gi + 1, NF, 2 * NJ)) gl = gi;
    int n = 0;
    for (int j = 0; j < NJ; ++j)
        for (int m = 0; m < 3; ++m) {
            const int pos = 3 * j + m, lpos = 3 * (gl / 2) + gl % 2 + 1;      // (the store's gap: its own reads come in front of it)
            if (pos > lpos) n += (m < 2 ? 2 : 0) + ((j == 1 && m == 2) ? 2 : 0);
        }
    return n < 15 ? n : 15;
}

#ifdef WG4_STAMP                 // diagnostic build only: cycles per phase summed over a wave's rows -- [wave][slot]: 0 whole kernel, 1 its 100-MHz ticks, 2 prologue, 3 row start -> barrier (and the rest of the row), 4 wait at the barrier
#define WG4_ST(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define WG4_ST(k) do { } while (0)
#endif
#ifndef WG4_PD
#define WG4_PD 2
#endif
#define WG4_SBAR() __builtin_amdgcn_sched_barrier(0)
#define WG4_PIN(v) asm volatile("" : "+v"(v))

// W: output columns; TP: entries per column of a ring row = output depth + 2; RT: GEMM rows per tap (32: a tile is a tap; 28: the taps' 25 + 3 rows packed, 24 tiles -- every
// tile straddles taps); CIN: input channels (25: normConv; 32: the reducers); MODE 0: 'same' zero pads (normConv).  MODE 1: the first reducer -- the input is tf.pad(REFLECT) in
// rows and columns and NOT padded in depth (input depth TP, output depth TP - 2: a ring row holds all TP depths of all W + 2 columns, the pad columns as data), and dY is masked by
// the layer's own output (`gate` > 0: its ReLU).  MODE 2: the reducers behind it -- no pads at all (input extents H + 2, W + 2, TP: the ring's outer columns / rows ARE input), dY masked.
template <int W, int TP, int RT, int CIN, int MODE>
__global__ __launch_bounds__(256, 1) void conv3_wgrad_w4_kernel(Wg4Args a, const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gate,
                                                               float* __restrict__ partial, float* __restrict__ partial_b, Amax am)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool REFL = MODE != 0, VALID = MODE == 2;                  // REFL: what the two reducer modes share (no depth pads, outer columns staged as data, dY masked)
    static_assert((CIN == 25 && !REFL) || CIN == 32, "25 channels: the zero-padded 'same' layer only");
    constexpr int T = TP - 2, WP = W + 2, NV = W * T, NKB = (NV + 15) / 16;
    // W = 11: a workgroup takes HALF the columns of a 22-column row (depth 13: four slots of 24 x 15 entries are 184 KB, of 13 x 15 entries 100 KB) -- ring column 0 of the right
    // half and ring column W + 1 of the left one are input columns of the other half (mode 1: or the mirror column), the outer one the pad
    constexpr int NSPLIT = W == 11 ? 2 : 1;
    static_assert(NSPLIT == 1 || !VALID, "column halves: the padded modes only");
    constexpr int WOF = W * NSPLIT;                                      // columns of the output tensor
    constexpr int WI = VALID ? WP : WOF;                                 // columns of the input tensor
    constexpr int TD = REFL ? TP : T;                                    // depth of the input tensor
    constexpr int PP = WP * TP * 64, ROWB = 2 * PP;                      // bytes of one piece plane / of one ring row (slot)
    constexpr int NQV = REFL ? WP * TP : (NSPLIT == 2 ? (W + 1) * T : NV), QPV = CIN == 25 ? 7 : 8;    // voxels a row stages (REFL: the pad columns too; a half: + the neighbour's column); channel quads per voxel
    constexpr int NQI = NQV * QPV, NIT = (NQI + 255) / 256;             // staging items (voxel, channel quad) of a row; per thread
    constexpr int NT = (27 * RT + 31) / 32, NJ = (NT + 3) / 4;          // M tiles; per wave (tile w + 4 j)
    constexpr int RS = 5;                                                // register sets of the dY fragments: set = k-block % RS
    static_assert((NKB - 1) % RS != 0, "dY fragment sets: a row's last k-block and the next row's first must not share a set");
    constexpr int FB0 = 4 * ROWB + 1024;                                 // the row's dY fragments: [k-block][piece][lane] x 16 bytes, behind the ring (and 1 KB that a tail k-block's reads may run into)
    constexpr int MB = NKB / 2;                                          // the row's second barrier stands in front of k-block MB
    constexpr int LA = MB < 3 ? MB : 3;                                  // k-blocks a unit's requests run ahead of its cut
    constexpr bool LIFT = WG4_LIFT != 0;
    constexpr int LOPS = REFL ? 5 : 3, COPS = (REFL ? 10 : 8) + (LIFT ? 2 : 0);             // micro-operations of a unit's requests / of its cut (REFL: + the gate values, + two selects)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, col = lane & 31, li = lane & 15, gcol = (lane >> 4) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = a.nstrips * NSPLIT;
    const int n = blockIdx.x / per, srem = blockIdx.x - n * per;
    const int strip = srem / NSPLIT, sp = srem - strip * NSPLIT;         // sp: the workgroup's column half (0 without the split)
    const int hb = strip * a.SR;
    const int SRr = a.H - hb < a.SR ? a.H - hb : a.SR;
#ifdef WG4_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_pro = 0, st_prev = st_t0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    float sx = 1.f, sd = 1.f;                                            // H3 scales of x and dY (set in the prologue, behind its requests)
    // the operands' per-sample amax slots (at most 256 samples: wg4_plan), requested FIRST and read last: two dependent round trips through a loop (amax_over_samples)
    // in front of everything else were 4 000 cycles of the prologue
    unsigned amx[4], amw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int idx = lane + 64 * k < a.N ? lane + 64 * k : 0; amx[k] = am.x[idx]; amw[k] = am.w[idx]; }
    // ---- staging constants: item i <-> (voxel i / 7 of the row, channel quad i % 7); a thread's items are the same for every row.  Quad 6 is channel 24 and three values of
    // the next voxel, which take the scale 0.  A thread beyond the last item repeats it (the same bytes to the same place): no predicate anywhere in the staging.
    int s_src[NIT], s_dst[NIT];
    float s_scz[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k;
        const int ic = i < NQI ? i : NQI - 1;
        const int vox = ic / QPV, quad = ic - QPV * vox;
        if constexpr (REFL) {                                             // ring column cw <-> input column cw - 1, mirrored at both ends (tf.pad REFLECT); all TP depths
            const int cw = vox / TP, t = vox - cw * TP;
            const int gcw = cw + sp * W;                                   // the ring column in the full row's numbering
            const int sw = VALID ? cw : (gcw == 0 ? 1 : gcw == WOF + 1 ? WOF - 2 : gcw - 1);
            s_src[k] = ((sw * TP + t) * CIN + 4 * quad) * 4;
            s_dst[k] = (cw * TP + t) * 64 + quad * 8;
            s_scz[k] = 1.f;
        } else {
            const int w = vox / T, t = vox - w * T;
            const int cb = sp ? W * sp - 1 : 0;                            // first input column a half stages (its W + 1 columns are consecutive in memory)
            s_src[k] = ((vox + cb * T) * CIN + 4 * quad) * 4;
            s_dst[k] = ((w + ((NSPLIT == 2 && sp) ? 0 : 1)) * TP + t + 1) * 64 + quad * 8;
            s_scz[k] = (CIN == 25 && quad == 6) ? 0.f : 1.f;
        }
    }
    const long xsample = (long)(a.H + (VALID ? 2 : 0)) * WI * TD * CIN;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long)n * xsample), 0, (unsigned)xsample * 4u, 0x00020000);
    constexpr int xrowb = WI * TD * CIN * 4;
    auto row_of = [&](int key, int& rowoff, float& rokf) {               // ring key k <-> input row hb - 1 + k: clamped, 1.0 / 0.0 = inside / outside the patch -- REFL: mirrored
        int ih = hb - 1 + key;
        if constexpr (VALID) {
            ih = hb + key;
            ih = ih > a.H + 1 ? a.H + 1 : ih;                               // (rows a strip's last tiles stage for nobody)
            rowoff = ih * xrowb; rokf = 1.f;
        } else if constexpr (REFL) {
            ih = ih < 0 ? -ih : ih >= a.H ? 2 * a.H - 2 - ih : ih;
            ih = ih < 0 ? 0 : ih >= a.H ? a.H - 1 : ih;                    // (rows a strip's last tiles stage for nobody)
            rowoff = ih * xrowb; rokf = 1.f;
        } else {
            const bool rok = ih >= 0 && ih < a.H;
            rowoff = (rok ? ih : 0) * xrowb;
            rokf = rok ? 1.f : 0.f;
        }
    };
    u32x4b sv[NIT];                                                      // the row the current tile cuts (requested a tile earlier)
    auto stage_load = [&](int rowoff, u32x4b (&q)[NIT]) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) q[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_src[k], rowoff, 0);
    };
    // One row's cut and stores as micro-operations of one instruction each (dealt out over MFMA gaps).  Per item: its scale; the four first pieces fp16(v s) and the four
    // second pieces fp16(v s - h0) -- one mixed-precision fma each: v s is exact (a power of two) and so is the difference, the bits are those of multiply / convert / subtract /
    // convert --; the slot address; two 8-byte stores.  reload: the item's registers are requested again (the row the NEXT tile cuts) behind their last read.
    struct Cut { float sc, d0, d1; unsigned h0a, h0b, h1a, h1b; int adr; };
    constexpr int IM = WG4_IM;
    constexpr int ROW_MOPS = NIT * IM;
    float k11 = 2048.f;                                                  // (kept in a register: as a literal every use would be a move in front of it)
    WG4_PIN(k11);
    auto stage_mop = [&](int m, int slotoff, float sxr, u32x4b (&q)[NIT], Cut& c, bool reload, int rel) {
        const int k = m / IM, op = m % IM;
        if (op == 0) c.sc = s_scz[k] * sxr;
        // (order: a register's two halves are never written by consecutive instructions -- the partial write forwards one wait state late, and hipcc puts an s_nop between them)
        if (op == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h0a) : "v"(q[k][0]), "v"(sxr));
        if (op == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h0a) : "v"(q[k][1]), "v"(c.sc));
        if (op == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h0b) : "v"(q[k][2]), "v"(c.sc));
        if (op == 4) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h0b) : "v"(q[k][3]), "v"(c.sc));
        if constexpr (LIFT) {
            // the remainders v s - h0 in fp32 (exact), then lifted by 2^11 into the second pieces
            if (op == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.d0) : "v"(q[k][0]), "v"(sxr), "v"(c.h0a));
            if (op == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.d1) : "v"(q[k][2]), "v"(c.sc), "v"(c.h0b));
            if (op == 7) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h1a) : "v"(c.d0), "v"(k11));
            if (op == 8) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h1b) : "v"(c.d1), "v"(k11));
            if (op == 9) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.d0) : "v"(q[k][1]), "v"(c.sc), "v"(c.h0a));
            if (op == 10) {
                asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.d1) : "v"(q[k][3]), "v"(c.sc), "v"(c.h0b));
                if (reload) q[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_src[k], rel, 0);
            }
            if (op == 11) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h1a) : "v"(c.d0), "v"(k11));
            if (op == 12) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h1b) : "v"(c.d1), "v"(k11));
        } else {
            if (op == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.h1a) : "v"(q[k][0]), "v"(sxr), "v"(c.h0a));
            if (op == 7) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(c.h1a) : "v"(q[k][1]), "v"(c.sc), "v"(c.h0a));
            if (op == 6) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.h1b) : "v"(q[k][2]), "v"(c.sc), "v"(c.h0b));
            if (op == 8) {
                asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(c.h1b) : "v"(q[k][3]), "v"(c.sc), "v"(c.h0b));
                if (reload) q[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, s_src[k], rel, 0);
            }
        }
        if (op == IM - 3) c.adr = s_dst[k] + slotoff;
        if (op == IM - 2) *reinterpret_cast<u32x2b*>(lds + c.adr) = (u32x2b){c.h0a, c.h0b};
        if (op == IM - 1) *reinterpret_cast<u32x2b*>(lds + c.adr + PP) = (u32x2b){c.h1a, c.h1b};
    };

    // ---- prologue: the requests first (ring keys 0 .. 2 = this strip's first three input rows, key 3 for the first tile's staging), then the ring is cleared ----
    u32x4b p0[3][NIT];
    float p0_rok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { int ro; row_of(q, ro, p0_rok[q]); stage_load(ro, p0[q]); }
    // dY of an output row through its own descriptor: the k-blocks' tail voxels (beyond the row) are out of range and read as zero; a row beyond the strip has no records at all.
    // The B operand of a k-block (16 voxels x 32 channels) is cut ONCE per workgroup: a UNIT = one k-block's fragments, a quarter per wave -- lane l of wave w takes the
    // voxel pair l >> 4 of fragment lane 16 w + (l & 15): two 4-byte requests, four mixed fmas, two bias-sum additions, two 4-byte LDS stores (a wave's 64 stores of a piece
    // are 256 consecutive bytes).  Every wave then reads a k-block's two fragments with two ds_read_b128.  (Cut by every wave for itself -- the first version -- the
    // eight requests, sixteen fmas and eight additions per k-block were a third of all the instructions between the MFMAs.)
    const long dyrow0 = (((long)n * a.H + hb) * WOF + sp * W) * T * 32;
    auto dy_rsrc = [&](int i) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + dyrow0 + (long)i * WOF * T * 32), 0, i < SRr ? (unsigned)NV * 128u : 0u, 0x00020000); };
    const int lp = 16 * wave + li, pp = lane >> 4;                       // fragment lane (channel lp & 31, voxels 8 (lp >> 5) ..), pair of its eight voxels
    const int dvo = (8 * (lp >> 5) + 2 * pp) * 128 + (lp & 31) * 4;
    const int fbw = FB0 + lp * 16 + pp * 4, fbr = FB0 + lane * 16;
    auto gt_rsrc = [&](int i) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((REFL ? gate : dy) + dyrow0 + (long)i * WOF * T * 32), 0, i < SRr ? (unsigned)NV * 128u : 0u, 0x00020000); };
    float raw[NKB][2];                                                   // requested values of the units in flight (indexed by unit: compile-time everywhere)
    float rawg[REFL ? NKB : 1][2];                                       // REFL: the layer's output at the same voxels (its ReLU mask)
    __amdgpu_buffer_rsrc_t drs = dy_rsrc(0), drsN = dy_rsrc(1), grs = gt_rsrc(0), grsN = gt_rsrc(1);
    // Schedule (compile-time): unit u of a row is CUT in k-block cutK(u) -- units MB + 1 .. of the row itself in its k-blocks 0 .. (they are read from k-block MB on, behind
    // the second barrier), units 0 .. MB of the NEXT row in k-blocks MB .. NKB - 2 (their slots are free behind the second barrier; they are read from k-block NKB - 1 on,
    // behind the row barrier) -- and REQUESTED LA k-blocks earlier.
    auto cutK = [](int u) -> int { return wg4_cutK<NKB, MB>(u); };
    int vo_t = 0;
    unsigned cq0 = 0u, cq1 = 0u;
    float cd0 = 0.f, cd1 = 0.f;
    float bsum[2] = {0.f, 0.f};
    auto dy_load_op = [&](int u, int op, const __amdgpu_buffer_rsrc_t& rs, const __amdgpu_buffer_rsrc_t& gs) {
        if (op == 0) { vo_t = dvo + u * 2048; WG4_PIN(vo_t); }
        if (op == 1) raw[u][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo_t, 0, 0));
        if (op == 2) raw[u][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo_t + 128, 0, 0));
        if constexpr (REFL) {
            if (op == 3) rawg[u][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gs, vo_t, 0, 0));
            if (op == 4) rawg[u][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gs, vo_t + 128, 0, 0));
        }
    };
    auto dy_cut_op = [&](int u, int op) {                                 // (order: no register's halves by consecutive instructions, see stage_mop)
        if constexpr (REFL) {                                             // the ReLU mask first: dY where the layer's output is positive
            if (op == 0) { raw[u][0] = rawg[u][0] > 0.f ? raw[u][0] : 0.f; WG4_PIN(raw[u][0]); }
            if (op == 1) { raw[u][1] = rawg[u][1] > 0.f ? raw[u][1] : 0.f; WG4_PIN(raw[u][1]); }
            op -= 2;
        }
        if (op == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(cq0) : "v"(raw[u][0]), "v"(sd));
        if (op == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[0]) : "v"(raw[u][0]));
        if (op == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(cq0) : "v"(raw[u][1]), "v"(sd));
        if (op == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(bsum[1]) : "v"(raw[u][1]));
        if constexpr (LIFT) {
            if (op == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(cd0) : "v"(raw[u][0]), "v"(sd), "v"(cq0));
            if (op == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(cd1) : "v"(raw[u][1]), "v"(sd), "v"(cq0));
            if (op == 6) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(cq1) : "v"(cd0), "v"(k11));
            if (op == 7) *reinterpret_cast<unsigned*>(lds + fbw + u * 2048) = cq0;
            if (op == 8) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(cq1) : "v"(cd1), "v"(k11));
            if (op == 9) *reinterpret_cast<unsigned*>(lds + fbw + u * 2048 + 1024) = cq1;
        } else {
            if (op == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(cq1) : "v"(raw[u][0]), "v"(sd), "v"(cq0));
            if (op == 5) *reinterpret_cast<unsigned*>(lds + fbw + u * 2048) = cq0;
            if (op == 6) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(cq1) : "v"(raw[u][1]), "v"(sd), "v"(cq0));
            if (op == 7) *reinterpret_cast<unsigned*>(lds + fbw + u * 2048 + 1024) = cq1;
        }
    };
#pragma unroll
    for (int u = 0; u < NKB; ++u)
        if (u <= MB || cutK(u) < LA) {                                    // the first row's units 0 .. MB are cut here, and the units its first LA k-blocks cut are requested
#pragma unroll
            for (int op = 0; op < LOPS; ++op) dy_load_op(u, op, drs, grs);
        }
    // H3: both operands are contracted over the voxels of ALL samples, so each takes ONE scale: that of its largest sample (as the general form)
    unsigned mx = 0u, mw = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (lane + 64 * k < a.N) { mx = amx[k] > mx ? amx[k] : mx; mw = amw[k] > mw ? amw[k] : mw; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned vx = (unsigned)__shfl_xor((int)mx, o, 64), vw = (unsigned)__shfl_xor((int)mw, o, 64); mx = vx > mx ? vx : mx; mw = vw > mw ? vw : mw; }
    const int ex = h3_exp(mx), ed = h3_exp(mw);
    sx = pow2i(ex); sd = pow2i(ed);
    const int kun = -(ex + ed);
    // (All of it: what must be zero are the pad entries, the channel quad 28 .. 31 the staging never writes and the KB behind the ring -- 47 KB of the 160 --, but clearing
    //  just those was measured slower: the index arithmetic of the scattered 8-byte stores costs more than the 3 400 cycles the LDS takes for everything at 48 bytes per cycle.)
    for (int i = tid; i < (FB0 + NKB * 2048) / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    // transposed-read addresses of the lane, per (k-block, half of its eight voxels): voxel 16 kb + 8 h + 4 jj + (li >> 2) at tap (0, 0), channel quad 16 gcol + 4 (li & 3)
    int addrs[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int vi = 16 * kb + 8 * h + 4 * jj + (li >> 2);
            addrs[kb][jj] = (vi + 2 * (vi / T)) * 64;
        }
    // the wave's tiles: wave + 4 j.  A lane's four rows of a tile (16 gcol + 4 (li & 3) .. + 3) are four consecutive channels of ONE tap (RT is a multiple of 4): its tap's
    // ring-row step and (column, depth) offset plus the channels' byte offset.  (RT = 32: a tile beyond the 27th is tap 26 again: computed, dropped.)
    int tdh[NJ], tcc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int R0 = 32 * (wave + 4 * j) + 16 * gcol + 4 * (li & 3);
        int tau = R0 / RT;
        const int ci0 = R0 - tau * RT;
        tau = tau < 27 ? tau : 26;
        const int dh = tau / 9, dw = (tau / 3) % 3, dt = tau % 3;
        tdh[j] = dh; tcc[j] = (dw * TP + dt) * 64 + ci0 * 2;
    }
#ifdef WG4_STAMP
    const unsigned long long st_p1 = __builtin_amdgcn_s_memtime();
#endif
    // byte offsets of the lane's accumulator registers in the workgroup's slab [27 * 25][32] (worked out here, where the wave waits for its requests anyway): registers
    // 4 q .. 4 q + 3 of a lane are four consecutive rows R0 .. R0 + 3 of the GEMM with R0 a multiple of 4, and RT is one too: they belong to ONE tap.  A row that is no
    // (tap, input channel) pair gets an offset beyond the slab's descriptor and its store is dropped: so0 for register 4 q, so1 (+ 128 e) for registers 4 q + e
    int so0[NJ][4], so1[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int Tl = wave + 4 * j, Rj = 32 * Tl + 4 * h;
        int tau = Rj / RT, ci0 = Rj - tau * RT;                          // (one division per tile: the next quad is eight rows on, at most one tap further)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = Tl < NT && tau < 27;
            so0[j][q] = (ok && ci0 < CIN) ? ((tau * CIN + ci0) * 32 + col) * 4 : (int)0x40000000;
            so1[j][q] = (ok && ci0 + 3 < CIN) ? so0[j][q] : (int)0x40000000;
            ci0 += 8;
            if (ci0 >= RT) { ci0 -= RT; ++tau; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // ring cleared
#ifdef WG4_STAMP
    const unsigned long long st_p2 = __builtin_amdgcn_s_memtime();
#endif
    // (the row the first tile stages is requested behind the rows the first tile reads: every workgroup of the launch is in its prologue at once, and the 80 KB each
    //  asks for are what the prologue waits for -- 256 workgroups x 80 KB at the memory's rate)
    { int ro; float rk; row_of(3, ro, rk); stage_load(ro, sv); }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        Cut c;
        const float sxr = sx * p0_rok[q];
#pragma unroll
        for (int m = 0; m < ROW_MOPS; ++m) stage_mop(m, q * ROWB, sxr, p0[q], c, false, 0);
    }
#pragma unroll
    for (int u = 0; u <= MB; ++u)
#pragma unroll
        for (int op = 0; op < COPS; ++op) dy_cut_op(u, op);

#ifdef WG4_STAMP
    const unsigned long long st_p3 = __builtin_amdgcn_s_memtime();
#endif
    f32x16 acc[NJ], accL[LIFT ? NJ : 1];                                 // LIFT: acc = sum x0 d0, accL = sum (x1' d0 + x0 d1')
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; if (LIFT) accL[LIFT ? j : 0][r] = 0.f; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the first three rows are in the ring

    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    Frag A[NJ][2];                                                        // ring of operand fragments: [tile of the k-block][piece]
    Frag Bf[RS][2];                                                       // dY fragments of k-block K in set K % RS
    auto read_B = [&](int K) {
#pragma unroll
        for (int p = 0; p < 2; ++p) Bf[K % RS][p].u = *reinterpret_cast<const uint4*>(lds + fbr + K * 2048 + p * 1024);
    };
    read_B(0);
    int tapoff[NJ], tapoffN[NJ];
    auto set_taps = [&](int i, int (&to)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) to[j] = ((i + tdh[j]) & 3) * ROWB + tcc[j];
    };
    set_taps(0, tapoff);
    auto read_A = [&](int K, int j, const int (&to)[NJ], int piece) {
        const unsigned char* p0 = lds + addrs[K][0] + to[j];
        const unsigned char* p1 = lds + addrs[K][1] + to[j];
        A[j][piece].hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + piece * PP));
        A[j][piece].hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p1 + piece * PP));
    };
    constexpr int PD = WG4_PD;                                           // tiles the operand reads run ahead
#pragma unroll
    for (int j = 0; j < PD; ++j) { read_A(0, j, tapoff, 0); read_A(0, j, tapoff, 1); }
#ifdef WG4_STAMP
    st_pro = __builtin_amdgcn_s_memtime() - st_t0;
    st_prev = __builtin_amdgcn_s_memtime();
#endif

    // ---- the rows ----
    // A single wave issues one instruction per ~4.9 cycles whatever its kind (MI355X_MICROARCH.md; docs/notebook_r1-r5.md section 4.00): an MFMA leaves room for about five beside itself.  Per tile
    // the gap behind the first MFMA takes the two address sums and two transposed reads of the tile PD ahead (piece 0), the second gap the other two reads; a k-block's
    // fillers -- the units it cuts (COPS each) and requests (LOPS each), the staging micro-operations (k-blocks MB .. NKB - 2) -- are dealt evenly over its second and third gaps.
    // Two barriers per row (a wave arrives with s_waitcnt lgkmcnt(n), n = the LDS operations it has issued behind its last LDS store -- wg4_lds_behind_last_store):
    //   in front of k-block MB -- behind it the slots of units 0 .. MB are free (their fragments have been read) and the ring slot of input row i - 1 is (every wave has
    //   finished output row i - 1); in front of it the row's own units MB + 1 .. were stored;
    //   in front of k-block NKB - 1 -- behind it the next row's unit 0 and its first operands are read; in front of it the staged input row and the next row's units 0 .. MB were stored.
    constexpr int SE = NKB >= 8 ? NKB - 3 : NKB - 2;                     // k-blocks MB .. SE carry the staging (short rows: up to the row barrier's k-block)
    constexpr int NSK = SE - MB + 1;
    constexpr int STG = (ROW_MOPS + NSK - 1) / NSK;
    constexpr int WAIT_MB = wg4_lds_behind_last_store<NKB, MB, NJ, STG, ROW_MOPS, LA, LOPS, COPS, SE>(MB - 1), WAIT_RB = wg4_lds_behind_last_store<NKB, MB, NJ, STG, ROW_MOPS, LA, LOPS, COPS, SE>(NKB - 2);
    static_assert(wg4_lds_behind_last_store<NKB, MB, NJ, STG, ROW_MOPS, LA, LOPS, COPS, SE>(NKB - 1) == 15 && wg4_ncut<NKB, MB>(NKB - 1) == 0, "no LDS store in a row's last k-block");
#pragma unroll 1
    for (int i = 0; i < SRr; ++i) {
        set_taps(i + 1, tapoffN);
        int rel, dummy;
        float rokf, rokn;
        row_of(i + 3, dummy, rokf);                                      // the row cut and stored now (slot of key i + 3 = the slot of key i - 1)
        row_of(i + 4, rel, rokn);                                        // the row requested now
        float sxr = sx * rokf;
        WG4_PIN(sxr);
        const int slotoff = ((i + 3) & 3) * ROWB;
        Cut cut;
#pragma unroll
        for (int K = 0; K < NKB; ++K) {
            // the k-block's filler list
            const int ncut = wg4_ncut<NKB, MB>(K), nld = wg4_nld<NKB, MB, LA>(K);
            const int nstg = (K >= MB && K <= SE) ? STG : 0;
            const int NF = COPS * ncut + LOPS * nld + nstg;
            auto filler = [&](int f) {
#pragma unroll
                for (int u = 0; u < NKB; ++u) if (cutK(u) == K) { if (f >= 0 && f < COPS) dy_cut_op(u, f); f -= COPS; }
#pragma unroll
                for (int u = 0; u < NKB; ++u) if ((cutK(u) + NKB - LA) % NKB == K) {
                    // (a unit of the next row, or one of this row's successor that its first LA k-blocks cut: the next row's descriptor)
                    if (f >= 0 && f < LOPS) { if (u <= MB || cutK(u) < LA) dy_load_op(u, f, drsN, grsN); else dy_load_op(u, f, drs, grs); }
                    f -= LOPS;
                }
                if (f >= 0 && f < nstg) { const int ms = (K - MB) * STG + f; if (ms < ROW_MOPS) stage_mop(ms, slotoff, sxr, sv, cut, true, rel); }
            };
#ifndef WG4_ABL_NOBAR
            if (K == MB) { WG4_ST(3); asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" :: "n"(WAIT_MB) : "memory"); WG4_ST(4); }
            if (K == NKB - 1) { WG4_ST(3); asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" :: "n"(WAIT_RB) : "memory"); WG4_ST(4); }
#endif
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    {
                        const f16x8 af = (m == 0) ? A[j][1].h : A[j][0].h;
                        if (LIFT && m < 2) accL[LIFT ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, Bf[K % RS][m == 1 ? 1 : 0].h, accL[LIFT ? j : 0], 0, 0, 0);
                        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, Bf[K % RS][m == 1 ? 1 : 0].h, acc[j], 0, 0, 0);
                    }
                    WG4_SBAR();
                    // operand requests PD tiles ahead (a row's last PD: the next row's first, through its slots)
#ifndef WG4_ABL_NOREAD           // (timing-only ablations of the diagnostic build: tools/wg4diag.hip)
                    if (m < 2) {
                        const int sn = j + PD;
                        if (sn < NJ) read_A(K, sn, tapoff, m);
                        else if (K + 1 < NKB) read_A(K + 1, sn - NJ, tapoff, m);
                        else read_A(0, sn - NJ, tapoffN, m);
                    }
                    if (j == 1 && m == 2) read_B(K + 1 < NKB ? K + 1 : 0);      // the next k-block's dY fragments
#endif
                    // fillers
                    if (m > 0) {
                        const int gi = 2 * j + m - 1, f0 = wg4_deal(gi, NF, 2 * NJ), f1 = wg4_deal(gi + 1, NF, 2 * NJ);
#pragma unroll
                        for (int f = f0; f < f1; ++f) {
#ifdef WG4_ABL_NOFILL
                            continue;
#endif
                            filler(f);
                        }
                    }
                    WG4_SBAR();
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) tapoff[j] = tapoffN[j];
        drs = drsN; grs = grsN;
        drsN = dy_rsrc(i + 2); grsN = gt_rsrc(i + 2);
    }

#ifdef WG4_STAMP
    const unsigned long long st_end = __builtin_amdgcn_s_memtime();
#endif
    // ---- the slab of this workgroup: [27 * 25][32] (+ the bias sums).  Through a buffer descriptor: a GEMM row that is no (tap, input channel) pair gets an offset beyond
    // the slab and the store is dropped ----
    {
        float* pp = partial + (long)blockIdx.x * 27 * CIN * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pp, 0, 27 * CIN * 32 * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                {
                    const float v = LIFT ? fmaf(accL[LIFT ? j : 0][4 * q + e], 1.f / 2048.f, acc[j][4 * q + e]) : acc[j][4 * q + e];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ldexpf(v, kun)), rs, e ? so1[j][q] + e * 128 : so0[j][q], 0, 0);
                }
        // bias sums: a lane summed the voxel pairs pp of channel lp & 31; lanes l, l + 16, l + 32, l + 48 share the channel, and so do waves w and w + 2
        float b = bsum[0] + bsum[1];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the ring is dead)
        if (lane < 16) reinterpret_cast<float*>(lds)[wave * 16 + lane] = b;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < 32) partial_b[(long)blockIdx.x * 32 + tid] = reinterpret_cast<const float*>(lds)[(tid >> 4) * 16 + (tid & 15)] + reinterpret_cast<const float*>(lds)[((tid >> 4) + 2) * 16 + (tid & 15)];
    }
#ifdef WG4_STAMP
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_wg4_stamps + (blockIdx.x * 4 + wave) * 8;
        const unsigned long long te = __builtin_amdgcn_s_memtime();
        o[0] = te - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0; o[2] = st_pro; o[3] = st_acc[3]; o[4] = st_acc[4]; o[5] = te - st_end; o[6] = st_p2 - st_t0; o[7] = st_p3 - st_p2; o[3] = st_p1 - st_t0;
    }
#endif
}

#ifndef WG4_RT
#define WG4_RT 28                // (32: 27 tap tiles, conflict-free reads, 12.5 % more MFMAs -- measured 9 % slower, docs/notebook_r1-r5.md section 4.00)
#endif
// ---- the instances of the other translation units (the reducers': mirrored pads at depths 9 / 7 / 5 / 11 and 13 on column halves, the two unpadded layers) ----
#define WG4_SIG(W, TP, RT, C, M) __global__ void conv3_wgrad_w4_kernel<W, TP, RT, C, M>(Wg4Args, const float*, const float*, const float*, float*, float*, Amax)
#define WG4_PART1(X) X(22, 9, 32, 32, 1) X(22, 7, 32, 32, 1) X(22, 5, 32, 32, 1) X(18, 5, 32, 32, 2)
#define WG4_PART2(X) X(22, 11, 32, 32, 1) X(11, 13, 32, 32, 1) X(20, 7, 32, 32, 2)
#ifndef WG4_DIAG
#define WG4_DEF(W, TP, RT, C, M) template WG4_SIG(W, TP, RT, C, M);
#define WG4_EXT(W, TP, RT, C, M) extern template WG4_SIG(W, TP, RT, C, M);
#if WG4_PART == 0
WG4_PART1(WG4_EXT) WG4_PART2(WG4_EXT)
#elif WG4_PART == 1
WG4_PART1(WG4_DEF)
#else
WG4_PART2(WG4_DEF)
#endif
#undef WG4_DEF
#undef WG4_EXT
#endif

#if WG4_PART == 0
// ---- host side ----
static bool wg4_plan(const ConvGeom& g, const float* gate, Wg4Args& p, int& grid)
{
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.reflect_t || g.Cout != 32) return false;
    int nsplit = 1;
    if (g.Cin == 25) {                                                   // normConv: 'same' padding with zeros, no gate (depth 13: column halves)
        if (g.Wo != 22 || g.Wi != 22 || g.Ho != g.Hi || g.ph != 1 || g.pw != 1) return false;
        if (g.reflect_hw || g.pt != 1 || g.To != g.Ti || gate || (g.To != 13 && g.To != 9 && g.To != 7)) return false;
        if (g.To == 13) nsplit = 2;
    } else if (g.Cin == 32 && g.reflect_hw) {                            // the mirrored-pad reducers: tf.pad(REFLECT) in rows / columns, no depth pads, dY masked by the layer's output (depth 13 -> 11: column halves)
        if (g.Wo != 22 || g.Wi != 22 || g.Ho != g.Hi || g.ph != 1 || g.pw != 1) return false;
        if (g.pt != 0 || g.To != g.Ti - 2 || !gate || g.Hi < 2 || (g.To != 11 && g.To != 9 && g.To != 7 && g.To != 5 && g.To != 3)) return false;
        if (g.To == 11) nsplit = 2;
    } else if (g.Cin == 32) {                                            // the reducers behind it: no pads at all, dY masked
        if (g.ph != 0 || g.pw != 0 || g.pt != 0 || g.Ho != g.Hi - 2 || g.Wo != g.Wi - 2 || g.To != g.Ti - 2 || !gate) return false;
        if (!((g.Wo == 20 && g.To == 5) || (g.Wo == 18 && g.To == 3))) return false;
    } else return false;
    if (g.N < 1 || g.N * nsplit > 256 || g.Ho < 1) return false;         // one slab per workgroup, at most 256 of them (x6_wgrad_partial_floats)
    int nstrips = 256 / (g.N * nsplit);
    if (nstrips > g.Ho) nstrips = g.Ho;
    const int SR = (g.Ho + nstrips - 1) / nstrips;
    nstrips = (g.Ho + SR - 1) / SR;
    p.N = g.N; p.H = g.Ho; p.SR = SR; p.nstrips = nstrips; p.nsplit = nsplit;
    grid = g.N * nsplit * nstrips;
    return true;
}

#ifndef WG4_DIAG
static int g_wg4_enabled = -1;
bool wg4_enabled()
{
    if (g_wg4_enabled < 0) { const char* e = getenv("PROBAV_GEN1"); g_wg4_enabled = !(e && (e[0] == '1' || e[0] == 'w')); }      // PROBAV_GEN1 = 1 (every general form) | conv | wg | pw | pwf | pwb
    return g_wg4_enabled != 0;
}
void wg4_set_enabled(int on) { g_wg4_enabled = on ? 1 : 0; }
#endif

bool wg4_wgrad_supported(const ConvGeom& g, const float* gate)
{
    Wg4Args p;
    int grid;
    return wg4_plan(g, gate, p, grid);
}

int wg4_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db, float* partial, const Amax& am, hipStream_t s)
{
    Wg4Args p;
    int grid;
    if (!wg4_plan(g, gate, p, grid)) { set_error("wg4_conv_wgrad: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    if (!am.x || !am.w) { set_error("wg4_conv_wgrad: H3 arithmetic needs the per-sample amax slots of x (am.x) and dY (am.w)", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] {
#define WG4_BIG(W, TP, RT, C, M) note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wgrad_w4_kernel<W, TP, RT, C, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840))
        WG4_BIG(22, 11, WG4_RT, 25, 0); WG4_BIG(22, 9, WG4_RT, 25, 0); WG4_BIG(22, 9, 32, 32, 1); WG4_BIG(22, 7, 32, 32, 1); WG4_BIG(22, 5, 32, 32, 1);
        WG4_BIG(20, 7, 32, 32, 2); WG4_BIG(18, 5, 32, 32, 2);
        WG4_BIG(11, 15, WG4_RT, 25, 0); WG4_BIG(11, 13, 32, 32, 1); WG4_BIG(22, 11, 32, 32, 1); });
#undef WG4_BIG
    const long nw = (long)27 * g.Cin * g.Cout;
    float* partial_b = partial + (size_t)grid * nw;
    const int Tp = g.To + 2;
    const int Wk = g.Wo / p.nsplit;                                        // columns of a workgroup's rows
    const size_t lds_bytes = (size_t)4 * 2 * (Wk + 2) * Tp * 64 + 1024 + (size_t)((Wk * g.To + 15) / 16) * 2048;      // ring, slack, the row's dY fragments
#define WG4_LAUNCH(W, TP, RT, C, M) hipLaunchKernelGGL((conv3_wgrad_w4_kernel<W, TP, RT, C, M>), dim3(grid), dim3(256), lds_bytes, s, p, x, dy, gate, partial, partial_b, am)
    if (g.Cin == 25) { if (g.To == 13) WG4_LAUNCH(11, 15, WG4_RT, 25, 0); else if (g.To == 9) WG4_LAUNCH(22, 11, WG4_RT, 25, 0); else WG4_LAUNCH(22, 9, WG4_RT, 25, 0); }
    else if (!g.reflect_hw) { if (g.Wo == 20) WG4_LAUNCH(20, 7, 32, 32, 2); else WG4_LAUNCH(18, 5, 32, 32, 2); }
    else if (g.To == 11) WG4_LAUNCH(11, 13, 32, 32, 1);
    else if (g.To == 9) WG4_LAUNCH(22, 11, 32, 32, 1);
    else if (g.To == 7) WG4_LAUNCH(22, 9, 32, 32, 1);
    else if (g.To == 5) WG4_LAUNCH(22, 7, 32, 32, 1);
    else WG4_LAUNCH(22, 5, 32, 32, 1);
#undef WG4_LAUNCH
    int rc = check_launch("conv3_wgrad_w4");
    if (rc) return rc;
    return mfma_wgrad_reduce(partial, partial_b, dw, db, nw, g.Cout, grid, s);
}

#endif  // WG4_PART == 0

#ifdef WG4_DIAG
}  // namespace diag
#endif
}  // namespace probav
