// fp32 MFMA kernels for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).
//
//  conv3_mfma_kernel        3x3x3 convolution as an im2col-free implicit GEMM: M = 32 output voxels,
//                           N = 32 output channels, K = (tap, cin).  The input halo tile of a few output rows
//                           is staged ONCE in LDS ([row][w][t][c], odd channel stride => conflict-free
//                           ds_read_b32); the A operand of every one of the 27 taps is the same LDS image
//                           read at a shifted address, the B operand (filter) streams from L2 in a
//                           pre-packed fragment order.  Forward and backward-data (flipped, channel-swapped
//                           filter) are the same kernel.
//  conv3_wgrad_mfma_kernel  backward-filter: M = 32 rows of the flattened (tap, cin) filter matrix, N = cout,
//                           K = output voxels.  Persistent workgroups keep the filter-gradient tiles in
//                           accumulators across all their row tiles; per-workgroup slabs are summed in a fixed
//                           order afterwards (bitwise reproducible, no float atomics).
//  pw_fwd_mfma_kernel       fused expConv(1x1x1, 32->256)+ReLU -> decConv(1x1x1, 256->25): the 256-channel
//                           tensor lives only in accumulators (the first product's accumulator tile is the next
//                           MFMA's B operand, no LDS round trip), 1 KB/voxel/block of HBM traffic removed.
//
// Lane maps (MI355X guide §3): A: lane l holds A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31];
// D: register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
#include "probav_common.h"
#include "kernels_mfma.h"
#include "x6_device.h"
#include <type_traits>
#include <cstdint>

#include <mutex>

namespace probav {

// In-kernel phase stamps for tools/diag_conv.hip (a separate diagnostic build defines PROBAV_STAMP; the product
// library never does, so no stamp executes in it).  Stamps go to a buffer nothing else reads.
#if defined(PROBAV_STAMP_CLOCK) && !defined(PROBAV_STAMP)
__device__ unsigned long long g_stamps[8192 * 8];
#endif
#ifdef PROBAV_STAMP
__device__ unsigned long long g_stamps[8192 * 8];
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_DECL unsigned long long st_t0 = 0, st_fill = 0, st_steps = 0
#define STAMP_T0 st_t0 = __builtin_amdgcn_s_memtime()
#define STAMP_ACC(var) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - st_t0; st_t0 = t_; } while (0)
#define STAMP_OUT do { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_stamps[blockIdx.x * 8 + 1] = st_fill; g_stamps[blockIdx.x * 8 + 2] = st_steps; } } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_DECL do { } while (0)
#define STAMP_T0 do { } while (0)
#define STAMP_ACC(var) do { } while (0)
#define STAMP_OUT do { } while (0)
#endif
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// floor(v / d) for 0 <= v, v*d < 2^32, branch-free: m = ceil(2^32 / d), or m = 0 when d == 1 (then the quotient is v itself)
__device__ __forceinline__ int fdiv(int v, int d, unsigned m) { return (int)__umulhi((unsigned)v, m) + (d == 1 ? v : 0); }
__device__ __forceinline__ int reflect_clamped(int i, int n)
{
    i = i < -(n - 1) ? -(n - 1) : (i > 2 * n - 2 ? 2 * n - 2 : i);
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
static unsigned magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// XCD-aware block remap (guide §5.5 T1, bijective form): blocks b and b+8 share an XCD, so give each XCD a
// contiguous run of (patch, row) tiles -- neighbouring rows re-read each other's halo rows from that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---------------------------------------------------------------------------------------------------
// weight fragment packing (one launch per forward for all layers)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const PackJob* __restrict__ jobs, const float* __restrict__ weff,
                                                  const float* __restrict__ weffT, float* __restrict__ wpack,
                                                  const unsigned* __restrict__ amax)
{
    const PackJob J = jobs[blockIdx.y];
    const float* src = (J.src_is_T ? weffT : weff) + J.src_off;
    float* dst = wpack + J.dst_off;
    const bool h3 = J.type >= PACK_H3_PW_W1;
    const int type = h3 ? J.type - 10 : J.type;
    // H3 scale: the power of two of the matrix' output column / row `col` this lane packs (amax_percol: slot amax_slot + col) or of the
    // whole tensor (slot amax_slot); kernels that read the fragments undo exactly this exponent (h3_exp_w of the same slot)
    const float wscale_t = (h3 && !J.amax_percol) ? pow2i(h3_exp_w(amax[J.amax_slot])) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < J.count; i += (long)gridDim.x * 256) {
        const int u = (int)(i & 3), lane = (int)((i >> 2) & 63), half = lane >> 5, col = lane & 31;
        long q = i >> 8;                                   // float4 index (per lane)
        if (type >= PACK_X6_PW_W1) {
            // split fragments: this dword holds k-slots j = 2u, 2u+1 of lane `lane` in fragment q = (outer * 2 + kb) * NP + piece
            const int np = h3 ? 2 : 3;
            int piece = (int)(q % np), kb = (int)((q / np) & 1), outer = (int)(q / (2 * np));
            if (type == PACK_X6_CONVK || type == PACK_X6_CONVP) { kb = (int)((q / np) % 5); outer = (int)(q / (5 * np)); }   // fragment q = (group * 5 + kb) * NP + piece
            float v2[2] = {0.f, 0.f};
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * u + e, kn = 16 * kb + 8 * half + j, kp = rowmap(8 * kb + j, half);   // natural / accumulator-order k
                if (type == PACK_X6_PW_W1) {          // [row|col = hidden 32c + col][k = cin kn]              W1 [cin 32][hidden 256]
                    v2[e] = src[(long)kn * J.Cout + 32 * outer + col];
                } else if (type == PACK_X6_PW_W2) {   // [row = out col][k-slot = hidden 32c + kp]             W2 [hidden 256][out D]
                    if (col < J.Cout) v2[e] = src[(long)(32 * outer + kp) * J.Cout + col];
                } else if (type == PACK_X6_PW_W2K) {  // [row|col = hidden 32c + col][k = out kn]              W2 [hidden 256][out D]
                    if (kn < J.Cout) v2[e] = src[(long)(32 * outer + col) * J.Cout + kn];
                } else if (type == PACK_X6_PW_W1C) {  // [row = cin col][k-slot = hidden 32c + kp]             W1 [cin 32][hidden 256]
                    v2[e] = src[(long)col * J.Cout + 32 * outer + kp];
                } else if (type == PACK_X6_CONVP) {   // chunk c = 2 kb + half of group `outer`: (dt, ci) = (c / 3, 8 (c % 3) + j), chunk 9 = channel 24 of dt = j
                    const int c = 2 * kb + half;
                    const int dt = c < 9 ? c / 3 : j, ci = c < 9 ? 8 * (c % 3) + j : 24;
                    if ((c < 9 || j < 3) && col < J.Cout) v2[e] = src[((long)(outer * 3 + dt) * J.Cin + ci) * J.Cout + col];
                } else if (type == PACK_X6_CONVK) {   // [k = dt * Cin + ci][col = cout] of (dh, dw) group `outer`: taps 3*outer + dt
                    if (kn < 3 * J.Cin && col < J.Cout) v2[e] = src[((long)outer * 3 * J.Cin + kn) * J.Cout + col];
                } else {                                // PACK_X6_CONV: [k = cin kn][col = cout] of tap `outer`
                    if (kn < J.Cin && col < J.Cout) v2[e] = src[((long)outer * J.Cin + kn) * J.Cout + col];
                }
            }
            if (h3) {
                unsigned pc[2];
                const float wscale = J.amax_percol ? (col < J.ncol ? pow2i(h3_exp_w(amax[J.amax_slot + col])) : 1.f) : wscale_t;
                cut_pair<H3>(v2[0], v2[1], wscale, pc);
                reinterpret_cast<unsigned*>(dst)[i] = pc[piece];
            } else {
                unsigned pc[3];
                split_pair(v2[0], v2[1], pc[0], pc[1], pc[2]);
                reinterpret_cast<unsigned*>(dst)[i] = pc[piece];
            }
            continue;
        }
        float v = 0.f;
        if (J.type == PACK_CONV) {
            // [chunk][tap][q4][lane][4]; step s = 4*q4+u; k-pair channel = 2s+half of chunk
            const int KS4 = (J.KS + 3) >> 2;
            const int q4 = (int)(q % KS4); q /= KS4;
            const int tap = (int)(q % J.taps), chunk = (int)(q / J.taps);
            const int s = 4 * q4 + u, cl = 2 * s + half, ci = chunk * J.CC + cl;
            if (s < J.KS && cl < J.CC && ci < J.Cin && col < J.Cout) v = src[((long)tap * J.Cin + ci) * J.Cout + col];
        } else if (J.type == PACK_PW_A_KCIN) {
            // A[i = hch col of chunk c][k: cin = 16*half + s]   from W1 [cin 32][hch 256];  [c][q4][lane][4]
            const int q4 = (int)(q & 3), c = (int)(q >> 2), s = 4 * q4 + u;
            v = src[(long)(16 * half + s) * J.Cout + 32 * c + col];
        } else if (J.type == PACK_PW_A_KHCH) {
            // A[i = out col][k: hch = 32c + rowmap(s, half)]      from W2 [hch 256][out 25];  [c][q4][lane][4]
            const int q4 = (int)(q & 3), c = (int)(q >> 2), s = 4 * q4 + u;
            if (col < J.Cout) v = src[(long)(32 * c + (s & 3) + 8 * (s >> 2) + 4 * half) * J.Cout + col];
        } else if (J.type == PACK_PW_A_KOUT) {
            // A[i = hch col of chunk c][k: out = 13*half + s], s < 13 (16 slots)  from W2 [hch][out]
            const int q4 = (int)(q & 3), c = (int)(q >> 2), s = 4 * q4 + u, o = 13 * half + s;
            if (s < 13 && o < J.Cout) v = src[(long)(32 * c + col) * J.Cout + o];
        } else if (J.type == PACK_PW_A_CIN_KHCH) {
            // A[i = cin col][k: hch = 32c + rowmap(s, half)]     from W1 [cin 32][hch 256]
            const int q4 = (int)(q & 3), c = (int)(q >> 2), s = 4 * q4 + u;
            v = src[(long)col * J.Cout + 32 * c + (s & 3) + 8 * (s >> 2) + 4 * half];
        }
        dst[i] = v;
    }
}

int mfma_pack(const PackJob* d_jobs, int njobs, const float* weff, const float* weffT, float* wpack, const unsigned* amax, hipStream_t s)
{
    if (njobs <= 0) return PROBAV_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(32, njobs), dim3(256), 0, s, d_jobs, weff, weffT, wpack, amax);
    return check_launch("mfma_pack");
}

// ---------------------------------------------------------------------------------------------------
// shared: stage the input halo tile of `rows` rows into LDS, [r][wp][tp][c] with channel stride CP
// ---------------------------------------------------------------------------------------------------
struct TileArgs {
    ConvGeom g;
    int R, rows, Wp, Tp;          // output rows per tile, staged rows (R + kh - 1), staged width / depth
    int ntile_rows;               // ceil(Ho / R)
    unsigned mTp, mWp, mTo, mWoTo, mColE;   // magic multipliers for fdiv (mColE: Tp * loads-per-voxel)
    unsigned mTi, mSrcCol, mSrcRow;         // Ti ; Ti * CC (floats per source column) ; Wi * Ti * loads-per-voxel
};

template <int CC, int CP, bool REFLECT, bool GATE>
__device__ __forceinline__ void fill_tile_impl(const TileArgs& a, float* lds, const float* __restrict__ x,
                                               const float* __restrict__ gate, int n, int h0, int c0, int tid)
{
    // Scalar (wave-uniform) loop over the staged rows: row validity / reflection and the 64-bit row base pointer are
    // SGPR work; a lane only derives (column, plane, channel group) of its element from ONE divide and uses 32-bit
    // offsets.  No conditional loads and no conditional stores: a runtime `if` around a load (even a uniform one such
    // as `if (gate)`) makes hipcc branch and drain vmcnt(0) per element -- measured, that serialised fill took 58 % of
    // the workgroup's lifetime.  Out-of-range elements load from offset 0 of a valid row and select 0 afterwards; dead
    // lanes of a row's last batch store their 0 into the slack word behind the tile.
    const ConvGeom& g = a.g;
    constexpr int V = (CC % 4 == 0) ? 4 : 1;
    constexpr int CG = CC / V;                      // loads per voxel
    constexpr int U = (V == 4) ? 4 : 7;             // loads in flight per thread before the LDS stores
    const int colE = a.Tp * CG;                     // elements per (row, column)
    const int rowE = a.Wp * colE;                   // elements per staged row
    const int slack = a.rows * a.Wp * a.Tp * CP;
    for (int r = 0; r < a.rows; ++r) {
        int ih = h0 + r - g.ph;
        bool rok = true;
        if constexpr (REFLECT) ih = reflect_clamped(ih, g.Hi);
        else { rok = ih >= 0 && ih < g.Hi; ih = rok ? ih : 0; }
        const long rbase = (((long)n * g.Hi + ih) * g.Wi) * (long)g.Ti * g.Cin + c0;
        const float* xrow = x + rbase;
        const float* grow = GATE ? gate + rbase : nullptr;
        const int lrow = r * a.Wp * a.Tp * CP;
        for (int j0 = tid; j0 < rowE; j0 += 256 * U) {
            float4 val[U];
            int dof[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ju = j0 + u * 256;
                const bool live = ju < rowE;
                const int j = live ? ju : 0;
                const int wp = fdiv(j, colE, a.mColE), e = j - wp * colE;
                const int tp = e / CG, cg = e - tp * CG;             // CG is a compile-time constant
                int iw = wp - g.pw;
                const int it = tp - g.pt;
                bool ok = live && rok && it >= 0 && it < g.Ti;
                if constexpr (REFLECT) iw = reflect_clamped(iw, g.Wi);
                else ok = ok && iw >= 0 && iw < g.Wi;
                dof[u] = live ? lrow + (wp * a.Tp + tp) * CP + cg * V : slack;
                const int so = ok ? (iw * g.Ti + it) * g.Cin + cg * V : 0;
                if constexpr (V == 4) {
                    float4 v = *reinterpret_cast<const float4*>(xrow + so);
                    if constexpr (GATE) {
                        const float4 m = *reinterpret_cast<const float4*>(grow + so);
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    val[u].x = ok ? v.x : 0.f; val[u].y = ok ? v.y : 0.f; val[u].z = ok ? v.z : 0.f; val[u].w = ok ? v.w : 0.f;
                } else {
                    float v = xrow[so];
                    if constexpr (GATE) v = grow[so] > 0.f ? v : 0.f;
                    val[u] = make_float4(ok ? v : 0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float* d = lds + dof[u];
                d[0] = val[u].x;
                if constexpr (V == 4) {
                    const bool dead = dof[u] == slack;       // a dead lane's extra stores must stay on the slack word
                    d[dead ? 0 : 1] = val[u].y; d[dead ? 0 : 2] = val[u].z; d[dead ? 0 : 3] = val[u].w;
                }
            }
        }
    }
    if (tid == 0) lds[slack] = 0.f;          // slack word: the odd-K alias read of the last voxel lands here
}

// Zero-padded (non-reflect) layers: a source row [w][t][c] is contiguous in HBM and its image inside the padded LDS tile
// is the same sequence with a constant gap per column, so only VALID source elements are visited and the destination is
// `source index + gap * column + const` -- a handful of integer instructions per element, no validity tests.  The halo
// (pads, out-of-range rows) is produced by zeroing the whole tile first with 16-byte stores.
template <int CC, int CP, bool GATE>
__device__ __forceinline__ void fill_tile_linear(const TileArgs& a, float* lds, const float* __restrict__ x,
                                                 const float* __restrict__ gate, int n, int h0, int c0, int tid,
                                                 int rsel = -1, int rot = 0)
{
    // rsel < 0: stage every row of the tile; rsel >= 0: refill only tile row `rsel` (ring reuse: the other rows are
    // still in LDS from the previous output row).  Tile row r lives in LDS row slot (r + rot) % rows.
    const ConvGeom& g = a.g;
    constexpr int V = (CC % 4 == 0) ? 4 : 1;
    constexpr int CG = CC / V;
    constexpr int U = (V == 4) ? 4 : 10;            // loads in flight per thread (float4 resp. float)
    const int rowfloats = a.Wp * a.Tp * CP;
    const int tile_floats = a.rows * rowfloats + 1;                          // + slack word
    {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rsel < 0) {
            float4* z = reinterpret_cast<float4*>(lds);
            for (int i = tid; i < (tile_floats + 3) / 4; i += 256) z[i] = zero;   // the launch reserves (tile + 8) floats
        } else {                                                              // one row slot (rowfloats % 4 == 0 is required)
            float4* z = reinterpret_cast<float4*>(lds + ((rsel + rot) % a.rows) * rowfloats);
            for (int i = tid; i < rowfloats / 4; i += 256) z[i] = zero;
        }
    }
    __syncthreads();
    // wave-uniform loop over the staged rows (row validity and the 64-bit row base are scalar work); per element only
    // 32-bit offsets.  (Flattening the rows into the element index was tried: the per-element 64-bit row base made the
    // scalar path 2x slower.)
    const int srcE = g.Wi * g.Ti * CG;                                        // valid elements per source row
    const int dead_slot = tile_floats - 1;
    for (int r = (rsel < 0 ? 0 : rsel); r < (rsel < 0 ? a.rows : rsel + 1); ++r) {
        const int ih = h0 + r - g.ph;
        if (ih < 0 || ih >= g.Hi) continue;                                   // wave-uniform: row stays zero
        const long rbase = (((long)n * g.Hi + ih) * g.Wi) * (long)g.Ti * g.Cin + c0;
        const float* xrow = x + rbase;
        const float* grow = GATE ? gate + rbase : nullptr;
        const int lrow = (((r + rot) % a.rows) * a.Wp + g.pw) * a.Tp * CP + g.pt * CP;   // LDS offset of source voxel (w=0, t=0)
        for (int j0 = tid; j0 < srcE; j0 += 256 * U) {
            float4 val[U];
            int dof[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ju = j0 + u * 256;
                const bool live = ju < srcE;
                const int j = live ? ju : 0;
                int so, d;
                if constexpr (V == 1 && CC == CP) {
                    const int w = fdiv(j, g.Ti * CC, a.mSrcCol);              // source column
                    so = j + w * (g.Cin - CC) * g.Ti;                          // (Cin == CC here: so == j)
                    d = lrow + j + w * (a.Tp - g.Ti) * CP;
                } else {
                    const int vs = j / CG, cg = j - vs * CG;                   // source voxel (w*Ti + t), channel group
                    const int w = fdiv(vs, g.Ti, a.mTi);
                    so = vs * g.Cin + cg * V;
                    d = lrow + (vs + w * (a.Tp - g.Ti)) * CP + cg * V;
                }
                dof[u] = live ? d : dead_slot;
                if constexpr (V == 4) {
                    float4 v = *reinterpret_cast<const float4*>(xrow + so);
                    if constexpr (GATE) {
                        const float4 m = *reinterpret_cast<const float4*>(grow + so);
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    val[u].x = live ? v.x : 0.f; val[u].y = live ? v.y : 0.f; val[u].z = live ? v.z : 0.f; val[u].w = live ? v.w : 0.f;
                } else {
                    float v = xrow[so];
                    if constexpr (GATE) v = grow[so] > 0.f ? v : 0.f;
                    val[u] = make_float4(live ? v : 0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float* d = lds + dof[u];
                d[0] = val[u].x;
                if constexpr (V == 4) {
                    const bool dead = dof[u] == dead_slot;
                    d[dead ? 0 : 1] = val[u].y; d[dead ? 0 : 2] = val[u].z; d[dead ? 0 : 3] = val[u].w;
                }
            }
        }
    }
}

template <int CC, int CP>
__device__ __forceinline__ void fill_tile(const TileArgs& a, float* lds, const float* __restrict__ x,
                                          const float* __restrict__ gate, int n, int h0, int c0, int tid,
                                          int rsel = -1, int rot = 0)
{
    // straight-line instantiations chosen by ONE uniform branch (never a branch per element)
    if (a.g.reflect_hw) {
        if (gate) fill_tile_impl<CC, CP, true, true>(a, lds, x, gate, n, h0, c0, tid);
        else      fill_tile_impl<CC, CP, true, false>(a, lds, x, gate, n, h0, c0, tid);
    } else {
        if (gate) fill_tile_linear<CC, CP, true>(a, lds, x, gate, n, h0, c0, tid, rsel, rot);
        else      fill_tile_linear<CC, CP, false>(a, lds, x, gate, n, h0, c0, tid, rsel, rot);
    }
}

// LDS float offset of output voxel `vi` of the tile (row-major over (rr, w, t)) at tap (0,0,0)
__device__ __forceinline__ int tile_voxel_off(const TileArgs& a, int vi, int CP)
{
    const int rr = fdiv(vi, a.g.Wo * a.g.To, a.mWoTo), rem = vi - rr * a.g.Wo * a.g.To;
    const int w = fdiv(rem, a.g.To, a.mTo), t = rem - w * a.g.To;
    return ((rr * a.Wp + w) * a.Tp + t) * CP;
}

// ---------------------------------------------------------------------------------------------------
// conv3 forward / backward-data
// ---------------------------------------------------------------------------------------------------
template <int CC, int KS, int MT>
__device__ __forceinline__ void conv_taps(const TileArgs& a, const float* ldsA0, const float* ldsA1,
                                          const float4* __restrict__ wf, f32x16& acc0, f32x16& acc1)
{
    // Software pipeline, one tap deep: the A operands (LDS, shifted view of the halo tile) and the B fragments
    // (L2) of tap t+1 are requested BEFORE the MFMAs of tap t, which then run register-only, back to back.
    // hipcc on its own emits ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs and exposes the LDS latency on every
    // pair; the scheduling barriers pin the order written here.
    constexpr int CP = (CC & 1) ? CC : CC + 1;
    constexpr int KS4 = (KS + 3) / 4;
    float4 bcur[KS4], bnxt[KS4];
    float a0c[KS], a1c[KS], a0n[KS], a1n[KS];
#pragma unroll
    for (int q = 0; q < KS4; ++q) bcur[q] = wf[q * 64];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a0c[s] = ldsA0[2 * s]; a1c[s] = MT == 2 ? ldsA1[2 * s] : 0.f; }
#pragma unroll 1
    for (int tap = 0; tap < 27; ++tap) {
        const int tn = tap + 1 < 27 ? tap + 1 : tap;
        const int dh = tn / 9, dw = (tn / 3) % 3, dt = tn % 3;
        const int toff = ((dh * a.Wp + dw) * a.Tp + dt) * CP;
#pragma unroll
        for (int q = 0; q < KS4; ++q) bnxt[q] = wf[(tn * KS4 + q) * 64];
#pragma unroll
        for (int s = 0; s < KS; ++s) { a0n[s] = ldsA0[toff + 2 * s]; a1n[s] = MT == 2 ? ldsA1[toff + 2 * s] : 0.f; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float b = f4c(bcur[s >> 2], s & 3);
            acc0 = MFMA32(a0c[s], b, acc0);
            if (MT == 2) acc1 = MFMA32(a1c[s], b, acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KS4; ++q) bcur[q] = bnxt[q];
#pragma unroll
        for (int s = 0; s < KS; ++s) { a0c[s] = a0n[s]; a1c[s] = a1n[s]; }
    }
}

// x6 form of conv_taps for 16-channel chunks (CP = 17): one k-block per tap and chunk.  The activation operand is read as fp32
// from the halo tile (channels 8*half + j of the lane's voxel) and cut into bf16 pieces in registers; filters arrive pre-cut
// ([tap][chunk][piece][lane] x 16 B, PACK_X6_CONV).  Two M tiles share every filter fragment.
template <int MT, class AR>
__device__ __forceinline__ void conv_taps_x6(const TileArgs& a, const float* ldsA0, const float* ldsA1, const uint4* __restrict__ wf,
                                             int chunk, float sa, f32x16& acc0, f32x16& acc1)
{
    // groups (dh, dw) at run time, the three dt taps of a group unrolled: compile-time offsets inside a group (strip_taps_x6k)
    constexpr int CP = 17, NP = AR::NP;
    float r0[2][8], r1[2][8];
    Frag W[2][NP], a0[NP], a1[NP];
    auto group_off = [&](int g) -> int { const int dh = g / 3, dw = g - 3 * dh; return (dh * a.Wp + dw) * a.Tp * CP; };
    auto request = [&](int goff, const uint4* pw, int dt, Frag (&w)[NP], float (&q0)[8], float (&q1)[8]) {   // dt: compile-time
#pragma unroll
        for (int j = 0; j < 8; ++j) { q0[j] = ldsA0[goff + dt * CP + j]; q1[j] = MT == 2 ? ldsA1[goff + dt * CP + j] : 0.f; }
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(dt * 2 * NP + p) * 64];            // fragment ((tap * 2 + chunk) * NP + p), tap = 3 g + dt
    };
    int goff = group_off(0);
    const uint4* pw = wf + (long)chunk * NP * 64;
    request(goff, pw, 0, W[0], r0[0], r1[0]);
    cut8<AR>(r0[0], sa, a0);
    if (MT == 2) cut8<AR>(r1[0], sa, a1);
#pragma unroll 1
    for (int g = 0; g < 9; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g + u > 8) break;                                               // 9 groups: the second half of the last pair is empty
            const int gn = g + u + 1 <= 8 ? g + u + 1 : 8;
            const int goffn = group_off(gn);
            const uint4* pwn = wf + ((long)gn * 6 * NP + chunk * NP) * 64;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int s = 3 * u + dt;                                       // compile-time step parity (3 is odd)
                if (dt < 2) request(goff, pw, dt + 1, W[(s + 1) & 1], r0[(s + 1) & 1], r1[(s + 1) & 1]);
                else request(goffn, pwn, 0, W[(s + 1) & 1], r0[(s + 1) & 1], r1[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = mac<AR>(a0, W[s & 1], acc0);
                if (MT == 2) acc1 = mac<AR>(a1, W[s & 1], acc1);
                cut8<AR>(r0[(s + 1) & 1], sa, a0);
                if (MT == 2) cut8<AR>(r1[(s + 1) & 1], sa, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            goff = goffn; pw = pwn;
        }
    }
}

// conv_taps_x6 with operands requested two steps ahead (three buffers = the three dt steps of a group: static indices, a
// branch-free group body; see strip_taps_k2), used by the H3 arithmetic
template <int MT, class AR>
__device__ __forceinline__ void conv_taps_c2(const TileArgs& a, const float* ldsA0, const float* ldsA1, const uint4* __restrict__ wf,
                                             int chunk, float sa, f32x16& acc0, f32x16& acc1)
{
    constexpr int CP = 17, NP = AR::NP;
    float r0[3][8], r1[3][8];
    Frag W[3][NP], a0[NP], a1[NP];
    auto group_off = [&](int g) -> int { const int dh = g / 3, dw = g - 3 * dh; return (dh * a.Wp + dw) * a.Tp * CP; };
    auto request = [&](int goff, const uint4* pw, int dt, Frag (&w)[NP], float (&q0)[8], float (&q1)[8]) {   // dt: compile-time
#pragma unroll
        for (int j = 0; j < 8; ++j) { q0[j] = ldsA0[goff + dt * CP + j]; q1[j] = MT == 2 ? ldsA1[goff + dt * CP + j] : 0.f; }
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(dt * 2 * NP + p) * 64];
    };
    int goff = group_off(0);
    const uint4* pw = wf + (long)chunk * NP * 64;
    request(goff, pw, 0, W[0], r0[0], r1[0]);
    request(goff, pw, 1, W[1], r0[1], r1[1]);
    cut8<AR>(r0[0], sa, a0);
    if (MT == 2) cut8<AR>(r1[0], sa, a1);
#pragma unroll 1
    for (int g = 0; g < 9; ++g) {
        const int gn = g + 1 <= 8 ? g + 1 : 8;
        const int goffn = group_off(gn);
        const uint4* pwn = wf + ((long)gn * 6 * NP + chunk * NP) * 64;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            if (dt == 0) request(goff, pw, 2, W[2], r0[2], r1[2]);
            else request(goffn, pwn, dt - 1, W[dt - 1], r0[dt - 1], r1[dt - 1]);     // (after the last group: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            acc0 = mac<AR>(a0, W[dt], acc0);
            if (MT == 2) acc1 = mac<AR>(a1, W[dt], acc1);
            cut8<AR>(r0[(dt + 1) % 3], sa, a0);
            if (MT == 2) cut8<AR>(r1[(dt + 1) % 3], sa, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
        goff = goffn; pw = pwn;
    }
}

// x6 tap loop of the row-tile kernel for 25-channel inputs, dt taps concatenated along K (see strip_taps_x6k): 45 k-blocks of 16,
// filters PACK_X6_CONVK, two M tiles share every filter fragment.
template <int MT, class AR>
__device__ __forceinline__ void conv_taps_x6k(const TileArgs& a, const float* ldsA0, const float* ldsA1, const uint4* __restrict__ wf,
                                              float sa, f32x16& acc0, f32x16& acc1)
{
    constexpr int NP = AR::NP;
    float r0[2][8], r1[2][8];
    Frag W[2][NP], a0[NP], a1[NP];
    auto group_off = [&](int g) -> int { const int dh = g / 3, dw = g - 3 * dh; return (dh * a.Wp + dw) * a.Tp * 25; };
    auto request = [&](int goff, const uint4* pw, int kb, Frag (&w)[NP], float (&q0)[8], float (&q1)[8]) {   // kb: compile-time
#pragma unroll
        for (int j = 0; j < 8; ++j) { q0[j] = ldsA0[goff + 16 * kb + j]; q1[j] = MT == 2 ? ldsA1[goff + 16 * kb + j] : 0.f; }
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(kb * NP + p) * 64];
    };
    int goff = group_off(0);
    const uint4* pw = wf;
    request(goff, pw, 0, W[0], r0[0], r1[0]);
    cut8<AR>(r0[0], sa, a0);
    if (MT == 2) cut8<AR>(r1[0], sa, a1);
#pragma unroll 1
    for (int g = 0; g < 9; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g + u > 8) break;
            const int gn = g + u + 1 <= 8 ? g + u + 1 : 8;
            const int goffn = group_off(gn);
            const uint4* pwn = wf + (long)gn * 5 * NP * 64;
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const int s = 5 * u + kb;
                if (kb < 4) request(goff, pw, kb + 1, W[(s + 1) & 1], r0[(s + 1) & 1], r1[(s + 1) & 1]);
                else request(goffn, pwn, 0, W[(s + 1) & 1], r0[(s + 1) & 1], r1[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = mac<AR>(a0, W[s & 1], acc0);
                if (MT == 2) acc1 = mac<AR>(a1, W[s & 1], acc1);
                cut8<AR>(r0[(s + 1) & 1], sa, a0);
                if (MT == 2) cut8<AR>(r1[(s + 1) & 1], sa, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            goff = goffn; pw = pwn;
        }
    }
}

// AM: arithmetic of the tap loop -- 0 fp32 MFMA, 1 X6 (bf16 pieces), 2 H3 (scaled fp16 pieces; am.x / am.w = amax slots of the
// input tensor and of the filter, x6_device.h).  am.y (optional, any AM): slot that receives the largest output magnitude.
template <int CC, int KS, int AM>
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(TileArgs a, const float* __restrict__ x, const float* __restrict__ gate,
                                                           const float4* __restrict__ wfrag, const float* __restrict__ bias,
                                                           const float* __restrict__ skip, float* __restrict__ y, Amax am)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool X6 = AM != 0;
    using AR = std::conditional_t<AM == 2, H3, probav::X6>;
    constexpr int CP = (CC & 1) ? CC : CC + 1;
    constexpr int KS4 = (KS + 3) / 4;
    float sa = 1.f; int eun = 0; float omax = 0.f;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int n = wg / a.ntile_rows, h0 = (wg - n * a.ntile_rows) * a.R;
    // H3 scales: this patch's activations (slot n) and, per lane, the filter column the lane accumulates
    if constexpr (AM == 2) { const int ea = h3_exp(am.x[n]), ew = h3_exp_w(am.w[col < g.Cout ? col : 0]); sa = pow2i(ea); eun = -(ea + ew); }
    const int Rr = g.Ho - h0 < a.R ? g.Ho - h0 : a.R;
    const int nv = Rr * g.Wo * g.To, ntiles = (nv + 31) >> 5;
    const int nchunk = g.Cin / CC;
    const long out_base = ((long)n * g.Ho + h0) * g.Wo * g.To;

    // K-concatenated taps read 80 floats where a (dh, dw) group holds 75: the tile's last voxel runs 5 floats into the slack
    // words behind the tile.  Their filter rows are zero, but stale LDS bits that happen to spell Inf / NaN would still
    // poison the sum, so the slack is cleared (made visible by the barrier after fill_tile).
    if constexpr (X6 && CC == 25) { if (tid < 8) lds[a.rows * a.Wp * a.Tp * CP + tid] = 0.f; }
    STAMP(0);
    for (int pass = 0; pass * 8 < ntiles; ++pass) {
        const int t0 = pass * 8 + 2 * wave, t1 = t0 + 1;
        const bool v0 = t0 < ntiles, v1 = t1 < ntiles;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        int vi0 = t0 * 32 + col, vi1 = t1 * 32 + col;
        vi0 = vi0 < nv ? vi0 : nv - 1; vi1 = vi1 < nv ? vi1 : nv - 1;
        const float* ldsA0 = lds + tile_voxel_off(a, vi0, CP) + (X6 ? 8 * half : half);
        const float* ldsA1 = lds + tile_voxel_off(a, vi1, CP) + (X6 ? 8 * half : half);
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            if (pass == 0 || nchunk > 1) {
                if (pass > 0 || chunk > 0) __syncthreads();
                fill_tile<CC, CP>(a, lds, x, gate, n, h0, chunk * CC, tid);
                STAMP(1 + 2 * chunk);
                __syncthreads();
            }
            if constexpr (X6 && CC == 25) {
                const uint4* wf6 = reinterpret_cast<const uint4*>(wfrag) + lane;
                if (v1) conv_taps_x6k<2, AR>(a, ldsA0, ldsA1, wf6, sa, acc0, acc1);
                else if (v0) conv_taps_x6k<1, AR>(a, ldsA0, ldsA1, wf6, sa, acc0, acc1);
            } else if constexpr (X6) {
                const uint4* wf6 = reinterpret_cast<const uint4*>(wfrag) + lane;
                if constexpr (AM == 2) {
                    if (v1) conv_taps_c2<2, AR>(a, ldsA0, ldsA1, wf6, chunk, sa, acc0, acc1);
                    else if (v0) conv_taps_c2<1, AR>(a, ldsA0, ldsA1, wf6, chunk, sa, acc0, acc1);
                } else {
                    if (v1) conv_taps_x6<2, AR>(a, ldsA0, ldsA1, wf6, chunk, sa, acc0, acc1);
                    else if (v0) conv_taps_x6<1, AR>(a, ldsA0, ldsA1, wf6, chunk, sa, acc0, acc1);
                }
            } else {
                const float4* wf = wfrag + (long)chunk * 27 * KS4 * 64 + lane;
                if (v1) conv_taps<CC, KS, 2>(a, ldsA0, ldsA1, wf, acc0, acc1);
                else if (v0) conv_taps<CC, KS, 1>(a, ldsA0, ldsA1, wf, acc0, acc1);
            }
            STAMP(2 + 2 * chunk);
        }
        // epilogue: D row = output voxel, column = output channel.  All loads (bias, skip) and the arithmetic happen in
        // straight-line code first; full tiles (the common case) then store without any predicate.  (Predicated stores
        // written naively become one exec-masked block each with its own s_waitcnt vmcnt(0): measured 19k cycles.)
        const float bv = (bias && col < g.Cout) ? bias[col] : 0.f;
        float* ybase = y + out_base * g.Cout;                     // wave-uniform; per-lane offsets stay 32-bit
        const float* sbase = skip ? skip + out_base * g.Cout : nullptr;
#pragma unroll
        for (int tsel = 0; tsel < 2; ++tsel) {
            const int tt = tsel ? t1 : t0;
            if (tt >= ntiles) continue;
            const bool full = (tt * 32 + 32 <= nv) && g.Cout == 32;          // wave-uniform
            int oo[16];
            float ov[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vi = tt * 32 + rowmap(r, half);
                const bool live = vi < nv && col < g.Cout;
                oo[r] = live ? vi * g.Cout + col : -1;
            }
            if (sbase) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = sbase[oo[r] < 0 ? 0 : oo[r]];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = tsel ? acc1[r] : acc0[r];
                if constexpr (AM == 2) v = ldexpf(v, eun);                // back from the operands' power-of-two scales
                v += bv;
                if (g.relu) v = fmaxf(v, 0.f);
                ov[r] += v;
                omax = fmaxf(omax, oo[r] >= 0 ? fabsf(ov[r]) : 0.f);
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ybase[oo[r]] = ov[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (oo[r] >= 0) ybase[oo[r]] = ov[r];
            }
        }
    }
    if (am.y) amax_commit(omax, am.y + n);
    STAMP(7);
}

struct ConvPlan { bool ok; int CC, KS, R; size_t lds_bytes; TileArgs a; };

static ConvPlan conv_plan(const ConvGeom& g, bool all_channels)
{
    ConvPlan p;
    p.ok = false;
    if (g.kh != 3 || g.kw != 3 || g.kt != 3) return p;
    if (g.Cout > 32 && !all_channels) return p;
    int CC;
    if (all_channels) CC = g.Cin;                           // wgrad stages every input channel
    else if (g.Cin == 25) CC = 25;
    else if (g.Cin == 1) CC = 1;
    else if (g.Cin % 16 == 0) CC = 16;
    else return p;
    if (all_channels && g.Cin != 25 && g.Cin != 32 && g.Cin != 1) return p;
    // consistency of the geometry: a stride-1 correlation with these pads
    if (g.Ho != g.Hi + 2 * g.ph - 2 || g.Wo != g.Wi + 2 * g.pw - 2 || g.To != g.Ti + 2 * g.pt - 2) return p;
    const int CP = (CC & 1) ? CC : CC + 1;
    const int Wp = g.Wo + 2, Tp = g.To + 2;
    int R = 0;
    size_t lds = 0;
    const size_t lim2 = 81920, lim1 = 163840;
    for (int r = 1; r <= g.Ho; ++r) {
        const size_t need = ((size_t)(r + 2) * Wp * Tp * CP + 8) * sizeof(float);
        const int nv = r * g.Wo * g.To;
        if (r > 1 && (nv > 256 || need > lim2)) break;
        if (r == 1 && need > lim1) return p;
        R = r; lds = need;
    }
    if (R == 0) return p;
    if (R * g.Wo * g.To > 512) return p;                    // at most two passes of 8 tiles
    p.ok = true; p.CC = CC; p.KS = (CC + 1) / 2; p.R = R; p.lds_bytes = (lds + 15) & ~(size_t)15;
    p.a.g = g; p.a.R = R; p.a.rows = R + 2; p.a.Wp = Wp; p.a.Tp = Tp; p.a.ntile_rows = (g.Ho + R - 1) / R;
    p.a.mTp = magic(Tp); p.a.mWp = magic(Wp); p.a.mTo = magic(g.To); p.a.mWoTo = magic(g.Wo * g.To);
    p.a.mColE = magic(Tp * ((CC % 4 == 0) ? CC / 4 : CC));
    p.a.mTi = magic(g.Ti); p.a.mSrcCol = magic(g.Ti * CC); p.a.mSrcRow = magic(g.Wi * g.Ti * ((CC % 4 == 0) ? CC / 4 : CC));
    return p;
}

bool mfma_conv_supported(const ConvGeom& g) { return conv_plan(g, false).ok; }

size_t mfma_conv_wfrag_floats(int Cin, int Cout)
{
    if (Cout > 32) return 0;
    int CC;
    if (Cin == 25) CC = 25; else if (Cin == 1) CC = 1; else if (Cin % 16 == 0) CC = 16; else return 0;
    const int KS = (CC + 1) / 2, KS4 = (KS + 3) / 4;
    return (size_t)(Cin / CC) * 27 * KS4 * 256;
}

void mfma_conv_pack_job(PackJob& J, int Cin, int Cout)
{
    J.type = PACK_CONV; J.Cin = Cin; J.Cout = Cout; J.taps = 27;
    J.CC = Cin == 25 ? 25 : (Cin == 1 ? 1 : 16); J.KS = (J.CC + 1) / 2;
    J.count = (long)mfma_conv_wfrag_floats(Cin, Cout);
}

template <typename K>
static void allow_big_lds(K kernel)
{
    note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
}

int mfma_conv_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag, const float* bias,
                      const float* skip, float* y, const Amax& am, hipStream_t s)
{
    const ConvPlan p = conv_plan(g, false);
    if (!p.ok) { set_error("mfma_conv_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    const dim3 grid((unsigned)(g.N * p.a.ntile_rows)), block(256);
    static std::once_flag once;
    std::call_once(once, [] { allow_big_lds(conv3_mfma_kernel<25, 13, 0>); allow_big_lds(conv3_mfma_kernel<16, 8, 0>); allow_big_lds(conv3_mfma_kernel<1, 1, 0>); });
    if (p.CC == 1) hipLaunchKernelGGL((conv3_mfma_kernel<1, 1, 0>), grid, block, p.lds_bytes, s, p.a, x, gate, (const float4*)wfrag, bias, skip, y, am);
    else if (p.CC == 25) hipLaunchKernelGGL((conv3_mfma_kernel<25, 13, 0>), grid, block, p.lds_bytes, s, p.a, x, gate, (const float4*)wfrag, bias, skip, y, am);
    else            hipLaunchKernelGGL((conv3_mfma_kernel<16, 8, 0>), grid, block, p.lds_bytes, s, p.a, x, gate, (const float4*)wfrag, bias, skip, y, am);
    return check_launch("conv3_mfma");
}

// row-tile kernel with a split-operand tap loop: 32-channel inputs (two 16-channel chunks), any pads / reflect (reducers, upscale)
bool x6_conv_rowtile_supported(const ConvGeom& g)
{
    const ConvPlan p = conv_plan(g, false);
    return p.ok && ((p.CC == 16 && g.Cin == 32) || (p.CC == 25 && g.Cin == 25));
}
int x6_conv_rowtile_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag6, const float* bias,
                            const float* skip, float* y, int arith, const Amax& am, hipStream_t s)
{
    const ConvPlan p = conv_plan(g, false);
    if (!x6_conv_rowtile_supported(g)) { set_error("x6_conv_rowtile_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    if (arith == 2 && (!am.x || !am.w)) { set_error("x6_conv_rowtile_forward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
    const dim3 grid((unsigned)(g.N * p.a.ntile_rows)), block(256);
    static std::once_flag once;
    std::call_once(once, [] {
        allow_big_lds(conv3_mfma_kernel<16, 8, 1>); allow_big_lds(conv3_mfma_kernel<25, 13, 1>);
        allow_big_lds(conv3_mfma_kernel<16, 8, 2>); allow_big_lds(conv3_mfma_kernel<25, 13, 2>); });
#define PROBAV_RT(C, K, A) hipLaunchKernelGGL((conv3_mfma_kernel<C, K, A>), grid, block, p.lds_bytes, s, p.a, x, gate, (const float4*)wfrag6, bias, skip, y, am)
    if (arith == 2) { if (p.CC == 25) PROBAV_RT(25, 13, 2); else PROBAV_RT(16, 8, 2); }
    else            { if (p.CC == 25) PROBAV_RT(25, 13, 1); else PROBAV_RT(16, 8, 1); }
#undef PROBAV_RT
    return check_launch("conv3_mfma_x6");
}

// ---------------------------------------------------------------------------------------------------
// conv3 forward / backward-data, "strip" form (zero-padded layers whose output row has >= 128 voxels)
//
// One workgroup = a strip of consecutive output rows of ONE patch.  What it fixes relative to conv3_mfma_kernel
// (profiles/r01_pmc_summary.csv: 56 % MFMA-busy there):
//   * input rows live in a 5-slot LDS ring: every row is staged ONCE (not three times), and asynchronously -- the loads
//     of the next row are issued before a round's MFMAs and written to LDS after them;
//   * M tiles are cut from the flattened voxel stream of the strip, so only the strip's last tile is partial (a 22x9 row
//     is 6.19 tiles: the row-tile kernel pays 7 and leaves one of its four waves half idle);
//   * 8 waves: waves w and w+4 split the 27 taps of the SAME tile (14 / 13) and meet in LDS.  Two partial sums add
//     commutatively, so the result does not depend on arrival order (bitwise reproducible); both waves sit on the same
//     SIMD, which therefore always has two independent MFMA streams;
//   * one workgroup per CU, no tail: a round = 4 tiles = 128 voxels, one barrier (+ a short one for the exchange).
// Inputs with more channels than CC run nchunk passes over the strip; pass p > 0 adds to the partial output of pass p-1
// (re-read from L2), the last pass applies bias / ReLU / skip.
// ---------------------------------------------------------------------------------------------------
struct StripArgs {
    ConvGeom g;
    int Wp, Tp;                 // staged width / depth (Wo + 2, To + 2; piece-ring kernel: Wt + 2)
    int SR, nstrips;            // output rows per strip, strips per patch
    unsigned mTo, mNvr, mTi, mSrcCol;
    int nsplit, Wt;             // piece-ring kernel: output rows cut into nsplit column ranges of Wt columns when four full rows do not fit the LDS
    int nslot;                  // alternating-halves kernel: ring depth
    unsigned mNslot;
};


template <int CC, int KS, typename Mid>
__device__ __forceinline__ void strip_taps(const StripArgs& a, const float* lds, int base0, int base1, int base2, int tap0, int ntap,
                                           const float4* __restrict__ wf, f32x16& acc, int ks, Mid mid)
{
    // (CC == 32: the filter fragments are the two packed 16-channel chunks [chunk][tap][2][lane]; k-steps 0..7 come from
    //  chunk 0, 8..15 from chunk 1)
    // taps tap0 .. tap0+ntap-1 of one tile; the A operands (LDS) and B fragments (L2) of tap i+1 are requested before
    // the MFMAs of tap i, which then run register-only (scheduling barriers pin that order).  The loop stays rolled:
    // fully unrolled it needed > 400 VGPRs.
    constexpr int CP = (CC & 1) ? CC : CC + 1;
    constexpr int KS4 = (KS + 3) / 4;
    float4 bcur[KS4], bnxt[KS4];
    float acur[KS], anxt[KS];
    auto a_ptr = [&](int tap) -> const float* {
        const int dh = tap / 9, dw = (tap / 3) % 3, dt = tap % 3;             // wave-uniform
        const int b = dh == 0 ? base0 : (dh == 1 ? base1 : base2);                  // (three scalars, not an array: an indexed
                                                                                  //  array lands in scratch memory, one load per tap)
        return lds + b + (dw * a.Tp + dt) * CP;
    };
    auto b_load = [&](int tap, float4 (&dst)[KS4]) {
        if constexpr (CC == 32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = wf[(((q >> 1) * 27 + tap) * 2 + (q & 1)) * 64];
        } else {
#pragma unroll
            for (int q = 0; q < KS4; ++q) dst[q] = wf[(tap * KS4 + q) * 64];
        }
    };
    {
        const float* pa = a_ptr(tap0);
        b_load(tap0, bcur);
#pragma unroll
        for (int s = 0; s < KS; ++s) acur[s] = pa[2 * s];
    }
#pragma unroll 1
    for (int i = 0; i < ntap; ++i) {
        const int tn = (i + 1 < ntap) ? tap0 + i + 1 : tap0 + i;
        const float* pa = a_ptr(tn);
        b_load(tn, bnxt);
#pragma unroll
        for (int s = 0; s < KS; ++s) anxt[s] = pa[2 * s];
        if (i == 1) mid();      // other global requests of the round go HERE: vmcnt retires in order, so behind the filters of tap 2
                                // they get two taps of slack; in front of the loop they made the first tap wait for HBM
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s < ks) acc = MFMA32(acur[s], f4c(bcur[s >> 2], s & 3), acc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KS4; ++q) bcur[q] = bnxt[q];
#pragma unroll
        for (int s = 0; s < KS; ++s) acur[s] = anxt[s];
    }
}

// x6 form of the tap loop: the activation operand is read as fp32 from the same LDS ring (16 channels per k-block:
// channels 16kb + 8*half + j of the lane's voxel) and cut into bf16 pieces in registers; the filter operand arrives
// pre-split from L2 ([tap][kb][piece][lane] x 16 B, X6_CONV packing).  For CC = 25 the second k-block reads seven
// floats past the voxel's channels (the next voxel's, finite) against zero filter pieces.
// Three-stage software pipeline per tap: loads of tap i+1 | split of tap i+1's activations | MFMAs of tap i.
template <int CC, class AR, typename Mid>
__device__ __forceinline__ void strip_taps_x6(const StripArgs& a, const float* lds, int base0, int base1, int base2, int g0, int ng,
                                              const uint4* __restrict__ wf, float sa, f32x16& acc, Mid mid)
{
    // This wave's (dh, dw) groups g0 .. g0 + ng - 1; inside a group the three dt taps x two k-blocks are unrolled, so every LDS and
    // filter offset of a step is a compile-time constant from one pointer per group (see strip_taps_x6k for why that matters).
    // Operands are requested one step ahead (two buffers; six steps per group keep the buffer parity static).
    constexpr int CP = (CC & 1) ? CC : CC + 1, NP = AR::NP;
    const int rowstep = a.Tp * CP;
    const int glast = g0 + ng - 1;
    auto group_ptr = [&](int g) -> const float* {
        const int dh = g / 3, dw = g - 3 * dh;                                  // wave-uniform
        const int b = dh == 0 ? base0 : (dh == 1 ? base1 : base2);
        return lds + b + dw * rowstep;
    };
    float R[2][8];
    Frag W[2][NP], acur[NP];
    auto request = [&](const float* pa, const uint4* pw, int st, Frag (&w)[NP], float (&r)[8]) {   // st = dt * 2 + kb: compile-time
        const int dt = st >> 1, kb = st & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pa[dt * CP + 16 * kb + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(st * NP + p) * 64];
    };
    const float* pa = group_ptr(g0);
    const uint4* pw = wf + (long)g0 * 6 * NP * 64;
    request(pa, pw, 0, W[0], R[0]);
    cut8<AR>(R[0], sa, acur);
    bool did_mid = false;
#pragma unroll 1
    for (int g = g0; g <= glast; ++g) {
        const int gn = g + 1 <= glast ? g + 1 : glast;
        const float* pan = group_ptr(gn);
        const uint4* pwn = wf + (long)gn * 6 * NP * 64;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            if (st < 5) request(pa, pw, st + 1, W[(st + 1) & 1], R[(st + 1) & 1]);
            else request(pan, pwn, 0, W[0], R[0]);                              // (after the last group: a harmless re-read)
            if (st == 2 && !did_mid) { mid(); did_mid = true; }
            __builtin_amdgcn_sched_barrier(0);
            acc = mac<AR>(acur, W[st & 1], acc);
            cut8<AR>(R[(st + 1) & 1], sa, acur);
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = pan; pw = pwn;
    }
}

// x6 tap loop for Cin = 25 with the three dt taps of a (dh, dw) group CONCATENATED along K: with an unpadded channel stride of
// 25 the voxels (w, t), (w, t+1), (w, t+2) are 75 contiguous floats, i.e. 5 k-blocks of 16 instead of 3 taps x 2 k-blocks of a
// 25 -> 32 padded K (-17 % MFMAs and cuts).  k-blocks kbi = group * 5 + kb, kbi in [kb0, kb0 + nkb); filters: PACK_X6_CONVK.
template <class AR, typename Mid>
__device__ __forceinline__ void strip_taps_x6k(const StripArgs& a, const float* lds, int base0, int base1, int base2, int g0, int ng,
                                               const uint4* __restrict__ wf, float sa, f32x16& acc, Mid mid)
{
    constexpr int NP = AR::NP;
    // This wave's (dh, dw) groups g0 .. g0 + ng - 1, five k-blocks each.  The loop is NESTED -- groups at run time, the five blocks of
    // a group unrolled -- so that everything inside a group is a compile-time offset from one LDS pointer and one filter pointer:
    // decoding a flat block index (divisions by 5 and 3, clamps, 64-bit pointer arithmetic) cost ~25 dependent scalar instructions
    // per block in the wave's in-order stream and was what bounded the loop (removing the MFMAs did not change its time).
    // Operands are requested one block ahead (two buffers each); the group loop is unrolled by two so the buffer parity is static.
    const int rowstep = a.Tp * 25;
    const int glast = g0 + ng - 1;
    auto group_ptr = [&](int g) -> const float* {
        const int dh = g / 3, dw = g - 3 * dh;                                  // wave-uniform
        const int b = dh == 0 ? base0 : (dh == 1 ? base1 : base2);
        return lds + b + dw * rowstep;
    };
    float R[2][8];
    Frag W[2][NP], acur[NP];
    auto request = [&](const float* pa, const uint4* pw, int kb, Frag (&w)[NP], float (&r)[8]) {   // kb: compile-time
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pa[16 * kb + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(kb * NP + p) * 64];
    };
    const float* pa = group_ptr(g0);
    const uint4* pw = wf + (long)g0 * 5 * NP * 64;
    request(pa, pw, 0, W[0], R[0]);
    cut8<AR>(R[0], sa, acur);
    bool did_mid = false;
#pragma unroll 1
    for (int g = g0; g <= glast; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int gg = g + u;                                               // this group (may be one past the end: skipped)
            if (gg > glast) break;                                              // wave-uniform
            const int gn = gg + 1 <= glast ? gg + 1 : glast;
            const float* pan = group_ptr(gn);
            const uint4* pwn = wf + (long)gn * 5 * NP * 64;
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const int s = 5 * u + kb;                                       // compile-time step parity (5 is odd: parity flips per group)
                if (kb < 4) request(pa, pw, kb + 1, W[(s + 1) & 1], R[(s + 1) & 1]);
                else request(pan, pwn, 0, W[(s + 1) & 1], R[(s + 1) & 1]);      // (after the last group: a harmless re-read)
                if (kb == 2 && !did_mid) { mid(); did_mid = true; }
                __builtin_amdgcn_sched_barrier(0);
                acc = mac<AR>(acur, W[s & 1], acc);
                cut8<AR>(R[(s + 1) & 1], sa, acur);
                __builtin_amdgcn_sched_barrier(0);
            }
            pa = pan; pw = pwn;
        }
    }
}

// The same loop with operands requested PF k-blocks ahead, used by the H3 arithmetic: with half the MFMAs and half the cutting per
// k-block a wave's step became shorter than the L2 latency of its filter fragments, and with one block of look-ahead every step
// ended up waiting for them (709 cycles per step, 96 of them MFMA).  Five buffers = the five k-blocks of a group, so a buffer's
// index is its k-block (static) and the loop body is one branch-free group: hipcc's s_waitcnt insertion keeps exact counts only
// inside a basic block (the partly unrolled form with an early exit per group waited for lgkmcnt(0) / vmcnt(0) after every branch).
template <class AR, typename Mid>
__device__ __forceinline__ void strip_taps_k2(const StripArgs& a, const float* lds, int base0, int base1, int base2, int g0, int ng,
                                              const uint4* __restrict__ wf, float sa, f32x16& acc, Mid mid)
{
    constexpr int NP = AR::NP, PF = 3;
    const int rowstep = a.Tp * 25;
    const int glast = g0 + ng - 1;
    auto group_ptr = [&](int g) -> const float* {
        const int dh = g / 3, dw = g - 3 * dh;                                  // wave-uniform
        const int b = dh == 0 ? base0 : (dh == 1 ? base1 : base2);
        return lds + b + dw * rowstep;
    };
    float R[5][8];
    Frag W[5][NP], acur[NP];
    auto request = [&](const float* pa, const uint4* pw, int kb, Frag (&w)[NP], float (&r)[8]) {   // kb: compile-time
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pa[16 * kb + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(kb * NP + p) * 64];
    };
    const float* pa = group_ptr(g0);
    const uint4* pw = wf + (long)g0 * 5 * NP * 64;
#pragma unroll
    for (int kb = 0; kb < PF; ++kb) request(pa, pw, kb, W[kb], R[kb]);
    mid();
    cut8<AR>(R[0], sa, acur);
#pragma unroll 1
    for (int g = g0; g <= glast; ++g) {
        const int gn = g + 1 <= glast ? g + 1 : glast;
        const float* pan = group_ptr(gn);
        const uint4* pwn = wf + (long)gn * 5 * NP * 64;
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) {
            if (kb + PF < 5) request(pa, pw, kb + PF, W[kb + PF], R[kb + PF]);
            else request(pan, pwn, kb + PF - 5, W[kb + PF - 5], R[kb + PF - 5]);  // (after the last group: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            acc = mac<AR>(acur, W[kb], acc);
            cut8<AR>(R[(kb + 1) % 5], sa, acur);
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = pan; pw = pwn;
    }
}

// 32-channel form of strip_taps_k2: six steps (dt, k-block) per group = six buffers, look-ahead 3
template <class AR, typename Mid>
__device__ __forceinline__ void strip_taps_c2(const StripArgs& a, const float* lds, int base0, int base1, int base2, int g0, int ng,
                                              const uint4* __restrict__ wf, float sa, f32x16& acc, Mid mid)
{
    constexpr int NP = AR::NP, PF = 3, CP = 33;
    const int rowstep = a.Tp * CP;
    const int glast = g0 + ng - 1;
    auto group_ptr = [&](int g) -> const float* {
        const int dh = g / 3, dw = g - 3 * dh;                                  // wave-uniform
        const int b = dh == 0 ? base0 : (dh == 1 ? base1 : base2);
        return lds + b + dw * rowstep;
    };
    float R[6][8];
    Frag W[6][NP], acur[NP];
    auto request = [&](const float* pa, const uint4* pw, int st, Frag (&w)[NP], float (&r)[8]) {   // st = dt * 2 + kb: compile-time
        const int dt = st >> 1, kb = st & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = pa[dt * CP + 16 * kb + j];
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p].u = pw[(st * NP + p) * 64];
    };
    const float* pa = group_ptr(g0);
    const uint4* pw = wf + (long)g0 * 6 * NP * 64;
#pragma unroll
    for (int st = 0; st < PF; ++st) request(pa, pw, st, W[st], R[st]);
    mid();
    cut8<AR>(R[0], sa, acur);
#pragma unroll 1
    for (int g = g0; g <= glast; ++g) {
        const int gn = g + 1 <= glast ? g + 1 : glast;
        const float* pan = group_ptr(gn);
        const uint4* pwn = wf + (long)gn * 6 * NP * 64;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            if (st + PF < 6) request(pa, pw, st + PF, W[st + PF], R[st + PF]);
            else request(pan, pwn, st + PF - 6, W[st + PF - 6], R[st + PF - 6]);  // (after the last group: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            acc = mac<AR>(acur, W[st], acc);
            cut8<AR>(R[(st + 1) % 6], sa, acur);
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = pan; pw = pwn;
    }
}

template <int CC, int KS, bool GATE, int STRIP_SLOTS, int AM>            // AM / am: see conv3_mfma_kernel
__global__ __launch_bounds__(512, 2) void conv3_strip_kernel(StripArgs a, const float* __restrict__ x, const float* __restrict__ gate,
                                                            const float4* __restrict__ wfrag, const float* __restrict__ bias,
                                                            const float* __restrict__ skip, float* __restrict__ y, Amax am)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool X6 = AM != 0;
    using AR = std::conditional_t<AM == 2, H3, probav::X6>;
    float sa = 1.f; int eun = 0; float omax = 0.f;
    if constexpr (AM == 2) {   // H3 scales: this patch's activations and, per lane, the filter column the lane accumulates
        const int ea = h3_exp(am.x[blockIdx.x / a.nstrips]), ew = h3_exp_w(am.w[(threadIdx.x & 31) < a.g.Cout ? (threadIdx.x & 31) : 0]);
        sa = pow2i(ea); eun = -(ea + ew);
    }
    constexpr int CP = (CC & 1) ? CC : CC + 1;
    constexpr int KS4 = (KS + 3) / 4;
    constexpr int V = (CC % 4 == 0) ? 4 : 1;        // floats per load when staging
    constexpr int CG = CC / V;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                   // wave-uniform: keep tap arithmetic on the scalar unit
    const int tsel = wave & 3, grp = wave >> 2;     // tile of the round, tap group (0: taps 0..13, 1: taps 14..26)
    const int rowfloats = a.Wp * a.Tp * CP;
    float* part = lds + STRIP_SLOTS * rowfloats + 8; // [4 tiles][16 regs][64 lanes]
    const int n = blockIdx.x / a.nstrips, strip = blockIdx.x - n * a.nstrips;
    const int hb = strip * a.SR;
    const int SRr = g.Ho - hb < a.SR ? g.Ho - hb : a.SR;
    const int nvr = g.Wo * g.To;                     // voxels per output row (>= 128)
    const int NV = SRr * nvr, NTL = (NV + 31) >> 5, nrounds = (NTL + 3) >> 2;
    const int nchunk = (g.Cin + CC - 1) / CC;
    const long out_base = ((long)n * g.Ho + hb) * nvr;
    float* ybase = y + out_base * g.Cout;
    const float* sbase = skip ? skip + out_base * g.Cout : nullptr;
    const int srcE = g.Wi * g.Ti * CG;               // staged loads per input row
    constexpr int RV = (V == 4) ? 4 : 10;            // staged loads per thread: ceil(srcE / 512) must be <= RV
    typedef typename std::conditional<V == 4, float4, float>::type stage_t;
    const float bv = (bias && col < g.Cout) ? bias[col] : 0.f;

    // ring row q <-> input row ih = hb - ph + q, slot q % 5.  stage(): row interior only; the pads of every slot are
    // zeroed once and never written again; rows outside the patch get a zero interior.
    auto row_src = [&](int q, int c0, const float*& xrow, const float*& grow) -> bool {
        const int ih = hb - g.ph + q;
        const bool ok = ih >= 0 && ih < g.Hi;
        const long rbase = (((long)n * g.Hi + (ok ? ih : 0)) * g.Wi) * (long)g.Ti * g.Cin + c0;
        xrow = x + rbase;
        grow = GATE ? gate + rbase : nullptr;
        return ok;
    };
    auto elem = [&](int j, int c0, int& so, int& d, bool& chan_ok) {           // j-th staged load of a row -> offsets
        chan_ok = true;
        if constexpr (V == 1 && CC == CP && (CC == 25 || CC == 1)) {
            const int w = fdiv(j, g.Ti * CC, a.mSrcCol);
            so = j;
            d = j + w * (a.Tp - g.Ti) * CP;
        } else {
            const int vs = j / CG, cgi = j - vs * CG;
            const int w = fdiv(vs, g.Ti, a.mTi);
            chan_ok = c0 + cgi * V < g.Cin;
            so = chan_ok ? vs * g.Cin + cgi * V : 0;
            d = (vs + w * (a.Tp - g.Ti)) * CP + cgi * V;
        }
    };
    auto stage_load = [&](int q, int c0, stage_t (&rv)[RV]) {
        const float *xrow, *grow;
        const bool rok = row_src(q, c0, xrow, grow);
#pragma unroll
        for (int k = 0; k < RV; ++k) {
            const int ju = tid + k * 512;
            const bool live = ju < srcE;
            int so, d; bool cok;
            elem(live ? ju : 0, c0, so, d, cok);
            const bool ok = live && rok && cok;
            if constexpr (V == 4) {
                float4 v = *reinterpret_cast<const float4*>(xrow + so);
                if constexpr (GATE) {
                    const float4 m = *reinterpret_cast<const float4*>(grow + so);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
                rv[k].x = ok ? v.x : 0.f; rv[k].y = ok ? v.y : 0.f; rv[k].z = ok ? v.z : 0.f; rv[k].w = ok ? v.w : 0.f;
            } else {
                float v = xrow[so];
                if constexpr (GATE) v = grow[so] > 0.f ? v : 0.f;
                rv[k] = ok ? v : 0.f;
            }
        }
    };
    auto stage_store = [&](int q, int c0, const stage_t (&rv)[RV]) {
        float* slot = lds + (q % STRIP_SLOTS) * rowfloats + (g.pw * a.Tp + g.pt) * CP;
        const int dead = STRIP_SLOTS * rowfloats;                                // slack word behind the ring
#pragma unroll
        for (int k = 0; k < RV; ++k) {
            const int ju = tid + k * 512;
            const bool live = ju < srcE;
            int so, d; bool cok;
            elem(live ? ju : 0, c0, so, d, cok);
            float* dst = live ? slot + d : lds + dead;
            if constexpr (V == 4) { dst[0] = rv[k].x; dst[live ? 1 : 0] = rv[k].y; dst[live ? 2 : 0] = rv[k].z; dst[live ? 3 : 0] = rv[k].w; }
            else dst[0] = rv[k];
        }
    };

    XS_DECL;
    for (int pass = 0; pass < nchunk; ++pass) {
        const int c0 = pass * CC;
        const int cv = g.Cin - c0 < CC ? g.Cin - c0 : CC, ks = (cv + 1) >> 1;
        const float4* wf = wfrag + (long)pass * 27 * KS4 * 64 + lane;
        __syncthreads();
        {   // zero the whole ring (pads stay zero for the rest of the pass), then rows q = 0..3
            float4* z = reinterpret_cast<float4*>(lds);
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = tid; i < (STRIP_SLOTS * rowfloats + 8) / 4; i += 512) z[i] = zero;
        }
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            stage_t rv[RV];
            stage_load(q, c0, rv);
            stage_store(q, c0, rv);
        }
        int hiq = 3;
        __syncthreads();
        XS_ACC(1);
        float skn[16];
        auto load_skip = [&](int tl) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int vi = tl * 32 + rowmap(i, half);
                const int o = (tl < NTL && vi < NV && col < g.Cout) ? vi * g.Cout + col : 0;
                skn[i] = sbase[o];
            }
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) skn[i] = 0.f;
        if (grp == 0 && sbase) load_skip(tsel);

        for (int r = 0; r < nrounds; ++r) {
            // does round r+1 need a row that is not resident yet?  (at most one new row per round: 128 <= voxels per row)
            const int vlast_next = (r + 2) * 128 - 1 < NV - 1 ? (r + 2) * 128 - 1 : NV - 1;
            const int need_next = fdiv(vlast_next, nvr, a.mNvr) + 2;
            const bool do_load = r + 1 < nrounds && need_next > hiq;                 // wave-uniform
            constexpr bool ASYNC = STRIP_SLOTS >= 5;                                 // a spare slot lets the load overlap the MFMAs
            stage_t rv[RV];
            auto mid = [&]() { if (ASYNC && do_load) stage_load(hiq + 1, c0, rv); };   // requested inside the tap loop

            const int tile = 4 * r + tsel;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            // skip-connection values of this wave's output rows: requested a whole round ahead, AFTER the round's last filter
            // load -- vmcnt retires in order, so a request placed in front of the tap loop made the first tap wait for HBM
            // (single-pass layers only: strip_plan() never yields nchunk > 1)
            float sk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sk[i] = skn[i];
            if (tile < NTL) {
                int vi = tile * 32 + col;
                vi = vi < NV ? vi : NV - 1;
                const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
                const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
                const int voff = (w * a.Tp + t) * CP + (X6 ? 8 * half : half);
                const int base0 = (hrel % STRIP_SLOTS) * rowfloats + voff, base1 = ((hrel + 1) % STRIP_SLOTS) * rowfloats + voff,
                          base2 = ((hrel + 2) % STRIP_SLOTS) * rowfloats + voff;
                // (H3: the first-dispatched half -- which also writes the tile -- takes 5 of the 9 groups: the younger half is the slower one per k-block)
                if constexpr (AM == 2 && CC == 25) strip_taps_k2<AR>(a, lds, base0, base1, base2, grp == 0 ? 0 : 5, grp == 0 ? 5 : 4, reinterpret_cast<const uint4*>(wfrag) + lane, sa, acc, mid);
                else if constexpr (X6 && CC == 25) strip_taps_x6k<AR>(a, lds, base0, base1, base2, grp == 0 ? 0 : 4, grp == 0 ? 4 : 5, reinterpret_cast<const uint4*>(wfrag) + lane, sa, acc, mid);   // (the epilogue wave takes 4 of the 9 groups)
                else if constexpr (AM == 2) strip_taps_c2<AR>(a, lds, base0, base1, base2, grp == 0 ? 0 : 5, grp == 0 ? 5 : 4, reinterpret_cast<const uint4*>(wfrag) + lane, sa, acc, mid);
                else if constexpr (X6) strip_taps_x6<CC, AR>(a, lds, base0, base1, base2, grp == 0 ? 0 : 4, grp == 0 ? 4 : 5, reinterpret_cast<const uint4*>(wfrag) + lane, sa, acc, mid);
                else strip_taps<CC, KS>(a, lds, base0, base1, base2, grp == 0 ? 0 : 14, grp == 0 ? 14 : 13, wf, acc, ks, mid);
                if (grp == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) part[(tsel * 16 + i) * 64 + lane] = acc[i];
                }
            }
            if (grp == 0 && sbase && r + 1 < nrounds) load_skip(4 * (r + 1) + tsel);
            XS_ACC(2);
            if (ASYNC && do_load) { stage_store(hiq + 1, c0, rv); ++hiq; }
            XS_ACC(3);
            __syncthreads();                                   // partials (+ the new row) are in LDS
            XS_ACC(4);
            float pv[16];
            if (grp == 0 && tile < NTL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[i] = part[(tsel * 16 + i) * 64 + lane];
            }
            if (!ASYNC && do_load) {                           // 4-slot ring: every wave is past its taps, the oldest row is dead
                stage_load(hiq + 1, c0, rv);
                stage_store(hiq + 1, c0, rv);
                ++hiq;
            }
            XS_ACC(3);
            __syncthreads();                                   // partial buffer may be rewritten by the next round
            XS_ACC(4);
            if (grp == 0 && tile < NTL) {
                const bool full = (tile * 32 + 32 <= NV) && g.Cout == 32;
                int oo[16];
                float ov[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int vi = tile * 32 + rowmap(i, half);
                    oo[i] = (vi < NV && col < g.Cout) ? vi * g.Cout + col : -1;
                    ov[i] = acc[i] + pv[i];
                    if constexpr (AM == 2) ov[i] = ldexpf(ov[i], eun);      // back from the operands' power-of-two scales
                }
                // single-pass layers (the only ones routed here): sk = skip connection.  Multi-pass: pass 0 stores the raw
                // partial, later passes add the previous partial (sk), the last one finishes with bias / ReLU / skip.
                if (nchunk == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v = ov[i] + bv;
                        if (g.relu) v = fmaxf(v, 0.f);
                        ov[i] = v + sk[i];
                    }
                } else {
                    if (pass > 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) ov[i] += sk[i];
                    }
                    if (pass == nchunk - 1) {
                        float s2[16];
                        if (sbase) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) s2[i] = sbase[oo[i] < 0 ? 0 : oo[i]];
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; ++i) s2[i] = 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float v = ov[i] + bv;
                            if (g.relu) v = fmaxf(v, 0.f);
                            ov[i] = v + s2[i];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) omax = fmaxf(omax, oo[i] >= 0 ? fabsf(ov[i]) : 0.f);
                if (full) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) ybase[oo[i]] = ov[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) if (oo[i] >= 0) ybase[oo[i]] = ov[i];
                }
            }
            XS_ACC(5);
        }
    }
    if (am.y) amax_commit(omax, am.y + n);
    XS_OUT;
}

struct StripPlan { bool ok; int CC, KS; size_t lds_bytes; int grid; StripArgs a; };

// ---------------------------------------------------------------------------------------------------
// Strip convolution with a ring of fp16 PIECE records (H3 arithmetic only).  Same decomposition as conv3_strip_kernel (strip of
// output rows per workgroup, flattened 32-voxel tiles, the taps of a tile split over a wave pair), but a staged row is cut ONCE,
// when it is written to LDS, instead of once per (dh, dw) group in the tap loop -- in the fp32 ring that cutting and the four
// ds_read2_b32 feeding it were ~25 % of the kernel (tools/diag_strip.hip: 135 -> 100 us with them stubbed out).
// Record of one voxel: 128 B = 8 chunks of 16 B, logical chunk p * 4 + cc = piece p of channels 8cc .. 8cc+7 (channels >= Cin are
// zero), stored at chunk (p * 4 + cc) ^ (voxel & 7): a wave's ds_read_b128 of one logical chunk of 32 consecutive voxels then
// touches all 32 banks evenly (a plain 128-byte stride would put every lane on the same four banks).  A k-block of 16 channels of
// one tap is one ds_read_b128 per piece; no VALU work in the loop beyond the record address.  Four slots (135 KB at 24 x 11 voxels),
// so the next row is REQUESTED under the taps but stored between the round's two barriers, when the oldest row is dead.
// ---------------------------------------------------------------------------------------------------
template <int CIN, bool GATE>
__global__ __launch_bounds__(512, 2) void conv3_pstrip_kernel(StripArgs a, const float* __restrict__ x, const float* __restrict__ gate,
                                                             const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ y, Amax am)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    using AR = H3;
    constexpr int NP = 2, SLOTS = 4, REC = 128, PF = 3;
    const ConvGeom& g = a.g;
    float omax = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    // H3 scales: this patch's activations (one slot per sample) and, per lane, the filter column the lane accumulates
    const int ea = h3_exp(am.x[blockIdx.x / (a.nstrips * a.nsplit)]), ew = h3_exp_w(am.w[col < g.Cout ? col : 0]);
    const float sa = pow2i(ea);
    const int eun = -(ea + ew);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tsel = wave & 3, grp = wave >> 2;
    const int rowbytes = a.Wp * a.Tp * REC;
    float* part = reinterpret_cast<float*>(plds + SLOTS * rowbytes);          // [4 tiles][16 regs][64 lanes]
    // workgroup = (patch n, strip of output rows, column range sp): output columns ws0 .. ws0 + Wt - 1 (pstrip_plan(): Wo % nsplit == 0)
    const int per = a.nstrips * a.nsplit;
    const int n = blockIdx.x / per, srem = blockIdx.x - n * per;
    const int strip = srem / a.nsplit, sp = srem - strip * a.nsplit;
    const int ws0 = sp * a.Wt;
    const int hb = strip * a.SR;
    const int SRr = g.Ho - hb < a.SR ? g.Ho - hb : a.SR;
    const int nvr = a.Wt * g.To;                                             // voxels per output row of this column range (>= 128)
    const int NV = SRr * nvr, NTL = (NV + 31) >> 5, nrounds = (NTL + 3) >> 2;
    const long out_base = ((long)n * g.Ho + hb) * g.Wo * g.To;
    float* ybase = y + out_base * g.Cout;
    const float* sbase = skip ? skip + out_base * g.Cout : nullptr;
    const float bv = (bias && col < g.Cout) ? bias[col] : 0.f;
    // offset of strip voxel vi (this lane's channel) from ybase / sbase: rows are contiguous only when the range is the whole row
    auto elem_off_ch = [&](int vi, int ch) -> int {
        if (a.nsplit == 1) return vi * g.Cout + ch;
        const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
        const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
        return ((hrel * g.Wo + ws0 + w) * g.To + t) * g.Cout + ch;
    };

    // staging: item i of a row = (local voxel i >> 2 = (local column lw, depth t), channel chunk i & 3); local column lw <-> input column
    // ws0 + lw - pw (zero outside the patch), lw < Wt + 2; RVP items per thread
    // (a whole-row range stages the patch columns only: its pad columns stay zero from the initial clear)
    const int lw0 = a.nsplit == 1 ? g.pw : 0, Wl = a.nsplit == 1 ? g.Wi : a.Wt + 2;
    const int items = Wl * g.Ti * 4;
    constexpr int RVP = 2;                                                   // pstrip_plan(): (Wt + 2) * Ti * 4 <= 512 * RVP
    auto stage_load = [&](int q, float (&v)[RVP][8]) {
        const int ih = hb - g.ph + q;
        const bool rok = ih >= 0 && ih < g.Hi;
        const long rbase = (((long)n * g.Hi + (rok ? ih : 0)) * g.Wi) * (long)g.Ti * CIN;
        const float* xrow = x + rbase;
        const float* grow = GATE ? gate + rbase : nullptr;
#pragma unroll
        for (int k = 0; k < RVP; ++k) {
            const int i = tid + 512 * k;
            const int ic = i < items ? i : 0;
            const int lvox = ic >> 2, cc = ic & 3;
            const int lwr = fdiv(lvox, g.Ti, a.mTi), t = lvox - lwr * g.Ti, lw = lw0 + lwr;
            const int iw = ws0 + lw - g.pw;
            const bool live = rok && i < items && iw >= 0 && iw < g.Wi;
            const int vox = (live ? iw : 0) * g.Ti + t;                      // input voxel of the row (clamped when dead)
            const float* src = xrow + vox * CIN + 8 * cc;
            const float* gsr = GATE ? grow + vox * CIN + 8 * cc : nullptr;
            if constexpr (CIN % 8 == 0) {
                const float4 t0 = reinterpret_cast<const float4*>(src)[0], t1 = reinterpret_cast<const float4*>(src)[1];
                float f[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                if constexpr (GATE) {
                    const float4 m0 = reinterpret_cast<const float4*>(gsr)[0], m1 = reinterpret_cast<const float4*>(gsr)[1];
                    const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = m[j] > 0.f ? f[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = live ? f[j] : 0.f;
            } else {
                // 25 channels: chunks 0..2 = channels 0..23 of the voxel; chunk 3 of the record at PADDED depth t' (the item's `vox`
                // is then (w, t')) gathers channel 24 of padded depths t', t'+1, t'+2 = input depths t'-1, t', t'+1 (pt = 1): the
                // tenth K chunk of a (dh, dw) group, so that a group is 5 k-blocks instead of 6
                // (integer selects only: with short-circuit conditions hipcc splits the lanes into two branches, each with its own load and a full vmcnt(0))
                const int is3 = cc == 3 ? 1 : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tj = t - 1 + j;                                 // (cc == 3 only: the gathered channel-24 chunk)
                    const int cok3 = j < 3 ? ((tj >= 0 ? 1 : 0) & (tj < g.Ti ? 1 : 0)) : 0;
                    const int o3 = cok3 ? (j - 1) * CIN + 24 : 24;            // (clamped to the voxel's own channel 24 when dead)
                    const int o = is3 ? o3 : 8 * cc + j;
                    float f = (xrow + vox * CIN)[o];
                    if constexpr (GATE) f = (grow + vox * CIN)[o] > 0.f ? f : 0.f;
                    v[k][j] = ((live ? 1 : 0) & ((1 - is3) | cok3)) ? f : 0.f;
                }
            }
        }
    };
    auto stage_store = [&](int q, const float (&v)[RVP][8]) {
        unsigned char* slot = plds + (q % SLOTS) * rowbytes;
#pragma unroll
        for (int k = 0; k < RVP; ++k) {
            const int i = tid + 512 * k;
            if (i < items) {
                const int lvox = i >> 2, cc = i & 3;
                const int lwr = fdiv(lvox, g.Ti, a.mTi), t = lvox - lwr * g.Ti, lw = lw0 + lwr;
                const int vd = lw * a.Tp + t + ((CIN == 25 && cc == 3) ? 0 : g.pt);   // (the gathered chunk is indexed by padded depth)
                Frag f[NP];
                cut8<AR>(v[k], sa, f);
                unsigned char* rec = slot + vd * REC;
                const int sw = (int)((rec - plds) >> 7) & 7;                 // swizzle = absolute record index & 7 (the readers use the same)
                *reinterpret_cast<uint4*>(rec + ((cc ^ sw) << 4)) = f[0].u;
                *reinterpret_cast<uint4*>(rec + (((4 + cc) ^ sw) << 4)) = f[1].u;
            }
        }
    };

    {   // zero the whole ring (pads stay zero), then rows q = 0..3
        uint4* z = reinterpret_cast<uint4*>(plds);
        for (int i = tid; i < SLOTS * rowbytes / 16; i += 512) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        float v[RVP][8];
        stage_load(q, v);
        stage_store(q, v);
    }
    int hiq = 3;
    __syncthreads();
    XS_DECL;
    XS_ACC(1);

    // Epilogue layout: after the two halves of a tile have met, the writing wave turns the tile around inside its 4 KB slot of the
    // exchange buffer ([voxel][32 channels]) and every lane ends up with FOUR CONSECUTIVE CHANNELS of four voxels: lane l <-> voxel rows
    // (l >> 3) + 8 jj, channels 4 (l & 7) .. + 3.  Skip tile and output then move as 16-byte accesses (4 instead of 16 per lane and
    // tile; dword stores are bound by store issue, ~7 B/clk/CU), and the skip loads are issued AFTER the tap loop, so that they no
    // longer sit in front of the filter fragments in the in-order vmcnt queue.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int eq = lane & 7, er = lane >> 3;                                // epilogue coordinates: channel quad, voxel row (mod 8)
    const uint4* wf = wfrag + lane;                                         // PACK_H3_CONV (32 channels) / PACK_H3_CONVP (25): fragment ((group * NST + st) * NP + piece) * 64 + lane
    for (int r = 0; r < nrounds; ++r) {
        const int vlast_next = (r + 2) * 128 - 1 < NV - 1 ? (r + 2) * 128 - 1 : NV - 1;
        const int need_next = fdiv(vlast_next, nvr, a.mNvr) + 2;
        const bool do_load = r + 1 < nrounds && need_next > hiq;                 // wave-uniform
        float nv_[RVP][8];
        const int tile = 4 * r + tsel;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (tile < NTL) {
            int vi = tile * 32 + col;
            vi = vi < NV ? vi : NV - 1;
            const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
            const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
            const int vox0 = w * a.Tp + t;                                   // record index of tap (dw, dt) = (0, 0) inside a slot
            const int sb0 = (hrel % SLOTS) * rowbytes, sb1 = ((hrel + 1) % SLOTS) * rowbytes, sb2 = ((hrel + 2) % SLOTS) * rowbytes;
            const int g0 = grp == 0 ? 0 : 5, glast = grp == 0 ? 4 : 8;      // the first-dispatched half takes 5 of the 9 (dh, dw) groups (4 : 5 measures the same)
            // operand address of step st = dt * 2 + kb of group gg: record vox0 + dw * Tp + dt of ring row hrel + dh, logical chunks 2 kb + half (+4)
            auto rec_addr = [&](int gg, int dt) -> int {
                const int dh = gg / 3, dw = gg - 3 * dh;                     // wave-uniform
                const int sb = dh == 0 ? sb0 : (dh == 1 ? sb1 : sb2);
                return sb + (vox0 + dw * a.Tp + dt) * REC;
            };
            constexpr int NST = CIN == 25 ? 5 : 6;                           // k-blocks per (dh, dw) group
            Frag A[NST][NP], W[NST][NP];
            auto request = [&](int gg, int st, Frag (&af)[NP], Frag (&wq)[NP]) {   // st: compile-time
                int ra, cc;
                if constexpr (CIN == 25) {
                    // chunk c = 2 st + half of the group's ten: c < 9 -> (dt, cc) = (c / 3, c % 3); c = 9 -> the gathered channel-24 chunk of dt = 0
                    const int c0 = 2 * st, c1 = 2 * st + 1;                 // (constants after unrolling)
                    const int dt0 = c0 / 3, cc0 = c0 % 3, dt1 = c1 < 9 ? c1 / 3 : 0, cc1 = c1 < 9 ? c1 % 3 : 3;
                    ra = rec_addr(gg, 0) + (half ? dt1 : dt0) * REC;
                    cc = half ? cc1 : cc0;
                } else {
                    const int dt = st >> 1, kb = st & 1;
                    ra = rec_addr(gg, dt);
                    cc = 2 * kb + half;
                }
                const int sw = (ra >> 7) & 7;                                // absolute record index & 7, as stored
                const int cp = cc ^ sw;
                af[0].u = *reinterpret_cast<const uint4*>(plds + ra + (cp << 4));
                af[1].u = *reinterpret_cast<const uint4*>(plds + ra + ((cp ^ 4) << 4));
                const uint4* pw = wf + ((long)gg * NST + st) * NP * 64;
                wq[0].u = pw[0]; wq[1].u = pw[64];
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) request(g0, st, A[st], W[st]);
            if (do_load) stage_load(hiq + 1, nv_);                           // in flight during the taps
#pragma unroll 1
            for (int gg = g0; gg <= glast; ++gg) {
                const int gn = gg + 1 <= glast ? gg + 1 : glast;
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    if (st + PF < NST) request(gg, st + PF, A[st + PF], W[st + PF]);
                    else request(gn, st + PF - NST, A[st + PF - NST], W[st + PF - NST]);   // (after the last group: a harmless re-read)
                    __builtin_amdgcn_sched_barrier(0);
                    acc = mac<AR>(A[st], W[st], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (grp == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) part[(tsel * 16 + i) * 64 + lane] = acc[i];
            }
        } else if (do_load) stage_load(hiq + 1, nv_);
        // The two waves of a pair SHARE the epilogue of their tile: the first one writes voxel rows er + 0, er + 8, the second one rows
        // er + 16, er + 24.  Each loads its half of the skip tile now (16-byte rows, consumed after the second barrier).
        f32x4u skq[2];
        int eoff[2];                                                         // element offset of (row er + 8 (2 grp + jj), channel 4 eq) from ybase / sbase, -1 = dead row
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int vi = tile * 32 + er + 8 * (2 * grp + jj);
            eoff[jj] = (tile < NTL && vi < NV) ? elem_off_ch(vi, 4 * eq) : -1;
            skq[jj] = (f32x4u){0.f, 0.f, 0.f, 0.f};
        }
        if (sbase && tile < NTL) {
            if (g.Cout == 32) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) skq[jj] = *reinterpret_cast<const f32x4u*>(sbase + (eoff[jj] < 0 ? 0 : eoff[jj]));
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int c = 0; c < 4; ++c) skq[jj][c] = sbase[(eoff[jj] < 0 || 4 * eq + c >= g.Cout) ? 0 : eoff[jj] + c];
            }
        }
        XS_ACC(2);
        __syncthreads();                                   // partials are in LDS; every wave is past its taps
        XS_ACC(4);
        // Between the two barriers the first wave of a pair turns the tile around inside the tile's own 4 KB slot of the exchange buffer (only
        // this wave touches the slot in this interval: the partner wrote it before the first barrier) and takes its two rows; the partner
        // reads ITS two rows after the second barrier, before it can write the slot again (program order): no access is left to chance.
        float* slot = part + tsel * 16 * 64;
        float4 tq[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) tq[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp == 0 && tile < NTL) {
            float fv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) fv[i] = part[(tsel * 16 + i) * 64 + lane];                // the partner's partial sums
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     // (all of them read before the slot is rewritten)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = ldexpf(acc[i] + fv[i], eun) + bv;                   // final values in the accumulator layout: the filter column's exponent and the bias are per lane here
                if (g.relu) v = fmaxf(v, 0.f);
                slot[rowmap(i, half) * 32 + col] = v;                         // [voxel][channel]
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (a wave's LDS operations execute in order: the wait orders the compiler)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) tq[jj] = *reinterpret_cast<const float4*>(slot + (er + 8 * jj) * 32 + 4 * eq);
        }
        if (do_load) { stage_store(hiq + 1, nv_); ++hiq; }  // replaces the oldest row, which no tile of the next round reads
        XS_ACC(3);
        __syncthreads();                                   // the turned tiles and the new row are visible
        XS_ACC(4);
        if (tile < NTL) {
            if (grp == 1) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) tq[jj] = *reinterpret_cast<const float4*>(slot + (er + 8 * (2 + jj)) * 32 + 4 * eq);
            }
            const bool full = (tile * 32 + 32 <= NV) && g.Cout == 32;            // wave-uniform
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4 t = tq[jj];
                f32x4u o = {t.x + skq[jj][0], t.y + skq[jj][1], t.z + skq[jj][2], t.w + skq[jj][3]};
                if (full) {
                    *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                } else if (eoff[jj] >= 0) {
                    if (4 * eq + 4 <= g.Cout) {
                        *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (4 * eq + c < g.Cout) { ybase[eoff[jj] + c] = o[c]; omax = fmaxf(omax, fabsf(o[c])); }
                    }
                }
            }
        }
        XS_ACC(5);
    }
    if (am.y) amax_commit(omax, am.y + n);
    XS_OUT;
}

// ---------------------------------------------------------------------------------------------------
// Piece-ring strip convolution with ALTERNATING HALVES (H3 arithmetic).  conv3_pstrip_kernel runs its eight waves through the same
// program: the two waves of a SIMD go through their taps together and through their epilogues together, so the matrix pipe idles
// during every epilogue, staging and pipeline refill (stamps: taps 70 % of a wave's life, and the pipe is busy for 45 % of that).
// Here the halves of the workgroup (waves 0-3 / 4-7: one wave of each on every SIMD) take turns: in segment s half (s & 1) runs
// the taps of four tiles -- one WHOLE tile per wave, no split of the taps over a wave pair and therefore no exchange of partial sums --
// while the other half finishes the tiles it computed in segment s - 1 (scale, bias, ReLU, turn-around through its own 4 KB of
// LDS, skip, 16-byte stores) and stages the input rows of segment s + 1, its own next taps.  One barrier per segment; on every SIMD
// a tap-loop wave always runs beside a wave doing memory and vector work.
// Ring: `nslot` rows (pp_plan(): the rows of two consecutive segments never collide); records, swizzle, filter fragments and the
// gathered channel-24 chunk are those of conv3_pstrip_kernel.
// ---------------------------------------------------------------------------------------------------
template <int CIN, bool GATE, int RVP, bool K32 = false>
__global__ __launch_bounds__(512, 2) void conv3_pp_kernel(StripArgs a, const float* __restrict__ x, const float* __restrict__ gate,
                                                         const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                         const float* __restrict__ skip, float* __restrict__ y, Amax am)
{
    XS_ENTRY;
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    using AR = H3;
    constexpr int NP = 2, REC = 128, PF = 3;
    // K32 (the fourth template argument) = the NEW FORM: planar ring, the two waves of a pair split a 64-voxel tile's k-blocks.  With 32 input channels it
    // also means the 16x16x32 MFMA shape (S16); with 25 (P25) the k-blocks stay v_mfma_f32_32x32x16_f16 ones (75 k-slots per (dh, dw) group = 5 of 16).
    constexpr bool S16 = K32 && CIN == 32, P25 = K32 && CIN == 25;
    const ConvGeom& g = a.g;
    float omax = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int gtid = tid & 255;                                              // thread of its half (staging index)
    const int per = a.nstrips * a.nsplit;
    const int n = blockIdx.x / per, srem = blockIdx.x - n * per;
    const int ea = h3_exp(am.x[n]), ew = h3_exp_w(am.w[col < g.Cout ? col : 0]);
    const float sa = pow2i(ea);
    const int eun = -(ea + ew);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tsel = wave & 3, grp = wave >> 2;
    // Ring row.  32x32x16 forms: Wp * Tp records of 128 bytes ([piece][chunk of 8 channels], swizzled: stage_store).  K32 form: PLANAR -- eight planes
    // (piece, chunk) of Wp * Tp 16-byte entries each, plane pitch PS = 16 * (a count that is 2 mod 16): a tap is then a constant byte offset from the lane's
    // (dw, dt) = (0, 0) entry -- no swizzle key that changes with the tap, no per-tap address arithmetic -- and the banks work out by parity: entry e of plane c
    // sits in bank group (e + 2 c) mod 16; a ds_read_b128 lane group holds eight lanes of chunk kq and eight of kq + 1 (MI355X_MICROARCH.md, LDS), the kernel
    // gives the former the tile's even voxels and the latter the odd ones (pm16 below), consecutive voxels alternate entry parity (Tp - 2 and Tp differ by 2), so
    // the two sets fall into bank groups of different parity whatever the tap; the staging stores (two voxels x four chunks per lane group) spread the same way.
    const int NS = a.nslot;
    const int PS = S16 ? 16 * ((((a.Wp * a.Tp) + 13) & ~15) + 2) : 0;
    const int rowbytes = S16 ? 8 * PS : a.Wp * a.Tp * REC;
    float* turn = reinterpret_cast<float*>(plds + NS * rowbytes) + tsel * 1024;      // [32 voxels][32 channels], shared by waves tsel and tsel + 4 (never in the same role; K32: unused)
    const int strip = srem / a.nsplit, sp = srem - strip * a.nsplit;
    const int ws0 = sp * a.Wt;
    const int hb = strip * a.SR;
    const int SRr = g.Ho - hb < a.SR ? g.Ho - hb : a.SR;
    const int nvr = a.Wt * g.To;                                             // voxels per output row of this column range
    const int NV = SRr * nvr, NTL = (NV + 31) >> 5, nseg = (NTL + 3) >> 2;
    const long out_base = ((long)n * g.Ho + hb) * g.Wo * g.To;
    float* ybase = y + out_base * g.Cout;
    const float* sbase = skip ? skip + out_base * g.Cout : nullptr;
    const float bv = (bias && col < g.Cout) ? bias[col] : 0.f;
    // K32 form (v_mfma_f32_16x16x32_f16, 32 input channels = ONE k-block per tap): the 32 x 32 tile is four 16 x 16 accumulators (voxel half u,
    // channel half v), computed TRANSPOSED -- the filter fragment is the instruction's A operand, the records its B -- so that a lane's four accumulator
    // registers are four CONSECUTIVE channels 16 v + 4 kq .. + 3 of one voxel 16 u + pm16: skip and output are 16-byte accesses straight from / to
    // memory, with no turn-around through LDS.  The filter columns' exponents and the biases of the lane's eight channels:
    const int m16 = lane & 15, kq = lane >> 4;
    // voxel of the 16 of a sub-tile that operand row / column m16 stands for: the lanes that share a ds_read_b128 group with the NEXT chunk's lanes
    // (m16 = 4..11) take the odd voxels, the others the even ones (the planar ring's bank argument above)
    const int pm16 = (m16 >= 4 && m16 < 12) ? 2 * (m16 - 4) + 1 : (m16 < 4 ? 2 * m16 : 2 * (m16 - 12) + 8);
    int eun8[2][4];
    float bv8[2][4];
    if constexpr (S16) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * v + 4 * kq + i;
                eun8[v][i] = -(ea + h3_exp_w(am.w[c < g.Cout ? c : 0]));
                bv8[v][i] = (bias && c < g.Cout) ? bias[c] : 0.f;
            }
    }
    auto elem_off_ch = [&](int vi, int ch) -> int {
        if (a.nsplit == 1) return vi * g.Cout + ch;
        const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
        const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
        return ((hrel * g.Wo + ws0 + w) * g.To + t) * g.Cout + ch;
    };
    // last ring row (relative to the strip's first input row) that segment sg reads
    auto need = [&](int sg) -> int {
        const int vlast = (sg + 1) * 128 - 1 < NV - 1 ? (sg + 1) * 128 - 1 : NV - 1;
        return fdiv(vlast, nvr, a.mNvr) + 2;
    };

    // Staging by ONE half (256 threads).  A thread's items are the same for every row, so what does not depend on the row is worked out once:
    // item i <-> (local voxel lv = i / NCH, channel chunk cc = i % NCH) with NCH = 4 chunks of 8 channels (32 channels) or 3 (25 channels:
    // channels 0..23; the voxel's fourth chunk GATHERS channel 24 of padded depths t', t'+1, t'+2 and is one extra item per voxel, thread lv).
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NCH = CIN == 25 ? 3 : 4;
    // (one column range and zero pads: the pad columns are never written and stay zero from the clear; MIRRORED pads are data and are staged like a range's halo columns --
    //  until round 5 they were not when the row was ONE range, which no layer of the reference's networks is: 22 x 22 x 7 and x 5 inputs through the C ABI came out wrong)
    const bool own_cols = a.nsplit == 1 && !g.reflect_hw;
    const int lw0 = own_cols ? g.pw : 0, Wl = own_cols ? g.Wi : a.Wt + 2;
    const int nvs = Wl * g.Ti, items = nvs * NCH;                            // pp_plan(): items <= 256 * RVP, nvs <= 256
    struct Staged { float v[RVP][8]; float g3[3]; float m[GATE ? RVP : 1][8]; float m3[3]; };   // m, m3: the gate tensor's values (GATE), applied in stage_store
    int s_src[RVP], s_vd[RVP], s_cc[RVP], s_live[RVP];                      // source offset (floats) inside an input row, record, chunk, 1 = inside the patch's columns
    int s_key[RVP], s3_key = 0;                                              // the records' swizzle keys (stage_store)
    int s3_src = 0, s3_vd = 0, s3_m = 0;                                    // gathered chunk: offset of (voxel, channel 24), record, validity bits of depths t-1, t, t+1 (bit 3: item exists)
    {
        auto locate = [&](int lv, int& lw, int& t, int& iwc, int& colok) {
            const int lwr = fdiv(lv, g.Ti, a.mTi);
            t = lv - lwr * g.Ti; lw = lw0 + lwr;
            int iw = ws0 + lw - g.pw;
            if (g.reflect_hw) iw = iw < 0 ? -iw : (iw >= g.Wi ? 2 * g.Wi - 2 - iw : iw);     // tf.pad REFLECT: the H / W pads mirror the input (convReducer_1)
            colok = (iw >= 0 ? 1 : 0) & (iw < g.Wi ? 1 : 0);
            iwc = iw < 0 ? 0 : (iw < g.Wi ? iw : g.Wi - 1);
        };
#pragma unroll
        for (int k = 0; k < RVP; ++k) {
            const int it = gtid + 256 * k;
            const int ic = it < items ? it : 0;
            const int lv = CIN == 25 ? (int)(((unsigned)ic * 43691u) >> 17) : ic >> 2;      // ic / 3 (ic < 2^16)
            const int cc = ic - lv * NCH;
            int lw, t, iwc, colok;
            locate(lv, lw, t, iwc, colok);
            s_src[k] = (iwc * g.Ti + t) * CIN + 8 * cc;
            s_vd[k] = it < items ? lw * a.Tp + t + g.pt : -1;
            s_key[k] = ((lw * (a.Tp - 2) + t + g.pt) >> 1) & 7;
            s_cc[k] = cc;
            s_live[k] = colok;
        }
        if constexpr (CIN == 25) {
            const int lv = gtid < nvs ? gtid : 0;
            int lw, t, iwc, colok;
            locate(lv, lw, t, iwc, colok);
            s3_src = (iwc * g.Ti + t) * CIN + 24;
            s3_vd = lw * a.Tp + t;                                             // (indexed by PADDED depth t' = t: the chunk holds padded depths t', t'+1, t'+2 = input depths t-1, t, t+1)
            s3_key = ((lw * (a.Tp - 2) + t) >> 1) & 7;
            s3_m = (colok && t - 1 >= 0 ? 1 : 0) | (colok ? 2 : 0) | (colok && t + 1 < g.Ti ? 4 : 0) | (gtid < nvs ? 8 : 0);
        }
    }
    // stage_load only REQUESTS (clamped addresses, nothing consumed): what must be zero -- rows and columns outside the patch, the dead
    // depths of the gathered chunk -- is zeroed in stage_store, so that the loads stay in flight across the barrier
    auto stage_load = [&](int q, Staged& sv) {
        int ih = hb - g.ph + q;
        if (g.reflect_hw) ih = ih < 0 ? -ih : (ih >= g.Hi ? 2 * g.Hi - 2 - ih : ih);
        const bool rok = ih >= 0 && ih < g.Hi;
        const long rbase = (((long)n * g.Hi + (rok ? ih : 0)) * g.Wi) * (long)g.Ti * CIN;
        const float* xrow = x + rbase;
#pragma unroll
        for (int k = 0; k < RVP; ++k) {
            const f32x4u t0 = *reinterpret_cast<const f32x4u*>(xrow + s_src[k]), t1 = *reinterpret_cast<const f32x4u*>(xrow + s_src[k] + 4);
            const float f[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            if constexpr (GATE) {
                const float* grow = gate + rbase;
                const f32x4u m0 = *reinterpret_cast<const f32x4u*>(grow + s_src[k]), m1 = *reinterpret_cast<const f32x4u*>(grow + s_src[k] + 4);
                const float m[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) sv.m[k][j] = m[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sv.v[k][j] = f[j];
        }
        if constexpr (CIN == 25) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int o = s3_src + (((s3_m >> j) & 1) ? (j - 1) * CIN : 0);
                sv.g3[j] = xrow[o];
                if constexpr (GATE) sv.m3[j] = (gate + rbase)[o];
            }
        }
    };
    auto stage_store = [&](int q, Staged& sv) {
        const int sl = q - fdiv(q, NS, a.mNslot) * NS;
        unsigned char* slot = plds + sl * rowbytes;
        int ih = hb - g.ph + q;
        if (g.reflect_hw) ih = ih < 0 ? -ih : (ih >= g.Hi ? 2 * g.Hi - 2 - ih : ih);
        const int rok = (ih >= 0 && ih < g.Hi) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < RVP; ++k) {
            if (s_vd[k] >= 0) {
                const int live = rok & s_live[k];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bool keep = live != 0;
                    if constexpr (GATE) keep = keep && sv.m[k][j] > 0.f;     // ReLU mask of the producing layer (backward-data of a ReLU layer)
                    sv.v[k][j] = keep ? sv.v[k][j] : 0.f;
                }
                Frag f[NP];
                cut8<AR>(sv.v[k], sa, f);
                if constexpr (S16) {                                         // planar ring: entry s_vd of planes (piece, chunk)
                    unsigned char* ent = slot + s_vd[k] * 16 + s_cc[k] * PS;
                    *reinterpret_cast<uint4*>(ent) = f[0].u;
                    *reinterpret_cast<uint4*>(ent + 4 * PS) = f[1].u;
                    continue;
                }
                unsigned char* rec = slot + s_vd[k] * REC;
                // Swizzle key of a record at padded coordinates (w', t'): made of x = w' (Tp - 2) + t', NOT of the record index w' Tp + t'.  x runs on through
                // a tile's voxels (t fastest) where the record index jumps by 3 at every column change, and its parity is the record's (address bit 7), so the
                // sixteen lanes of a ds_read_b128 group -- voxels {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a tile reading ONE chunk (32x32x16 forms) --
                // get sixteen different bank groups from key = (x >> 1) & 7 on the three chunk bits.  (With the record index in x's place 62 % of the LDS
                // cycles of these kernels were bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; 11 % now, at tile rows that wrap.)
                const int sw = s_key[k];
                *reinterpret_cast<uint4*>(rec + ((s_cc[k] ^ sw) << 4)) = f[0].u;
                *reinterpret_cast<uint4*>(rec + (((4 + s_cc[k]) ^ sw) << 4)) = f[1].u;
            }
        }
        if constexpr (CIN == 25) {
            if (s3_m & 8) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bool keep = (rok & (s3_m >> j) & 1) != 0;
                    if constexpr (GATE) keep = keep && sv.m3[j] > 0.f;
                    v8[j] = keep ? sv.g3[j] : 0.f;
                }
                Frag f[NP];
                cut8<AR>(v8, sa, f);
                unsigned char* rec = slot + s3_vd * REC;
                const int sw = s3_key;
                *reinterpret_cast<uint4*>(rec + ((3 ^ sw) << 4)) = f[0].u;
                *reinterpret_cast<uint4*>(rec + ((7 ^ sw) << 4)) = f[1].u;
            }
        }
    };

    int hiq = nseg > 0 ? need(0) : -1;
    {   // zero the ring (pads stay zero), then the rows of segment 0: the halves take alternate rows; each half's first row is requested
        // before the clearing, so that one memory round trip hides under it
        Staged v0;
        if (grp <= hiq) stage_load(grp, v0);
        uint4* z = reinterpret_cast<uint4*>(plds);
        for (int i = tid; i < NS * rowbytes / 16; i += 512) z[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (grp <= hiq) stage_store(grp, v0);
    }
#pragma unroll 1
    for (int q = grp + 2; q <= hiq; q += 2) {
        Staged v;
        stage_load(q, v);
        stage_store(q, v);
    }
    __syncthreads();
    XS_DECL;
    XS_ACC(1);

    const int eq = lane & 7, er = lane >> 3;                                // epilogue coordinates: channel quad, voxel row (mod 8)
    const int cc4[2] = {half << 4, (2 + half) << 4};                        // 32 channels: k-block kb reads logical chunk 2 kb + half (its byte offset in an unswizzled record)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int NST = CIN == 25 ? 5 : 6;                                   // k-blocks per (dh, dw) group
    Frag A[NST][NP], W[NST][NP];
    // what a half requests at the END of its taps, so that it has landed when its finishing segment begins: the skip tile of the
    // tile just computed and the next input row (global loads only; the barrier between the segments waits for LDS traffic alone)
    Staged nv_;
    f32x4u skq[4];
    int eoff[4];
    bool have_nv = false;
    auto load_skip = [&](int tile) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int vi = S16 ? tile * 32 + 16 * (jj >> 1) + pm16 : (P25 ? tile * 32 + col : tile * 32 + er + 8 * jj);      // S16: jj = 2 u + v; P25: jj = channel group g
            eoff[jj] = vi < NV ? elem_off_ch(vi, S16 ? 16 * (jj & 1) + 4 * kq : (P25 ? 8 * jj + 4 * half : 4 * eq)) : -1;
            skq[jj] = (f32x4u){0.f, 0.f, 0.f, 0.f};
        }
        if (sbase) {
            if (g.Cout == 32) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) skq[jj] = *reinterpret_cast<const f32x4u*>(sbase + (eoff[jj] < 0 ? 0 : eoff[jj]));
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int c = 0; c < 4; ++c) skq[jj][c] = sbase[(eoff[jj] < 0 || (S16 ? 16 * (jj & 1) + 4 * kq : (P25 ? 8 * jj + 4 * half : 4 * eq)) + c >= g.Cout) ? 0 : eoff[jj] + c];
            }
        }
    };
    auto request_W = [&](int gg, int st, Frag (&wq)[NP]) {
        const uint4* pw = wfrag + ((long)gg * NST + st) * NP * 64;      // wave-uniform base + lane: scalar-base loads
        wq[0].u = pw[lane]; wq[1].u = pw[64 + lane];
    };
    // K32 form: the fragments of PACK_H3_CONV are read with another lane order.  A 16x16x32 B operand wants, in lane (m16, kq), k = 8 kq .. 8 kq + 7 of
    // column 16 v + m16: that is lane (16 v + m16) + 32 (kq & 1) of the 32x32x16 fragment kb = kq >> 1 -- the same 16 bytes, no second packed copy.
    // (The kernel issues it as the A operand: the tile is computed transposed, see eun8 above.)  Rings: WK[2 (tap % WD) + v] here, A[2 (tap % 3) + u] for the records.
    const int wl16 = ((kq >> 1) * 128 + 32 * (kq & 1) + m16) << 4;          // byte offset inside a tap's 4 KB (piece: + 1024, channel half v: + 256)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wfrag), 0, 27 * 4096, 0x00020000);   // (K32: scalar base + 32-bit lane offset + immediate)
    typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
    auto wload16 = [&](int tap, int off) -> uint4 {
        const u32x4b r = __builtin_amdgcn_raw_buffer_load_b128(wrs, wl16 + off, tap * 4096, 0);
        return make_uint4(r[0], r[1], r[2], r[3]);
    };
#ifdef PPX_WD
    constexpr int WD = PPX_WD;
#else
    constexpr int WD = 2;                                                    // depth of the K32 form's filter ring (taps of 24 MFMAs; the L1 is bandwidth-, not latency-bound)
#endif
    // K32: the two waves of a PAIR (tsel >> 1) share a 64-voxel tile and split its TAPS -- member 0 taps 0..13, member 1 taps 14..26 -- so that a tap's
    // 4 KB of filter fragments feed 24 MFMAs instead of 12: four waves x 4 KB per 12-MFMA tap are exactly the 64 B/clk of the vector L1 and made that tap
    // 256 cycles long instead of 192 (tools/mfma_feed.hip, docs/notebook_r1-r5.md 4.1f).  Each member ends with partial sums of all 64 voxels, keeps the 32 it finishes
    // and hands the other 32 to its partner through LDS (xbuf: the 16 KB behind the ring; second barrier at the end of the segment).
    constexpr int KT0 = 14;                                                  // first tap of member 1
#ifdef PPX_AD
    constexpr int AD = PPX_AD;
#else
    constexpr int AD = 2;                                                    // depth of the K32 form's record ring (taps of 24 MFMAs)
#endif
    const int pr = tsel >> 1, tr = tsel & 1;
    Frag WK[S16 ? 2 * WD : 1][NP], AK[S16 ? 4 * AD : 1][NP];                 // WK[2 (L % WD) + v], AK[4 (L % AD) + u], L = tap - the member's first tap
    // P25: 45 k-blocks of 16 (nine (dh, dw) groups x five), member 0 k-blocks 0..22, member 1 23..44; rings of RK k-blocks: WP[L % RK], AP[2 (L % RK) + voxel half]
    constexpr int KB0 = 23, RK = 3;
    Frag WP[P25 ? RK : 1][NP], AP[P25 ? 2 * RK : 1][NP];
    const __amdgpu_buffer_rsrc_t wrp = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wfrag), 0, 9 * 5 * NP * 1024, 0x00020000);
    auto request_WP = [&](int K, int sl) {                                   // k-block K = 5 group + st: the fragments' own order; scalar base + lane offset (no 64-bit address per k-block)
        const u32x4b r0 = __builtin_amdgcn_raw_buffer_load_b128(wrp, lane * 16, K * (NP * 1024), 0);
        const u32x4b r1 = __builtin_amdgcn_raw_buffer_load_b128(wrp, lane * 16 + 1024, K * (NP * 1024), 0);
        WP[sl][0].u = make_uint4(r0[0], r0[1], r0[2], r0[3]); WP[sl][1].u = make_uint4(r1[0], r1[1], r1[2], r1[3]);
    };
    auto request_WP_first = [&]() {
        const int K0 = tr ? KB0 : 0;
#pragma unroll
        for (int L = 0; L < RK - 1; ++L) request_WP(K0 + L, L);
    };
    f32x16 keep16, send16;                                                   // P25: partial sums of the 32 voxels this wave finishes / of the partner's 32
    auto request_W16_p0 = [&](int tap, int sl) { WK[2 * sl][0].u = wload16(tap, 0); WK[2 * sl + 1][0].u = wload16(tap, 256); };       // piece 0 of both channel halves
    auto request_W16_p1 = [&](int tap, int sl) { WK[2 * sl][1].u = wload16(tap, 1024); WK[2 * sl + 1][1].u = wload16(tap, 1280); };
    auto request_W16_first = [&]() {                                         // the member's first WD - 1 taps
        const int T0 = tr ? KT0 : 0;
#pragma unroll
        for (int L = 0; L < WD - 1; ++L) { request_W16_p0(T0 + L, L); request_W16_p1(T0 + L, L); }
    };
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    f32x4a keep[4], xsend[4];                                                // K32: partial sums of the 32 voxels this wave finishes ([2 u' + v]) / of the partner's 32, from the end of the taps to the hand-over
#pragma unroll
    for (int k = 0; k < 4; ++k) { keep[k] = (f32x4a){0.f, 0.f, 0.f, 0.f}; xsend[k] = keep[k]; }
    bool xvalid = false;
    float* xbuf = reinterpret_cast<float*>(plds + NS * rowbytes);            // [4 waves][16 registers][64 lanes]
    // P25 computes its tiles transposed as well (a lane then holds channels 8 g + 4 half + i of ONE voxel: 16-byte skip loads and stores, no turn-around);
    // the 16 channels' exponents and biases do not fit its registers: two 32-entry tables behind xbuf
    int* tabE = reinterpret_cast<int*>(xbuf + 4096);
    float* tabB = reinterpret_cast<float*>(tabE + 32);
    if constexpr (P25) {
        if (tid < 32) {
            tabE[tid] = -(ea + h3_exp_w(am.w[tid < g.Cout ? tid : 0]));
            tabB[tid] = (bias && tid < g.Cout) ? bias[tid] : 0.f;
        }
    }
    // the filter fragments of a tile's first k-blocks are requested BEFORE the barrier that opens its segment (an L2 round trip per segment otherwise)
    if (grp == 0) {
        if constexpr (S16) request_W16_first();
        else if constexpr (P25) request_WP_first();
        else {
#pragma unroll
            for (int st = 0; st < PF; ++st) request_W(0, st, W[st]);
        }
    }
#pragma unroll 1
    for (int sg = 0; sg <= nseg; ++sg) {
        const int hi_next = sg + 1 < nseg ? need(sg + 1) : hiq;             // (uniform; both halves keep count)
        if (grp == (sg & 1)) {
            // ---- taps of one whole tile ----
            const int tile = 4 * sg + tsel;
            if constexpr (P25) {
              const int ptile = 2 * sg + pr;                                 // the pair's 64-voxel tile
              xvalid = sg < nseg && ptile * 64 < NV;
#pragma unroll
              for (int i = 0; i < 16; ++i) { keep16[i] = 0.f; send16[i] = 0.f; }
              if (xvalid) {
                const int dwb = a.Tp * REC;                                  // one column further
                const int Tu = a.Tp - 2;
                int rbp[2][3], xp0[2];                                       // byte address of the lane's record (voxel half uh: voxel 32 uh + col) in ring row hrel + dh, tap (0, 0); its key index (stage_store)
#pragma unroll
                for (int uh = 0; uh < 2; ++uh) {
                    int vi = ptile * 64 + 32 * uh + col;
                    vi = vi < NV ? vi : NV - 1;
                    const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
                    const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
                    const int e0 = (w * a.Tp + t) * REC;
                    const int s0 = hrel - fdiv(hrel, NS, a.mNslot) * NS;
                    const int s1 = s0 + 1 < NS ? s0 + 1 : s0 + 1 - NS, s2 = s1 + 1 < NS ? s1 + 1 : s1 + 1 - NS;
                    rbp[uh][0] = s0 * rowbytes + e0; rbp[uh][1] = s1 * rowbytes + e0; rbp[uh][2] = s2 * rowbytes + e0;
                    xp0[uh] = w * Tu + t;
                }
                // k-block st of a group covers its chunks 2 st (lanes 0..31) and 2 st + 1 (lanes 32..63): chunk c < 9 -> depth dt = c / 3, chunk c % 3 of the
                // record; c = 9 -> the gathered chunk 3 of the dt = 0 record.  Record at byte address ra holds chunk c at ra + ((c ^ key) << 4) (stage_store).
                // per (dh, dw) group and voxel half: the group's base and, for dt = 0, 1, 2, key << 4; set when the requests reach a group's first k-block
                int gab[2], gak[2][3];
                auto set_group = [&](int gg) {
                    const int dh = gg / 3, dw = gg % 3;
#pragma unroll
                    for (int uh = 0; uh < 2; ++uh) {
                        gab[uh] = rbp[uh][dh] + dw * dwb;
                        const int xg = xp0[uh] + dw * Tu;
#pragma unroll
                        for (int dt = 0; dt < 3; ++dt) gak[uh][dt] = ((xg + dt) << 3) & 0x70;      // (((x') >> 1) & 7) << 4 of the record of depth step dt
                    }
                };
                auto request_AP = [&](int K, int L, int uh0 = 0, int uh1 = 2) {
                    const int gg = K / NST, st = K % NST, sl = L % RK;
                    if (st == 0 && uh0 == 0) set_group(gg);
                    const int c0 = 2 * st, c1 = 2 * st + 1;
                    const int dt0 = c0 / 3, cc0 = c0 % 3, dt1 = c1 < 9 ? c1 / 3 : 0, cc1 = c1 < 9 ? c1 % 3 : 3;
#pragma unroll
                    for (int uh = uh0; uh < uh1; ++uh) {
                        const int k4 = half ? gak[uh][dt1] : gak[uh][dt0];      // (the halves of the wave read different chunks)
                        const int cd = half ? (cc1 << 4) + dt1 * REC : (cc0 << 4) + dt0 * REC;
                        const int a0 = ((cd & 0x70) ^ k4) + gab[uh] + (cd & ~0x7f);
                        AP[2 * sl + uh][0].u = *reinterpret_cast<const uint4*>(plds + a0);
                        AP[2 * sl + uh][1].u = *reinterpret_cast<const uint4*>(plds + (a0 ^ 64));
                    }
                };
                // member 0 finishes voxel half 0 and hands over half 1; member 1 the other way round
#define PP_C25(uh) (((uh) == 1) == R1 ? keep16 : send16)
                auto run_kb = [&](auto k0_tag, auto k1_tag) __attribute__((always_inline)) {
                    constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
                    constexpr bool R1 = K0 != 0;
                    if (K0 % NST != 0) set_group(K0 / NST);
#pragma unroll
                    for (int L = 0; L < RK - 1; ++L) request_AP(K0 + L, L);
#ifndef PPX_NOTAPS
#pragma unroll
                    for (int K = K0; K < K1; ++K) {
                        const int L = K - K0, sl = L % RK, KN = K + RK - 1, LN = L + RK - 1;       // k-block requested now (its ring slot held k-block K - 1)
                        PP_C25(0) = MFMA16H(WP[sl][0], AP[2 * sl][1], PP_C25(0));
                        __builtin_amdgcn_sched_barrier(0);
                        if (KN < K1) request_WP(KN, LN % RK);
                        __builtin_amdgcn_sched_barrier(0);
                        PP_C25(1) = MFMA16H(WP[sl][0], AP[2 * sl + 1][1], PP_C25(1));
                        __builtin_amdgcn_sched_barrier(0);
                        if (KN < K1) request_AP(KN, LN, 0, 1);
                        __builtin_amdgcn_sched_barrier(0);
                        PP_C25(0) = MFMA16H(WP[sl][1], AP[2 * sl][0], PP_C25(0));
                        __builtin_amdgcn_sched_barrier(0);
                        if (KN < K1) request_AP(KN, LN, 1, 2);
                        __builtin_amdgcn_sched_barrier(0);
                        PP_C25(1) = MFMA16H(WP[sl][1], AP[2 * sl + 1][0], PP_C25(1));
                        __builtin_amdgcn_sched_barrier(0);
                        PP_C25(0) = MFMA16H(WP[sl][0], AP[2 * sl][0], PP_C25(0));
                        __builtin_amdgcn_sched_barrier(0);
                        PP_C25(1) = MFMA16H(WP[sl][0], AP[2 * sl + 1][0], PP_C25(1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
#endif
                };
#ifndef PPX_NOPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                if (tr == 0) run_kb(std::integral_constant<int, 0>(), std::integral_constant<int, KB0>());
                else run_kb(std::integral_constant<int, KB0>(), std::integral_constant<int, 9 * NST>());
#ifndef PPX_NOPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
#undef PP_C25
#if !defined(PPX_IDLE) && !defined(PPX_NOEPI)
                if (tile < NTL) load_skip(tile);
#endif
              }
            } else
            if constexpr (K32) {
              const int ptile = 2 * sg + pr;                                 // the pair's 64-voxel tile
              xvalid = sg < nseg && ptile * 64 < NV;
#pragma unroll
              for (int k = 0; k < 4; ++k) { keep[k] = (f32x4a){0.f, 0.f, 0.f, 0.f}; xsend[k] = keep[k]; }      // (unconditionally: what they held is dead here)
              if (xvalid) {
                const int dwb = a.Tp * 16;                                   // one column further: Tp entries
                int rbu[4][3];                                               // byte address of the lane's entry (voxels 16 u .., its chunk's plane, piece 0) in ring row hrel + dh, tap (dw, dt) = (0, 0)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int vi = ptile * 64 + 16 * u + pm16;
                    vi = vi < NV ? vi : NV - 1;
                    const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
                    const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
                    const int e0 = (w * a.Tp + t) * 16 + kq * PS;
                    const int s0 = hrel - fdiv(hrel, NS, a.mNslot) * NS;
                    const int s1 = s0 + 1 < NS ? s0 + 1 : s0 + 1 - NS, s2 = s1 + 1 < NS ? s1 + 1 : s1 + 1 - NS;
                    rbu[u][0] = s0 * rowbytes + e0; rbu[u][1] = s1 * rowbytes + e0; rbu[u][2] = s2 * rowbytes + e0;
                }
                // tap T = 9 dh + 3 dw + dt, L = T - the member's first tap; rings AK[4 (L % AD) + u], WK[2 (L % WD) + v]
                auto request_A16 = [&](int T, int L, int u, int piece) {
                    const int dh = T / 9, dw = (T / 3) % 3, dt = T % 3, sl = L % AD;
                    const int a0 = rbu[u][dh] + dw * dwb;                     // (the same for the three dt of a (dh, dw) group: the depth step is the read's immediate offset)
                    if (piece) AK[4 * sl + u][1].u = *reinterpret_cast<const uint4*>(plds + a0 + 4 * PS + dt * 16);
                    else AK[4 * sl + u][0].u = *reinterpret_cast<const uint4*>(plds + a0 + dt * 16);
                };
                // accumulator of voxels 16 u .. and channel half v (transposed: a lane holds channels 16 v + 4 kq + i of voxel 16 u + pm16): member 0 finishes
                // u = 0, 1 and hands over u = 2, 3; member 1 the other way round
#define PP_CU(u, v) ((((u) >= 2) == R1) ? keep[2 * ((u) & 1) + (v)] : xsend[2 * ((u) & 1) + (v)])
#define PP_MM(u, v, pa, pb) PP_CU(u, v) = __builtin_amdgcn_mfma_f32_16x16x32_f16(WK[2 * (L % WD) + v][pb].h, AK[4 * (L % AD) + u][pa].h, PP_CU(u, v), 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
                // One tap = one k-block = 24 MFMAs (three piece pairs x eight accumulators); the requests go out in the gaps behind the first ones,
                // filters first, then the records' second pieces (the pair (1, 0) is the first the next tap multiplies), then their first pieces
                auto run_taps = [&](auto t0_tag, auto t1_tag) __attribute__((always_inline)) {
                    constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
                    constexpr bool R1 = T0 != 0;
#pragma unroll
                    for (int L = 0; L < AD - 1; ++L)
#pragma unroll
                        for (int u = 0; u < 4; ++u) { request_A16(T0 + L, L, u, 1); request_A16(T0 + L, L, u, 0); }
#ifndef PPX_NOTAPS
#pragma unroll
                    for (int T = T0; T < T1; ++T) {
                        const int L = T - T0;
                        const int TW = T + WD - 1, LW = L + WD - 1, TA = T + AD - 1, LA = L + AD - 1;     // the taps whose operands are requested now
                        PP_MM(0, 0, 1, 0);
                        if (TW < T1) request_W16_p0(TW, LW % WD);
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(0, 1, 1, 0);
                        if (TW < T1) request_W16_p1(TW, LW % WD);
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(1, 0, 1, 0);
                        if (TA < T1) { request_A16(TA, LA, 0, 1); request_A16(TA, LA, 1, 1); }
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(1, 1, 1, 0);
                        if (TA < T1) { request_A16(TA, LA, 2, 1); request_A16(TA, LA, 3, 1); }
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(2, 0, 1, 0);
                        if (TA < T1) { request_A16(TA, LA, 0, 0); request_A16(TA, LA, 1, 0); }
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(2, 1, 1, 0);
                        if (TA < T1) { request_A16(TA, LA, 2, 0); request_A16(TA, LA, 3, 0); }
                        __builtin_amdgcn_sched_barrier(0);
                        PP_MM(3, 0, 1, 0); PP_MM(3, 1, 1, 0);
                        PP_MM(0, 0, 0, 1); PP_MM(0, 1, 0, 1); PP_MM(1, 0, 0, 1); PP_MM(1, 1, 0, 1); PP_MM(2, 0, 0, 1); PP_MM(2, 1, 0, 1); PP_MM(3, 0, 0, 1); PP_MM(3, 1, 0, 1);
                        PP_MM(0, 0, 0, 0); PP_MM(0, 1, 0, 0); PP_MM(1, 0, 0, 0); PP_MM(1, 1, 0, 0); PP_MM(2, 0, 0, 0); PP_MM(2, 1, 0, 0); PP_MM(3, 0, 0, 0); PP_MM(3, 1, 0, 0);
                    }
#endif
                };
#ifndef PPX_NOPRIO
                __builtin_amdgcn_s_setprio(1);                               // the matrix phase before the other half's finishing work: an idle matrix pipe is the expensive kind of idle
#endif
                if (tr == 0) run_taps(std::integral_constant<int, 0>(), std::integral_constant<int, KT0>());
                else run_taps(std::integral_constant<int, KT0>(), std::integral_constant<int, 27>());
#ifndef PPX_NOPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
#undef PP_MM
#undef PP_CU
#if !defined(PPX_IDLE) && !defined(PPX_NOEPI)
                if (tile < NTL) load_skip(tile);
#endif
              }
            } else
            if (sg < nseg && tile < NTL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                int vi = tile * 32 + col;
                vi = vi < NV ? vi : NV - 1;
                const int hrel = fdiv(vi, nvr, a.mNvr), rem = vi - hrel * nvr;
                const int w = fdiv(rem, g.To, a.mTo), t = rem - w * g.To;
                const int vox0 = w * a.Tp + t;
                const int s0 = hrel - fdiv(hrel, NS, a.mNslot) * NS;
                const int s1 = s0 + 1 < NS ? s0 + 1 : s0 + 1 - NS, s2 = s1 + 1 < NS ? s1 + 1 : s1 + 1 - NS;
                // byte address of the lane's record for tap (dh, dw = 0, dt = 0): ring row hrel + dh
                const int rb0 = s0 * rowbytes + vox0 * REC, rb1 = s1 * rowbytes + vox0 * REC, rb2 = s2 * rowbytes + vox0 * REC;
                const int dwb = a.Tp * REC;                                  // one column further
                // Operand addresses with as little vector arithmetic as the swizzle allows.  Record at byte address ra holds logical chunk c at
                // ra + ((c ^ key) << 4), key = (ra >> 8) & 7.  Per (dh, dw) group: its base gb and, for dt = 0, 1, 2, key << 4 (K4); per request one
                // v_xad_u32 ((chunk << 4) ^ K4[dt]) + gb, the dt * 128 in the read's offset field where it is the same for the whole wave, and one
                // xor for the second piece (chunk ^ 4 = the address ^ 64: records are 128-byte aligned).
                struct GroupAddr { int gb, k4[3]; };
                const int Tu = a.Tp - 2, x0 = w * Tu + t;                    // the key index of the lane's (dw, dt) = (0, 0) record (stage_store)
                auto group_addr = [&](int base, int xg) -> GroupAddr {
                    GroupAddr q;
                    q.gb = base;
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) q.k4[dt] = ((xg + dt) << 3) & 0x70;      // (((xg + dt) >> 1) & 7) << 4
                    return q;
                };
                auto request_A = [&](const GroupAddr& ga, int st, Frag (&af)[NP]) {
                    int a0;
                    if constexpr (CIN == 25) {
                        const int c0 = 2 * st, c1 = 2 * st + 1;                 // chunk c of the group's ten: c < 9 -> (dt, cc) = (c / 3, c % 3); c = 9 -> the gathered channel-24 chunk of dt = 0
                        const int dt0 = c0 / 3, cc0 = c0 % 3, dt1 = c1 < 9 ? c1 / 3 : 0, cc1 = c1 < 9 ? c1 % 3 : 3;
                        const int k4 = half ? ga.k4[dt1] : ga.k4[dt0];          // (the halves of the wave read different chunks)
                        const int cd = half ? (cc1 << 4) + dt1 * REC : (cc0 << 4) + dt0 * REC;
                        a0 = ((cd & 0x70) ^ k4) + ga.gb + (cd & ~0x7f);
                        af[0].u = *reinterpret_cast<const uint4*>(plds + a0);
                    } else {
                        const int dt = st >> 1, kb = st & 1;
                        a0 = (cc4[kb] ^ ga.k4[dt]) + ga.gb;
                        af[0].u = *reinterpret_cast<const uint4*>(plds + a0 + dt * REC);
                    }
                    if constexpr (CIN == 25) af[1].u = *reinterpret_cast<const uint4*>(plds + (a0 ^ 64));
                    else af[1].u = *reinterpret_cast<const uint4*>(plds + (a0 ^ 64) + (st >> 1) * REC);
                };
                // One k-block = three MFMAs.  An MFMA holds the SIMD's vector issue for 8 of its 32 cycles and whatever else a wave issues in
                // the gap is hidden only while it fits the other 24 (MI355X_MICROARCH.md, issue costs): the requests of a later k-block are
                // therefore dealt out over the three gaps instead of standing in front of the first MFMA.
                auto group = [&](int gg, const GroupAddr& ga, const GroupAddr& gan, auto last_tag) {
                    constexpr bool LAST = decltype(last_tag)::value;         // the tile's last group requests nothing beyond itself: no load is left in flight at the barrier
#pragma unroll
                    for (int st = 0; st < NST; ++st) {
                        const bool same = st + PF < NST;                      // (compile-time after unrolling)
                        const int gq = same ? gg : gg + 1, sq = same ? st + PF : st + PF - NST;
                        acc = MFMA16H(A[st][1], W[st][0], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (same || !LAST) request_A(same ? ga : gan, sq, A[sq]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = MFMA16H(A[st][0], W[st][1], acc);
                        __builtin_amdgcn_sched_barrier(0);
                        if (same || !LAST) request_W(gq, sq, W[sq]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = MFMA16H(A[st][0], W[st][0], acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                // the three groups of one ring row (dh); the row base of dh + 1 for the requests that run ahead into it
                auto row_groups = [&](int dh, int rbh, int rbn, auto last_tag) {
                    constexpr bool LASTROW = decltype(last_tag)::value;
                    const GroupAddr g0 = group_addr(rbh, x0), g1 = group_addr(rbh + dwb, x0 + Tu), g2 = group_addr(rbh + 2 * dwb, x0 + 2 * Tu), gn = group_addr(rbn, x0);
                    group(3 * dh, g0, g1, std::false_type());
                    group(3 * dh + 1, g1, g2, std::false_type());
                    group(3 * dh + 2, g2, gn, std::integral_constant<bool, LASTROW>());
                };
                {
                    const GroupAddr g0 = group_addr(rb0, x0);
#pragma unroll
                    for (int st = 0; st < PF; ++st) request_A(g0, st, A[st]);
                }
#ifndef PPX_NOTAPS
#pragma unroll 1
                for (int dh = 0; dh < 2; ++dh) row_groups(dh, dh == 0 ? rb0 : rb1, dh == 0 ? rb1 : rb2, std::false_type());
                row_groups(2, rb2, rb2, std::true_type());
#endif
#if !defined(PPX_IDLE) && !defined(PPX_NOEPI)
                load_skip(tile);
#endif
            }
#if !defined(PPX_IDLE) && !defined(PPX_NOSTAGE)
            {   // the row this half stages in its finishing segment (the first one beyond what segment sg + 1 reads)
                const int hi_n2 = sg + 2 < nseg ? need(sg + 2) : hi_next;
                have_nv = hi_n2 > hi_next;
                if (have_nv) stage_load(hi_next + 1, nv_);
            }
#endif
            XS_ACC(2);
        } else {
            // ---- finish the tile of the previous segment; stage the rows of the next one ----
            const int tile = 4 * (sg - 1) + tsel;
#ifdef PPX_IDLE
            const bool fin = false, do_load = false;
            { float t_ = 0.f;
#pragma unroll
              for (int i = 0; i < 16; ++i) t_ += acc[i];
              if constexpr (S16) {                              // (the new forms keep their sums elsewhere: without this the ablation would drop half of the MFMAs as dead code)
#pragma unroll
                  for (int k = 0; k < 4; ++k) t_ += keep[k][0] + keep[k][1] + keep[k][2] + keep[k][3];
              }
              if constexpr (P25) {
#pragma unroll
                  for (int i = 0; i < 16; ++i) t_ += keep16[i];
              }
              if (t_ == 1234.5f) ybase[lane] = t_; }
#else
#ifdef PPX_NOEPI
            const bool fin = false;
            { float t_ = 0.f;
#pragma unroll
              for (int i = 0; i < 16; ++i) t_ += acc[i];
              if constexpr (S16) {                              // (the new forms keep their sums elsewhere: without this the ablation would drop half of the MFMAs as dead code)
#pragma unroll
                  for (int k = 0; k < 4; ++k) t_ += keep[k][0] + keep[k][1] + keep[k][2] + keep[k][3];
              }
              if constexpr (P25) {
#pragma unroll
                  for (int i = 0; i < 16; ++i) t_ += keep16[i];
              }
              if (t_ == 1234.5f) ybase[lane] = t_; }
#else
            const bool fin = sg >= 1 && tile < NTL;                          // wave-uniform
#endif
#ifdef PPX_NOSTAGE
            const bool do_load = false;
#else
            const bool do_load = hi_next > hiq;
#endif
#endif
            if (do_load && !have_nv) stage_load(hiq + 1, nv_);               // (only the very first finishing segment: no taps came before it)
            have_nv = false;
            if (fin) {
              if constexpr (P25) {
                // keep16[4 g + i] (+ the partner's share, accumulator layout, in this wave's slot of xbuf): channel 8 g + 4 half + i of voxel `col` -- the
                // layout of the 16-byte skip loads and output stores themselves
                const bool full = (tile * 32 + 32 <= NV) && g.Cout == 32;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int c0 = 8 * jj + 4 * half;
                    const float4 xp = *reinterpret_cast<const float4*>(xbuf + tsel * 1024 + (jj * 64 + lane) * 4);
                    const int4 e4 = *reinterpret_cast<const int4*>(tabE + c0);
                    const float4 b4 = *reinterpret_cast<const float4*>(tabB + c0);
                    const float xq[4] = {xp.x, xp.y, xp.z, xp.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
                    const int eq4[4] = {e4.x, e4.y, e4.z, e4.w};
                    f32x4u o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = ldexpf(keep16[4 * jj + i] + xq[i], eq4[i]) + bq[i];
                        if (g.relu) v = fmaxf(v, 0.f);
                        o[i] = v + skq[jj][i];
                    }
                    if (full) {
                        *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    } else if (eoff[jj] >= 0) {
                        if (c0 + 4 <= g.Cout) {
                            *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (c0 + c < g.Cout) { ybase[eoff[jj] + c] = o[c]; omax = fmaxf(omax, fabsf(o[c])); }
                        }
                    }
                }
              } else
              if constexpr (S16) {
                // keep[2 u + v][i] (+ the partner's share): channel 16 v + 4 kq + i of voxel 16 u + pm16 -- the layout of the 16-byte skip loads and output stores themselves
                const bool full = (tile * 32 + 32 <= NV) && g.Cout == 32;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int v_ = jj & 1, c0 = 16 * v_ + 4 * kq;
                    f32x4u o;
                    const float4 xp = *reinterpret_cast<const float4*>(xbuf + tsel * 1024 + (jj * 64 + lane) * 4);      // the partner's partial sums of these four values (its taps)
                    const float xq[4] = {xp.x, xp.y, xp.z, xp.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = ldexpf(keep[jj][i] + xq[i], eun8[v_][i]) + bv8[v_][i];
                        if (g.relu) v = fmaxf(v, 0.f);
                        o[i] = v + skq[jj][i];
                    }
                    if (full) {
                        *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    } else if (eoff[jj] >= 0) {
                        if (c0 + 4 <= g.Cout) {
                            *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (c0 + c < g.Cout) { ybase[eoff[jj] + c] = o[c]; omax = fmaxf(omax, fabsf(o[c])); }
                        }
                    }
                }
              } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = ldexpf(acc[i], eun) + bv;                       // the filter column's exponent and the bias are per lane in the accumulator layout
                    if (g.relu) v = fmaxf(v, 0.f);
                    turn[rowmap(i, half) * 32 + col] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (a wave's LDS operations execute in order: the wait orders the compiler)
                float4 tq[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) tq[jj] = *reinterpret_cast<const float4*>(turn + (er + 8 * jj) * 32 + 4 * eq);
                const bool full = (tile * 32 + 32 <= NV) && g.Cout == 32;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 t = tq[jj];
                    f32x4u o = {t.x + skq[jj][0], t.y + skq[jj][1], t.z + skq[jj][2], t.w + skq[jj][3]};
                    if (full) {
                        *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    } else if (eoff[jj] >= 0) {
                        if (4 * eq + 4 <= g.Cout) {
                            *reinterpret_cast<f32x4u*>(ybase + eoff[jj]) = o;
                            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) if (4 * eq + c < g.Cout) { ybase[eoff[jj] + c] = o[c]; omax = fmaxf(omax, fabsf(o[c])); }
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the turn buffer is free again before the partner wave can reach its own epilogue (next barrier)
              }
            }
            if (do_load) {
                stage_store(hiq + 1, nv_);
#pragma unroll 1
                for (int q = hiq + 2; q <= hi_next; ++q) {                   // (a short row range can need two new rows for one segment)
                    stage_load(q, nv_);
                    stage_store(q, nv_);
                }
            }
            if (sg + 1 < nseg) {
                if constexpr (S16) request_W16_first();
                else if constexpr (P25) request_WP_first();
                else {
#pragma unroll
                    for (int st = 0; st < PF; ++st) request_W(0, st, W[st]);
                }
            }
            XS_ACC(3);
        }
        hiq = hi_next;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: global loads requested for the next segment and the output stores stay in flight
        if constexpr (K32) {
            // hand-over of the partial sums: the half that finished tiles has read its share of the previous hand-over (barrier above); the half that ran taps
            // leaves, in its PARTNER's slot, what it summed for the partner's 32 voxels; second barrier; the partner adds it in its finishing segment
            if (grp == (sg & 1) && xvalid) {
                float* xs = xbuf + (tsel ^ 1) * 1024;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (P25) *reinterpret_cast<float4*>(xs + (k * 64 + lane) * 4) = make_float4(send16[4 * k], send16[4 * k + 1], send16[4 * k + 2], send16[4 * k + 3]);
                    else *reinterpret_cast<float4*>(xs + (k * 64 + lane) * 4) = make_float4(xsend[k][0], xsend[k][1], xsend[k][2], xsend[k][3]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        XS_ACC(4);
    }
    if (am.y) amax_commit(omax, am.y + n);
    XS_OUT;
}

// plan of the alternating-halves form: geometry of pstrip_plan(); a half (256 threads) stages a row; ring depth from the rows two consecutive segments touch
static bool pp_plan(const ConvGeom& g, StripPlan& p, int& rvp)
{
    p.ok = false;
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.Cout > 32 || (g.Cin != 25 && g.Cin != 32)) return false;
    if (g.reflect_hw && (g.ph != 1 || g.pw != 1 || g.Hi < 2 || g.Wi < 2)) return false;        // mirrored pads of 1: the staging mirrors its source row / column (convReducer_1)
    if (g.Ho != g.Hi + 2 * g.ph - 2 || g.Wo != g.Wi + 2 * g.pw - 2 || g.To != g.Ti + 2 * g.pt - 2) return false;
    if (g.Cin == 25 && (g.pt != 1 || g.To != g.Ti)) return false;
    if (g.Ho < 3) return false;
    const int Tp = g.To + 2;
    for (int ns = 1; ns <= 4; ++ns) {
        if (g.Wo % ns) continue;
        const int wt = g.Wo / ns, nvr = wt * g.To;
        const int nvs = ((ns == 1 && !g.reflect_hw) ? g.Wi : wt + 2) * g.Ti, items = nvs * (g.Cin == 25 ? 3 : 4);     // staged voxels / items per row (a half = 256 threads stages it; mirrored pad columns are staged)
        if (items > 256 * 3 || nvs > 256 || nvr < 32) continue;
        int nstrips = (256 + g.N * ns - 1) / (g.N * ns);
        if (nstrips < 1) nstrips = 1;
        if (nstrips > g.Ho / 4) nstrips = g.Ho / 4 > 0 ? g.Ho / 4 : 1;
        const int SR = (g.Ho + nstrips - 1) / nstrips;
        nstrips = (g.Ho + SR - 1) / SR;
        // ring depth: while a segment's taps read rows hlo(s) .. need(s), the other half writes the rows up to need(s + 1)
        const int NV = SR * nvr, nseg = ((NV + 31) / 32 + 3) / 4;
        auto needf = [&](int sg) { const int vl = std::min(NV - 1, (sg + 1) * 128 - 1); return vl / nvr + 2; };
        int nslot = needf(0) + 1;
        for (int sg = 0; sg + 1 < nseg; ++sg) nslot = std::max(nslot, needf(sg + 1) - (sg * 128) / nvr + 1);
        size_t rowb = (size_t)(wt + 2) * Tp * 128;
        if (g.Cin == 32) rowb = std::max(rowb, (size_t)8 * 16 * (((size_t)((wt + 2) * Tp + 13) & ~(size_t)15) + 2));     // the 32-channel new form's planar row (conv3_pp_kernel)
        const size_t need = (size_t)nslot * rowb + (size_t)4 * 1024 * sizeof(float) + 256;          // ring, hand-over / turn-around buffer, two 32-entry tables
        if (need > 163840) continue;
        p.ok = true; p.CC = g.Cin; p.KS = 16; p.lds_bytes = need; p.grid = g.N * nstrips * ns;
        p.a.g = g; p.a.Wp = wt + 2; p.a.Tp = Tp; p.a.SR = SR; p.a.nstrips = nstrips; p.a.nsplit = ns; p.a.Wt = wt;
        p.a.mTo = magic(g.To); p.a.mNvr = magic(nvr); p.a.mTi = magic(g.Ti); p.a.mSrcCol = 0;
        p.a.nslot = nslot; p.a.mNslot = magic(nslot);
        rvp = items <= 512 ? 2 : 3;
        return true;
    }
    return false;
}

// plan of the piece-ring form: 3x3x3, zero pads, Cin 25 / 32, Cout <= 32, >= 128 voxels per output row
static bool pstrip_plan(const ConvGeom& g, StripPlan& p)
{
    p.ok = false;
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.reflect_hw || g.Cout > 32 || (g.Cin != 25 && g.Cin != 32)) return false;
    if (g.Ho != g.Hi + 2 * g.ph - 2 || g.Wo != g.Wi + 2 * g.pw - 2 || g.To != g.Ti + 2 * g.pt - 2) return false;
    if (g.Cin == 25 && (g.pt != 1 || g.To != g.Ti)) return false;            // (the gathered channel-24 chunk is written for depth pad 1)
    if (g.Ho < 3) return false;
    const int Tp = g.To + 2;
    // column ranges per row: as few as make four staged rows fit the LDS (T = 13: two), each row of a range still >= 128 voxels
    int nsplit = 0, Wt = 0;
    size_t lds = 0;
    for (int ns = 1; ns <= 4; ++ns) {
        if (g.Wo % ns) continue;
        const int wt = g.Wo / ns;
        const size_t need = (size_t)4 * (wt + 2) * Tp * 128 + (size_t)4 * 16 * 64 * sizeof(float);
        if (need <= 163840 && wt * g.To >= 128 && (wt + 2) * g.Ti * 4 <= 512 * 2) { nsplit = ns; Wt = wt; lds = need; break; }
    }
    if (!nsplit) return false;
    int nstrips = (256 + g.N * nsplit - 1) / (g.N * nsplit);
    if (nstrips < 1) nstrips = 1;
    if (nstrips > g.Ho / 4) nstrips = g.Ho / 4 > 0 ? g.Ho / 4 : 1;
    const int SR = (g.Ho + nstrips - 1) / nstrips;
    nstrips = (g.Ho + SR - 1) / SR;
    p.ok = true; p.CC = g.Cin; p.KS = 16; p.lds_bytes = lds; p.grid = g.N * nstrips * nsplit;
    p.a.g = g; p.a.Wp = Wt + 2; p.a.Tp = Tp; p.a.SR = SR; p.a.nstrips = nstrips; p.a.nsplit = nsplit; p.a.Wt = Wt;
    p.a.mTo = magic(g.To); p.a.mNvr = magic(Wt * g.To); p.a.mTi = magic(g.Ti); p.a.mSrcCol = 0;
    return true;
}

static StripPlan strip_plan(const ConvGeom& g)
{
    StripPlan p;
    p.ok = false;
    if (g.kh != 3 || g.kw != 3 || g.kt != 3 || g.reflect_hw || g.Cout > 32) return p;
    if (g.Ho != g.Hi + 2 * g.ph - 2 || g.Wo != g.Wi + 2 * g.pw - 2 || g.To != g.Ti + 2 * g.pt - 2) return p;
    int CC, slots;
    if (g.Cin == 25) { CC = 25; slots = 5; }            // 5-slot ring: the next row is staged asynchronously
    else if (g.Cin == 32) { CC = 32; slots = 4; }       // CP = 33: only four rows fit, staged between rounds
    else return p;
    const int CP = (CC & 1) ? CC : CC + 1;
    const int nvr = g.Wo * g.To;
    if (nvr < 128 || g.Ho < 3) return p;
    const int Wp = g.Wo + 2, Tp = g.To + 2;
    const int CG = (CC % 4 == 0) ? CC / 4 : CC;
    if ((g.Wi * g.Ti * CG + 511) / 512 > ((CC % 4 == 0) ? 4 : 10)) return p;  // staging registers per thread
    const size_t lds = ((size_t)slots * Wp * Tp * CP + 8 + 4 * 16 * 64) * sizeof(float);
    if (lds > 163840) return p;
    // strips per patch: fill the 256 CUs, but keep strips long enough to amortise the 4-row prologue
    int nstrips = (256 + g.N - 1) / g.N;
    if (nstrips < 1) nstrips = 1;
    if (nstrips > g.Ho / 4) nstrips = g.Ho / 4 > 0 ? g.Ho / 4 : 1;
    const int SR = (g.Ho + nstrips - 1) / nstrips;
    nstrips = (g.Ho + SR - 1) / SR;
    p.ok = true; p.CC = CC; p.KS = (CC + 1) / 2; p.lds_bytes = (lds + 15) & ~(size_t)15; p.grid = g.N * nstrips;
    p.a.g = g; p.a.Wp = Wp; p.a.Tp = Tp; p.a.SR = SR; p.a.nstrips = nstrips; p.a.nsplit = 1; p.a.Wt = g.Wo;
    p.a.mTo = magic(g.To); p.a.mNvr = magic(nvr); p.a.mTi = magic(g.Ti); p.a.mSrcCol = magic(g.Ti * CC);
    return p;
}

bool mfma_conv_strip_supported(const ConvGeom& g) { return strip_plan(g).ok; }
// does x6_conv_strip_forward(g, ..., arith) read per-tap fragments (PACK_*_CONV) even for 25 input channels?  (the H3 piece-ring kernel does;
// the other 25-channel split kernels read the K-concatenated PACK_*_CONVK form)
bool x6_strip_wants_tap_fragments(const ConvGeom& g, int arith)
{
    StripPlan pp;
    int rvp;
    return arith == 2 && (pstrip_plan(g, pp) || pp_plan(g, pp, rvp));   // (the alternating-halves form also takes rows shorter than 128 voxels: the later reducers)
}


static int strip_launch(const ConvGeom& g, const float* x, const float* gate, const float* wfrag, const float* bias,
                        const float* skip, float* y, int arith, const Amax& am, hipStream_t s)
{
    if (arith == 2 && (!am.x || !am.w)) { set_error("x6_conv_strip_forward: H3 arithmetic needs the operands' amax slots", hipSuccess); return PROBAV_EINVAL; }
    if (arith == 2 && x6_strip_wants_tap_fragments(g, arith)) {               // H3: the piece-ring kernel (filters: PACK_H3_CONV)
        if (cw4_enabled() && cw4_conv_supported(g, gate)) return cw4_conv_forward(g, x, wfrag, bias, skip, y, am, s);      // one wave per SIMD, the filter in registers (kernels_cw4.hip)
        StripPlan pp;
        int rvp = 2;
        if (pp_plan(g, pp, rvp)) {
            static std::once_flag onceq;
            std::call_once(onceq, [] {
                allow_big_lds(conv3_pp_kernel<32, true, 3>);
                allow_big_lds(conv3_pp_kernel<32, false, 2, true>); allow_big_lds(conv3_pp_kernel<32, true, 2, true>);
                allow_big_lds(conv3_pp_kernel<32, false, 3, true>);
                allow_big_lds(conv3_pp_kernel<25, false, 2, true>); allow_big_lds(conv3_pp_kernel<25, false, 3, true>); });
            // the instances that ship: 25 input channels (the forward pass of the residual blocks) in the pair-split form; 32 input channels (backward-data, reducers) in
            // the 16x16x32 MFMA form -- except the gated layer with three staging items per thread (reducers at T = 13), whose new-form instance would spill 42
            // registers: it keeps the one-wave-per-tile 32x32x16 form.  (Rounds 3 / 4 carried the superseded forms behind PROBAV_PP_K16 / PROBAV_PP_OLD25 for A/B runs.)
            if (g.Cin == 25 && gate) { set_error("x6_conv_strip_forward: a gated 25-channel layer has no piece-ring instance (none occurs in the reference's networks)", hipSuccess); return PROBAV_EINVAL; }
#define PROBAV_PPK(C, G, R) hipLaunchKernelGGL((conv3_pp_kernel<C, G, R, true>), dim3(pp.grid), dim3(512), pp.lds_bytes, s, pp.a, x, gate, (const uint4*)wfrag, bias, skip, y, am)
            if (g.Cin == 25) { if (rvp == 2) PROBAV_PPK(25, false, 2); else PROBAV_PPK(25, false, 3); }
            else if (gate) {
                if (rvp == 2) PROBAV_PPK(32, true, 2);
                else hipLaunchKernelGGL((conv3_pp_kernel<32, true, 3>), dim3(pp.grid), dim3(512), pp.lds_bytes, s, pp.a, x, gate, (const uint4*)wfrag, bias, skip, y, am);
            } else { if (rvp == 2) PROBAV_PPK(32, false, 2); else PROBAV_PPK(32, false, 3); }
#undef PROBAV_PPK
            return check_launch("conv3_pp");
        }
        if (!pstrip_plan(g, pp)) { set_error("x6_conv_strip_forward: no piece-ring plan for this geometry", hipSuccess); return PROBAV_EINVAL; }
        static std::once_flag oncep;
        std::call_once(oncep, [] {
            allow_big_lds(conv3_pstrip_kernel<25, false>); allow_big_lds(conv3_pstrip_kernel<25, true>);
            allow_big_lds(conv3_pstrip_kernel<32, false>); allow_big_lds(conv3_pstrip_kernel<32, true>); });
#define PROBAV_PSTRIP(C, G) hipLaunchKernelGGL((conv3_pstrip_kernel<C, G>), dim3(pp.grid), dim3(512), pp.lds_bytes, s, pp.a, x, gate, (const uint4*)wfrag, bias, skip, y, am)
        if (g.Cin == 25) { if (gate) PROBAV_PSTRIP(25, true); else PROBAV_PSTRIP(25, false); }
        else             { if (gate) PROBAV_PSTRIP(32, true); else PROBAV_PSTRIP(32, false); }
#undef PROBAV_PSTRIP
        return check_launch("conv3_pstrip");
    }
    const StripPlan p = strip_plan(g);
    if (!p.ok) { set_error("mfma_conv_strip_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] {
        allow_big_lds(conv3_strip_kernel<25, 13, false, 5, 0>); allow_big_lds(conv3_strip_kernel<25, 13, true, 5, 0>);
        allow_big_lds(conv3_strip_kernel<32, 16, false, 4, 0>); allow_big_lds(conv3_strip_kernel<32, 16, true, 4, 0>);
        allow_big_lds(conv3_strip_kernel<25, 13, false, 5, 1>); allow_big_lds(conv3_strip_kernel<25, 13, true, 5, 1>);
        allow_big_lds(conv3_strip_kernel<32, 16, false, 4, 1>); allow_big_lds(conv3_strip_kernel<32, 16, true, 4, 1>);
        allow_big_lds(conv3_strip_kernel<25, 13, false, 5, 2>); allow_big_lds(conv3_strip_kernel<25, 13, true, 5, 2>);
        allow_big_lds(conv3_strip_kernel<32, 16, false, 4, 2>); allow_big_lds(conv3_strip_kernel<32, 16, true, 4, 2>); });
#define PROBAV_STRIP(C, K, G, S, X) hipLaunchKernelGGL((conv3_strip_kernel<C, K, G, S, X>), dim3(p.grid), dim3(512), p.lds_bytes, s, p.a, x, gate, (const float4*)wfrag, bias, skip, y, am)
#define PROBAV_STRIP_A(X) do { \
        if (p.CC == 25) { if (gate) PROBAV_STRIP(25, 13, true, 5, X); else PROBAV_STRIP(25, 13, false, 5, X); } \
        else            { if (gate) PROBAV_STRIP(32, 16, true, 4, X); else PROBAV_STRIP(32, 16, false, 4, X); } } while (0)
    if (arith == 2) PROBAV_STRIP_A(2);
    else if (arith == 1) PROBAV_STRIP_A(1);
    else PROBAV_STRIP_A(0);
#undef PROBAV_STRIP_A
#undef PROBAV_STRIP
    return check_launch("conv3_strip");
}

int mfma_conv_strip_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag, const float* bias,
                            const float* skip, float* y, const Amax& am, hipStream_t s)
{
    return strip_launch(g, x, gate, wfrag, bias, skip, y, 0, am, s);
}
int x6_conv_strip_forward(const ConvGeom& g, const float* x, const float* gate, const float* wfrag6, const float* bias,
                          const float* skip, float* y, int arith, const Amax& am, hipStream_t s)
{
    return strip_launch(g, x, gate, wfrag6, bias, skip, y, arith, am, s);
}

// ---------------------------------------------------------------------------------------------------
// conv3 backward-filter
// ---------------------------------------------------------------------------------------------------
template <int CIN, bool GATE>
__global__ __launch_bounds__(256, 1) void conv3_wgrad_mfma_kernel(TileArgs a, int total_tiles, const float* __restrict__ x,
                                                                 const float* __restrict__ dy, const float* __restrict__ gate,
                                                                 float* __restrict__ partial, float* __restrict__ partial_b)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CP = (CIN & 1) ? CIN : CIN + 1;
    constexpr int KR = 27 * CIN;                    // rows of the flattened filter matrix
    constexpr int NMT = (KR + 31) / 32;             // M tiles (22 for Cin 25, 27 for Cin 32)
    constexpr int MTW = (NMT + 3) / 4;              // M tiles per wave
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;

    // per-lane constants of filter-matrix row (tap, ci) for each M tile this wave owns: its dh and its offset inside a row slot
    int offB[MTW], dhA[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int row = (wave + 4 * j) * 32 + col;
        const int tap = row / CIN, ci = row - tap * CIN;
        const int dh = tap / 9, dw = (tap / 3) % 3, dt = tap % 3;
        dhA[j] = (row < KR) ? dh : 0;
        offB[j] = (row < KR) ? (dw * a.Tp + dt) * CP + ci : 0;
    }
    f32x16 acc[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    // Each workgroup walks a CONTIGUOUS run of (patch, row) tiles.  With one output row per tile the three staged input rows
    // form a ring in LDS: moving to the next row of the same patch re-stages ONE row instead of three.
    const int rowfloats = a.Wp * a.Tp * CP;
    const bool ring = a.R == 1 && a.rows == 3 && !g.reflect_hw && (rowfloats & 3) == 0;
    const int tbeg = (int)((long)blockIdx.x * total_tiles / gridDim.x), tend = (int)((long)(blockIdx.x + 1) * total_tiles / gridDim.x);
    int prev_n = -1, prev_h = -1000;
    STAMP_DECL;
    STAMP(0);
    for (int tile = tbeg; tile < tend; ++tile) {
        STAMP_T0;
        const int n = tile / a.ntile_rows, h0 = (tile - n * a.ntile_rows) * a.R;
        const int Rr = g.Ho - h0 < a.R ? g.Ho - h0 : a.R;
        const int nv = Rr * g.Wo * g.To, nsteps = (nv + 1) >> 1;
        const long out_base = ((long)n * g.Ho + h0) * g.Wo * g.To;
        const int rot = ring ? ((h0 - g.ph) % 3 + 3) % 3 : 0;
        __syncthreads();
        if (ring && n == prev_n && h0 == prev_h + 1) fill_tile<CIN, CP>(a, lds, x, nullptr, n, h0, 0, tid, 2, rot);
        else fill_tile<CIN, CP>(a, lds, x, nullptr, n, h0, 0, tid, -1, rot);
        __syncthreads();
        STAMP_ACC(st_fill);
        prev_n = n; prev_h = h0;
        int offA[MTW];
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            int slot = dhA[j] + rot;
            slot = slot >= 3 ? slot - 3 : slot;
            offA[j] = (ring ? slot : dhA[j]) * rowfloats + offB[j];
        }
        // B operand dy[voxel 2s+half][col] is requested a whole group of PF steps ahead (HBM/L2 latency), the A operands
        // (LDS) one step ahead; the MFMAs of a step then run register-only.  Scheduling barriers pin this order.
        constexpr int PF = 8;
        float bq[PF], bn[PF];
        auto load_dy = [&](int step) -> float {
            const int vi = 2 * step + half;
            const bool live = vi < nv && col < g.Cout;
            const long o = live ? (out_base + vi) * g.Cout + col : 0;      // unconditional load, clamped address
            float d = dy[o];
            if constexpr (GATE) d = gate[o] > 0.f ? d : 0.f;
            return live ? d : 0.f;
        };
        float acur[MTW], anxt[MTW];
        auto load_a = [&](int step, float* dst) {
            int vi = 2 * step + half;
            vi = vi < nv ? vi : nv - 1;
            const float* pv = lds + tile_voxel_off(a, vi, CP);
#pragma unroll
            for (int j = 0; j < MTW; ++j) dst[j] = pv[offA[j]];
        };
#pragma unroll
        for (int u = 0; u < PF; ++u) bq[u] = load_dy(u);
        load_a(0, acur);
        for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) bn[u] = load_dy(s0 + PF + u);
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                load_a(s0 + u + 1, anxt);
                __builtin_amdgcn_sched_barrier(0);
                if (s0 + u < nsteps) {
                    const float b = bq[u];
                    bsum += b;
#pragma unroll
                    for (int j = 0; j < MTW; ++j)
                        if (wave + 4 * j < NMT) acc[j] = MFMA32(acur[j], b, acc[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < MTW; ++j) acur[j] = anxt[j];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) bq[u] = bn[u];
        }
        STAMP_ACC(st_steps);
    }
    STAMP_OUT;
    STAMP(7);
    // slab of this workgroup: [KR][Cout]
    float* pp = partial + (long)blockIdx.x * KR * g.Cout;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        if (wave + 4 * j >= NMT) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wave + 4 * j) * 32 + rowmap(r, half);
            if (row < KR && col < g.Cout) pp[(long)row * g.Cout + col] = acc[j][r];
        }
    }
    if (wave == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (half == 0 && col < g.Cout) partial_b[(long)blockIdx.x * g.Cout + col] = bsum;
    }
}

static int wgrad_grid(const ConvPlan& p, const ConvGeom& g)
{
    const int total = g.N * p.a.ntile_rows;
    const int per_cu = p.lds_bytes <= 81920 ? 2 : 1;
    int grid = 256 * per_cu;
    return grid < total ? grid : total;
}

bool mfma_wgrad_supported(const ConvGeom& g)
{
    if (g.Cout > 32) return false;
    return conv_plan(g, true).ok;
}

size_t mfma_wgrad_partial_floats(const ConvGeom& g)
{
    const ConvPlan p = conv_plan(g, true);
    if (!p.ok) return 0;
    return (size_t)wgrad_grid(p, g) * ((size_t)27 * g.Cin * g.Cout + g.Cout);
}

int mfma_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate, float* dw, float* db,
                    float* partial, hipStream_t s)
{
    const ConvPlan p = conv_plan(g, true);
    if (!p.ok || g.Cout > 32) { set_error("mfma_conv_wgrad: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    const int grid = wgrad_grid(p, g), total = g.N * p.a.ntile_rows;
    const long nw = (long)27 * g.Cin * g.Cout;
    float* partial_b = partial + (size_t)grid * nw;
    static std::once_flag once;
#define PROBAV_WGRAD(CI, GT) hipLaunchKernelGGL((conv3_wgrad_mfma_kernel<CI, GT>), dim3(grid), dim3(256), p.lds_bytes, s, p.a, total, x, dy, gate, partial, partial_b)
    std::call_once(once, [] {
        allow_big_lds(conv3_wgrad_mfma_kernel<25, false>); allow_big_lds(conv3_wgrad_mfma_kernel<25, true>);
        allow_big_lds(conv3_wgrad_mfma_kernel<32, false>); allow_big_lds(conv3_wgrad_mfma_kernel<32, true>);
        allow_big_lds(conv3_wgrad_mfma_kernel<1, false>); allow_big_lds(conv3_wgrad_mfma_kernel<1, true>); });
    if (g.Cin == 1) { if (gate) PROBAV_WGRAD(1, true); else PROBAV_WGRAD(1, false); }
    else if (g.Cin == 25) { if (gate) PROBAV_WGRAD(25, true); else PROBAV_WGRAD(25, false); }
    else { if (gate) PROBAV_WGRAD(32, true); else PROBAV_WGRAD(32, false); }
#undef PROBAV_WGRAD
    int rc = check_launch("conv3_wgrad_mfma");
    if (rc) return rc;
    return mfma_wgrad_reduce(partial, partial_b, dw, db, nw, g.Cout, grid, s);
}

int mfma_wgrad_reduce(const float* partial, const float* partial_b, float* dw, float* db, long nw, int Cout, int slabs, hipStream_t s0)
{
    // (the engine's side stream during a backward pass, at its next flush -- batched with the other slab sums of that flush: probav_common.h, slab_sum_later)
    SlabSumJob jobs[2] = {{partial, dw, nw, (int)nw, slabs}, {partial_b, db, (long)Cout, Cout, slabs}};
    return slab_sum_later(s0, jobs, db ? 2 : 1);
}

// ---------------------------------------------------------------------------------------------------
// fused expConv + ReLU + decConv forward (1x1x1, F=32 -> E=256 -> D<=32)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void pw_fwd_mfma_kernel(const float* __restrict__ x, const float4* __restrict__ w1frag,
                                                            const float4* __restrict__ w2frag, const float* __restrict__ b1,
                                                            const float* __restrict__ b2, float* __restrict__ dec,
                                                            long nvox, int D)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* sW1 = reinterpret_cast<float4*>(lds);              // [8 chunks][4][64] float4  (32 KB)
    float4* sW2 = sW1 + 8 * 4 * 64;                              // same (32 KB)
    float* sB1 = reinterpret_cast<float*>(sW2 + 8 * 4 * 64);    // 256
    float* sB2 = sB1 + 256;                                      // 32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    for (int i = tid; i < 8 * 4 * 64; i += 256) { sW1[i] = w1frag[i]; sW2[i] = w2frag[i]; }
    sB1[tid] = b1[tid];
    if (tid < 32) sB2[tid] = tid < D ? b2[tid] : 0.f;
    __syncthreads();

    const long ntiles = (nvox + 31) >> 5;
    const long wstride = (long)gridDim.x * 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += wstride) {
        long v = tile * 32 + col;
        const bool vok = v < nvox;
        if (!vok) v = nvox - 1;
        // B operand of the first product: X^T, k = (s, half) <-> cin = 16*half + s
        float xs[16];
        const float4* xp = reinterpret_cast<const float4*>(x + v * 32 + 16 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t = xp[q]; xs[4 * q] = t.x; xs[4 * q + 1] = t.y; xs[4 * q + 2] = t.z; xs[4 * q + 3] = t.w; }
        f32x16 T;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[r] = sB2[rowmap(r, half)];
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            f32x16 H;
#pragma unroll
            for (int r = 0; r < 16; ++r) H[r] = sB1[32 * c + rowmap(r, half)];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 aw = sW1[(c * 4 + q) * 64 + lane];
                H = MFMA32(aw.x, xs[4 * q], H); H = MFMA32(aw.y, xs[4 * q + 1], H);
                H = MFMA32(aw.z, xs[4 * q + 2], H); H = MFMA32(aw.w, xs[4 * q + 3], H);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) H[r] = fmaxf(H[r], 0.f);
            // second product contracts over the hidden channels = the ROW index of H: accumulator registers
            // feed the B operand directly (k-step s <-> register s, k = lane half)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 aw = sW2[(c * 4 + q) * 64 + lane];
                T = MFMA32(aw.x, H[4 * q], T); T = MFMA32(aw.y, H[4 * q + 1], T);
                T = MFMA32(aw.z, H[4 * q + 2], T); T = MFMA32(aw.w, H[4 * q + 3], T);
            }
        }
        if (vok) {
            float* o = dec + v * D;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = rowmap(r, half);
                if (ch < D) o[ch] = T[r];
            }
        }
    }
}

bool mfma_pw_supported(int F, int E, int D) { return F == 32 && E == 256 && D >= 1 && D <= 26; }

int mfma_pw_forward(const float* x, const float* w1frag, const float* w2frag, const float* b1, const float* b2, float* dec,
                    long nvox, int D, hipStream_t s)
{
    static std::once_flag once;
    std::call_once(once, [] { allow_big_lds(pw_fwd_mfma_kernel); });
    const size_t lds = (size_t)(2 * 8 * 4 * 64 * 4 + 256 + 32) * sizeof(float);
    hipLaunchKernelGGL(pw_fwd_mfma_kernel, dim3(512), dim3(256), lds, s, x, (const float4*)w1frag, (const float4*)w2frag, b1, b2, dec, nvox, D);
    return check_launch("pw_fwd_mfma");
}


// ---------------------------------------------------------------------------------------------------
// fused backward of expConv + ReLU + decConv (1x1x1): per 32-voxel tile and 32-channel hidden chunk c
//   (a) H^T_c  = W1^T X^T + b1            M = hidden, N = voxel, K = cin      (recompute; never stored)
//   (b) dH^T_c = W2 dT^T                  M = hidden, N = voxel, K = out(25)
//       dH'    = dH * [H > 0]             same register layout -> elementwise
//   (c) dX^T  += W1 dH'^T_c               K = hidden = ROW index of dH'^T: accumulator registers are the B operand
//   (d) dW1_c += X^T dH'_c                K = voxel: needs dH' with the voxel on the K index -> one in-wave LDS transpose
//   (e) dW2_c += H'^T_c dT                K = voxel: second in-wave transpose
//
// Decomposition: ONE WAVE PER HIDDEN CHUNK.  A workgroup has 8 waves = the 8 chunks of 32 hidden channels; all of
// them work on the same 32-voxel tile.  Consequences:
//   * the chunk's weight fragments (W1 for (a), W2 for (b), W1 for (c)) and its bias live in REGISTERS for the whole
//     kernel -- no LDS traffic for weights at all;
//   * a wave carries only its own dW1/dW2 chunk (32 accumulator registers), so 2 waves per SIMD fit (<= 256 VGPRs);
//   * X / dT tiles are staged once per workgroup (double-buffered: tile t+1 is loaded while tile t is computed);
//   * dX = dOut + sum over the 8 chunks: every wave leaves its partial in LDS and each wave then reduces 4 voxels;
//   * one pass over the voxels, one slab per workgroup, summed afterwards in a fixed order (fp64).
// (A first decomposition -- every wave owning all chunks of its own tiles, two launches x 4 chunks, one wave per SIMD
// at ~445 VGPRs -- reached 69 TFLOP/s; this one reaches 90.)
// ---------------------------------------------------------------------------------------------------
constexpr int PW2_WAVES = 8;
constexpr int PW2_XT = 32 * 33, PW2_DT = 32 * 27, PW2_TB = 32 * 33;

__global__ __launch_bounds__(512, 2) void pw_bwd2_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ dT, const float* __restrict__ dOut,
    const float4* __restrict__ w1kcin, const float4* __restrict__ w2kout, const float4* __restrict__ w1khch,
    const float* __restrict__ b1, float* __restrict__ dX, float* __restrict__ slabs, long nvox, int D)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Xt = lds;                                   // [2][32][33]
    float* Dt = Xt + 2 * PW2_XT;                       // [2][32][27]
    float* TbAll = Dt + 2 * PW2_DT;                    // [8 waves][2][32][33]   (dH' | H' ; dH' slot reused for the dX partial)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    float* Tb = TbAll + wave * 2 * PW2_TB;
    float* Tb2 = Tb + PW2_TB;
    const int c = wave;                                // this wave's hidden chunk

    // chunk-resident operands
    float4 w1c[4], w2c[4], w3c[4];
    float hb[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { w1c[q] = w1kcin[(c * 4 + q) * 64 + lane]; w2c[q] = w2kout[(c * 4 + q) * 64 + lane]; w3c[q] = w1khch[(c * 4 + q) * 64 + lane]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) hb[r] = b1[32 * c + rowmap(r, half)];
    f32x16 dW1, dW2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW1[r] = 0.f; dW2[r] = 0.f; }
    float bs1 = 0.f, bs2 = 0.f;

    const long ntiles = (nvox + 31) >> 5;
    // staging roles: threads 0..255 move one float4 of the X tile, threads 0..(32*D-1) (two rounds of 512) the dT tile.
    // Loads are unconditional (clamped address), zero selected afterwards.
    auto stage_load = [&](long tile, float4& xv, float& d0, float& d1) {
        const long v0 = tile * 32;
        const long nrem = nvox - v0 < 32 ? nvox - v0 : 32;
        if (tid < 256) {
            const int vv = tid >> 3;
            const long vsrc = vv < nrem ? v0 + vv : v0;
            const float4 t = reinterpret_cast<const float4*>(x + vsrc * 32)[tid & 7];
            xv = vv < nrem ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int f0 = tid, f1 = tid + 512;
        const float a0 = dT[v0 * D + (f0 < nrem * D ? f0 : 0)];
        const float a1 = dT[v0 * D + (f1 < nrem * D ? f1 : 0)];
        d0 = f0 < nrem * D ? a0 : 0.f;
        d1 = f1 < nrem * D ? a1 : 0.f;
    };
    auto stage_store = [&](int buf, const float4& xv, float d0, float d1) {
        if (tid < 256) {
            float* d = Xt + buf * PW2_XT + (tid >> 3) * 33 + (tid & 7) * 4;
            d[0] = xv.x; d[1] = xv.y; d[2] = xv.z; d[3] = xv.w;
        }
        const int f0 = tid, f1 = tid + 512;
        if (f0 < 32 * D) { const int vv = f0 / D; Dt[buf * PW2_DT + vv * 27 + (f0 - vv * D)] = d0; }
        if (f1 < 32 * D) { const int vv = f1 / D; Dt[buf * PW2_DT + vv * 27 + (f1 - vv * D)] = d1; }
    };

    long tile = blockIdx.x;
    int buf = 0;
    {
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f); float d0 = 0.f, d1 = 0.f;
        if (tile < ntiles) { stage_load(tile, xv, d0, d1); stage_store(0, xv, d0, d1); }
    }
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        __syncthreads();                               // tile `tile` is staged in buffer `buf`; previous dX reduce is done
        const long tnext = tile + gridDim.x;
        float4 nxv = make_float4(0.f, 0.f, 0.f, 0.f); float nd0 = 0.f, nd1 = 0.f;
        if (tnext < ntiles) stage_load(tnext, nxv, nd0, nd1);          // in flight during this tile's MFMAs
        const long v0 = tile * 32;
        // dOut for the reduce step (this wave reduces voxels 4*wave .. 4*wave+3; lane -> (voxel, cin))
        const int rv0 = 4 * wave + (lane >> 5), rv1 = rv0 + 2;
        const bool rok0 = v0 + rv0 < nvox, rok1 = v0 + rv1 < nvox;
        const float do0 = dOut[(rok0 ? v0 + rv0 : v0) * 32 + col];
        const float do1 = dOut[(rok1 ? v0 + rv1 : v0) * 32 + col];

        const float* Xb = Xt + buf * PW2_XT;
        const float* Db = Dt + buf * PW2_DT;
        float xs[16], dts[13], xa[16], dtb[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            xs[s] = Xb[col * 33 + 16 * half + s];                                       // B of (a): X[vox col][16*half + s]
            xa[s] = Xb[(2 * s + half) * 33 + col];                                      // A of (d): X[vox 2s+half][cin col]
            dtb[s] = col < D ? Db[(2 * s + half) * 27 + col] : 0.f;                     // B of (e): dT[vox 2s+half][out col]
        }
#pragma unroll
        for (int s = 0; s < 13; ++s) dts[s] = (13 * half + s) < D ? Db[col * 27 + 13 * half + s] : 0.f;   // B of (b)
        if (c == 0) {
            float t2 = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) t2 += dtb[s];
            bs2 += t2;
        }
        f32x16 H, dH, dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { H[r] = hb[r]; dH[r] = 0.f; dx[r] = 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                                   // (a) H^T = W1^T X^T + b1
            H = MFMA32(w1c[q].x, xs[4 * q], H); H = MFMA32(w1c[q].y, xs[4 * q + 1], H);
            H = MFMA32(w1c[q].z, xs[4 * q + 2], H); H = MFMA32(w1c[q].w, xs[4 * q + 3], H);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                                   // (b) dH^T = W2 dT^T
            dH = MFMA32(w2c[q].x, dts[4 * q], dH);
            if (4 * q + 1 < 13) dH = MFMA32(w2c[q].y, dts[(4 * q + 1) % 13], dH);
            if (4 * q + 2 < 13) dH = MFMA32(w2c[q].z, dts[(4 * q + 2) % 13], dH);
            if (4 * q + 3 < 13) dH = MFMA32(w2c[q].w, dts[(4 * q + 3) % 13], dH);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { dH[r] = H[r] > 0.f ? dH[r] : 0.f; H[r] = fmaxf(H[r], 0.f); }
        // transposes: lane (voxel col, half) owns hidden rowmap(r, half).  LDS operations of one wave execute in order,
        // so only the COMPILER must be kept from hoisting the reads above the writes (memory clobber).
#pragma unroll
        for (int r = 0; r < 16; ++r) { Tb[col * 33 + rowmap(r, half)] = dH[r]; Tb2[col * 33 + rowmap(r, half)] = H[r]; }
        asm volatile("" ::: "memory");
        float tr[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) tr[s] = Tb[(2 * s + half) * 33 + col];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                                   // (c) dX^T partial = W1 dH'^T  (registers only)
            dx = MFMA32(w3c[q].x, dH[4 * q], dx); dx = MFMA32(w3c[q].y, dH[4 * q + 1], dx);
            dx = MFMA32(w3c[q].z, dH[4 * q + 2], dx); dx = MFMA32(w3c[q].w, dH[4 * q + 3], dx);
        }
        float tr2[16];                                                                  // requested under (c)/(d), consumed by (e)
#pragma unroll
        for (int s = 0; s < 16; ++s) tr2[s] = Tb2[(2 * s + half) * 33 + col];
        __builtin_amdgcn_sched_barrier(0);
        {
            float t1 = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) { t1 += tr[s]; dW1 = MFMA32(xa[s], tr[s], dW1); }              // (d) dW1_c += X^T dH'_c
            bs1 += t1;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) dW2 = MFMA32(tr2[s], dtb[s], dW2);                                  // (e) dW2_c += H'^T_c dT
        __builtin_amdgcn_sched_barrier(0);
        // this wave's dX partial -> its dH' slot ([voxel][33]); next tile's data -> the other staging buffer
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) Tb[col * 33 + rowmap(r, half)] = dx[r];
        if (tnext < ntiles) stage_store(buf ^ 1, nxv, nd0, nd1);
        __syncthreads();                               // all 8 partials (and the next tile) are in LDS
        {
            float s0 = do0, s1 = do1;
#pragma unroll
            for (int j = 0; j < PW2_WAVES; ++j) {
                const float* P = TbAll + j * 2 * PW2_TB;
                s0 += P[rv0 * 33 + col];
                s1 += P[rv1 * 33 + col];
            }
            if (rok0) dX[(v0 + rv0) * 32 + col] = s0;
            if (rok1) dX[(v0 + rv1) * 32 + col] = s1;
        }
    }
    // one slab per workgroup: [dW1 32x256 | dW2 256xD | db1 256 | db2 D]
    const long slab_floats = 8192 + 256 * (long)D + 256 + D;
    float* sl = slabs + (long)blockIdx.x * slab_floats;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rw = rowmap(r, half);
        sl[(long)rw * 256 + 32 * c + col] = dW1[r];                                    // [cin][hidden]
        if (col < D) sl[8192 + (long)(32 * c + rw) * D + col] = dW2[r];                // [hidden][out]
    }
    const float b = bs1 + __shfl_xor(bs1, 32, 64);
    if (half == 0) sl[8192 + 256 * (long)D + 32 * c + col] = b;
    if (c == 0) {
        const float b2s = bs2 + __shfl_xor(bs2, 32, 64);
        if (half == 0 && col < D) sl[8192 + 256 * (long)D + 256 + col] = b2s;
    }
}

static const int PW_BWD_GRID = 256;

int mfma_pw_backward_reduce(const float* slabs, int D, float* dW1, float* dW2, float* db1, float* db2, hipStream_t s)
{
    const long slab_floats = 8192 + 256 * (long)D + 256 + D;
    SlabSumJob jobs[4] = {{slabs, dW1, slab_floats, 8192, PW_BWD_GRID}, {slabs + 8192, dW2, slab_floats, 256 * D, PW_BWD_GRID},
                          {slabs + 8192 + 256 * (long)D, db1, slab_floats, 256, PW_BWD_GRID}, {slabs + 8192 + 256 * (long)D + 256, db2, slab_floats, D, PW_BWD_GRID}};
    return slab_sum_later(s, jobs, 4);
}
int mfma_pw_backward_grid() { return PW_BWD_GRID; }

size_t mfma_pw_backward_slab_floats(int D) { return (size_t)PW_BWD_GRID * (8192 + 256 * (size_t)D + 256 + D); }

int mfma_pw_backward(const float* x, const float* dT, const float* dOut, const float* w1kcin, const float* w2kout,
                     const float* w1khch, const float* b1, float* dX, float* dW1, float* dW2, float* db1, float* db2,
                     float* slabs, long nvox, int D, hipStream_t s)
{
    static std::once_flag once;
    std::call_once(once, [] { allow_big_lds(pw_bwd2_mfma_kernel); });
    const size_t lds = (size_t)(2 * PW2_XT + 2 * PW2_DT + PW2_WAVES * 2 * PW2_TB) * sizeof(float);
    const long slab_floats = 8192 + 256 * (long)D + 256 + D;
    hipLaunchKernelGGL(pw_bwd2_mfma_kernel, dim3(PW_BWD_GRID), dim3(64 * PW2_WAVES), lds, s, x, dT, dOut, (const float4*)w1kcin,
                       (const float4*)w2kout, (const float4*)w1khch, b1, dX, slabs, nvox, D);
    int rc = check_launch("pw_bwd2_mfma");
    if (rc) return rc;
    return mfma_pw_backward_reduce(slabs, D, dW1, dW2, db1, db2, s);
}

}  // namespace probav
