// Generic direct (VALU) stride-1 convolution over (H, W, T) for gfx950: forward / backward-data
// (same kernel, flipped+transposed weights) and backward-filter.  Any geometry the reference's
// graph produces (models/modelsTF.py:45-203): 3x3x3 / 3x3(x1) / 1x1x1 kernels, zero 'same' padding,
// 'valid', and tf.pad(reflect) on H,W folded into the indexing.
//
// This is the shape-agnostic path: it backs the small layers (mainConv1, residConv*, upscaleConv1),
// unusual configurations, and serves as the on-device cross-check for the MFMA kernels
// (kernels_mfma.hip) that carry the 12 residual blocks and the reducers.
//
// Forward: one thread per output voxel, all COUT_T output channels in registers; the filter taps are
// wave-uniform, so the compiler fetches them through the scalar cache (s_load) and the FMAs take
// them as SGPR operands -- filter reuse costs no vector memory traffic at all.
#include "probav_common.h"
#include "x6_device.h"

namespace probav {

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

template <int COUT_T, int VEC>
__global__ __launch_bounds__(256) void conv_direct_fwd_kernel(
    ConvGeom g, const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ y)
{
    const long nvox = (long)g.N * g.Ho * g.Wo * g.To;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvox) return;
    const int co0 = blockIdx.y * COUT_T;
    long r = v;
    const int t = (int)(r % g.To); r /= g.To;
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int h = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);

    float acc[COUT_T];
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) acc[j] = 0.f;

    for (int a = 0; a < g.kh; ++a) {
        int ih = h + a - g.ph;
        if (g.reflect_hw) ih = reflect_idx(ih, g.Hi);
        else if (ih < 0 || ih >= g.Hi) continue;
        for (int b = 0; b < g.kw; ++b) {
            int iw = wo + b - g.pw;
            if (g.reflect_hw) iw = reflect_idx(iw, g.Wi);
            else if (iw < 0 || iw >= g.Wi) continue;
            for (int c = 0; c < g.kt; ++c) {
                int it = t + c - g.pt;
                if (g.reflect_t) it = reflect_idx(it, g.Ti);
                else if (it < 0 || it >= g.Ti) continue;
                const long vin = (((long)n * g.Hi + ih) * g.Wi + iw) * g.Ti + it;
                const float* xp = x + vin * g.Cin;
                const float* gp = gate ? gate + vin * g.Cin : nullptr;
                const float* wp = w + (long)((a * g.kw + b) * g.kt + c) * g.Cin * g.Cout + co0;
                for (int ci = 0; ci < g.Cin; ci += VEC) {
                    float xv[VEC];
                    if constexpr (VEC == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(xp + ci);
                        xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
                        if (gp) {
                            const float4 m = *reinterpret_cast<const float4*>(gp + ci);
                            xv[0] = m.x > 0.f ? xv[0] : 0.f; xv[1] = m.y > 0.f ? xv[1] : 0.f;
                            xv[2] = m.z > 0.f ? xv[2] : 0.f; xv[3] = m.w > 0.f ? xv[3] : 0.f;
                        }
                    } else {
                        xv[0] = xp[ci];
                        if (gp) xv[0] = gp[ci] > 0.f ? xv[0] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < VEC; ++u) {
                        const float* wr = wp + (long)(ci + u) * g.Cout;
#pragma unroll
                        for (int j = 0; j < COUT_T; ++j) acc[j] = fmaf(xv[u], wr[j], acc[j]);
                    }
                }
            }
        }
    }
    float* yp = y + v * g.Cout + co0;
    const float* sp = skip ? skip + v * g.Cout + co0 : nullptr;
#pragma unroll
    for (int j = 0; j < COUT_T; ++j) {
        float o = acc[j] + (bias ? bias[co0 + j] : 0.f);
        if (g.relu) o = fmaxf(o, 0.f);
        if (sp) o += sp[j];
        yp[j] = o;
    }
}

template <int COUT_T>
static int launch_fwd(const ConvGeom& g, const float* x, const float* gate, const float* w, const float* bias,
                      const float* skip, float* y, hipStream_t s)
{
    const long nvox = (long)g.N * g.Ho * g.Wo * g.To;
    dim3 grid((unsigned)((nvox + 255) / 256), (unsigned)(g.Cout / COUT_T));
    const bool vec4 = (g.Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                      (!gate || (reinterpret_cast<uintptr_t>(gate) & 15) == 0);
    if (vec4) hipLaunchKernelGGL((conv_direct_fwd_kernel<COUT_T, 4>), grid, dim3(256), 0, s, g, x, gate, w, bias, skip, y);
    else      hipLaunchKernelGGL((conv_direct_fwd_kernel<COUT_T, 1>), grid, dim3(256), 0, s, g, x, gate, w, bias, skip, y);
    return check_launch("conv_direct_fwd");
}

// ---------------------------------------------------------------------------------------------------
// mainConv1 (models/modelsTF.py:23-24): ONE input channel -> 32, 3x3x3, zero pads of 1, ReLU.  0.5 GMAC against 71 MB of output: bound
// by the store stream, so no matrix unit: a thread keeps the 27 x 4 filter values of its four output channels in registers and walks
// the voxels of an output row; eight lanes x 16 bytes = one voxel's 128-byte row.  Input rows h-1..h+1 sit zero-padded in LDS.
// Also leaves the per-sample amax of its output (the next layer's H3 scale) instead of a separate pass over the 71 MB.
// ---------------------------------------------------------------------------------------------------
constexpr int C1_ROWS = 4;                  // consecutive (sample, row) pairs per workgroup
__global__ __launch_bounds__(256) void conv3_cin1_fwd_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, unsigned* __restrict__ amax)
{
    extern __shared__ float c1_in[];                                          // [3][W + 2][T + 2]
    const int tid = threadIdx.x, cg = tid & 7, vs = tid >> 3;
    const int Wp = g.Wi + 2, Tp = g.Ti + 2, nin = 3 * Wp * Tp, nv = g.Wo * g.To;
    float4 wt[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wt[k] = *reinterpret_cast<const float4*>(w + k * 32 + 4 * cg);
    const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned mT = 0xffffffffu / (unsigned)g.To + 1u, mTp = 0xffffffffu / (unsigned)Tp + 1u, mWT = 0xffffffffu / (unsigned)(Wp * Tp) + 1u;
    const long nrows = (long)g.N * g.Ho;
#pragma unroll 1
    for (int r = 0; r < C1_ROWS; ++r) {
        const long R = (long)blockIdx.x * C1_ROWS + r;
        if (R >= nrows) break;
        const int n = (int)(R / g.Ho), h = (int)(R - (long)n * g.Ho);
        __syncthreads();                                                       // the previous row's readers are done
        for (int i = tid; i < nin; i += 256) {
            const int dh = (int)__umulhi((unsigned)i, mWT), rem = i - dh * Wp * Tp;
            const int wp = (int)__umulhi((unsigned)rem, mTp), tp = rem - wp * Tp;
            const int ih = h - 1 + dh, iw = wp - 1, it = tp - 1;
            const bool ok = ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi && it >= 0 && it < g.Ti;
            c1_in[i] = ok ? x[(((long)n * g.Hi + ih) * g.Wi + iw) * g.Ti + it] : 0.f;
        }
        __syncthreads();
        float omax = 0.f;
        float* yrow = y + ((long)n * g.Ho + h) * nv * 32;
        for (int v = vs; v < nv; v += 32) {
            const int wo = (int)__umulhi((unsigned)v, mT), t = v - wo * g.To;
            const float* ip = c1_in + wo * Tp + t;
            float4 acc = b4;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        const float xv = ip[(dh * Wp + dw) * Tp + dt];
                        const float4 q = wt[(dh * 3 + dw) * 3 + dt];
                        acc.x = fmaf(xv, q.x, acc.x); acc.y = fmaf(xv, q.y, acc.y); acc.z = fmaf(xv, q.z, acc.z); acc.w = fmaf(xv, q.w, acc.w);
                    }
            if (g.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
            *reinterpret_cast<float4*>(yrow + (long)v * 32 + 4 * cg) = acc;
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        }
        if (amax) amax_commit(omax, amax + n);
    }
}

bool conv3d_cin1_forward_supported(const ConvGeom& g)
{
    return g.Cin == 1 && g.Cout == 32 && g.kh == 3 && g.kw == 3 && g.kt == 3 && g.ph == 1 && g.pw == 1 && g.pt == 1 && !g.reflect_hw && !g.reflect_t &&
           g.Ho == g.Hi && g.Wo == g.Wi && g.To == g.Ti && g.To >= 2 && (g.Wi + 2) * (g.Ti + 2) * 3 * sizeof(float) <= 48 * 1024;
}
// amax: per-sample slots of the output (may be null)
int conv3d_cin1_forward(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, unsigned* amax, hipStream_t s)
{
    if (!conv3d_cin1_forward_supported(g)) { set_error("conv3d_cin1_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    const long nrows = (long)g.N * g.Ho;
    const size_t lds = (size_t)3 * (g.Wi + 2) * (g.Ti + 2) * sizeof(float);
    hipLaunchKernelGGL(conv3_cin1_fwd_kernel, dim3((unsigned)((nrows + C1_ROWS - 1) / C1_ROWS)), dim3(256), lds, s, g, x, w, bias, y, amax);
    return check_launch("conv3_cin1_fwd");
}

int conv3d_direct_forward(const ConvGeom& g, const float* x, const float* gate, const float* w,
                          const float* bias, const float* skip, float* y, hipStream_t s)
{
    if (g.N <= 0) return PROBAV_OK;
    if (g.Cout % 32 == 0) return launch_fwd<32>(g, x, gate, w, bias, skip, y, s);
    if (g.Cout % 25 == 0) return launch_fwd<25>(g, x, gate, w, bias, skip, y, s);
    if (g.Cout % 9 == 0)  return launch_fwd<9>(g, x, gate, w, bias, skip, y, s);
    if (g.Cout % 4 == 0)  return launch_fwd<4>(g, x, gate, w, bias, skip, y, s);
    return launch_fwd<1>(g, x, gate, w, bias, skip, y, s);
}

// ---------------------------------------------------------------------------------------------------
// backward-filter.  256 threads = 32 output channels x 8 input-channel groups; every thread keeps
// TAPS x CI_PER partial sums in registers while the block walks its chunk of output voxels (all
// threads on the same voxel => bounds tests are scalar and the x loads are 32-lane broadcasts).
// Per-chunk partials go to scratch and are summed in a fixed order by reduce_partials_kernel, so the
// result is bitwise reproducible (no float atomics).
// ---------------------------------------------------------------------------------------------------
template <int KH, int KW, int KT, int CI_PER>
__global__ __launch_bounds__(256) void conv_direct_wgrad_kernel(
    ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gate,
    float* __restrict__ partial, float* __restrict__ partial_b, long vox_per_chunk)
{
    constexpr int TAPS = KH * KW * KT;
    const int tid = threadIdx.x;
    const int co = blockIdx.y * 32 + (tid & 31);
    const bool co_ok = co < g.Cout;
    const int ci0 = (blockIdx.z * 8 + (tid >> 5)) * CI_PER;
    const long nvox = (long)g.N * g.Ho * g.Wo * g.To;
    const long v0 = (long)blockIdx.x * vox_per_chunk;
    const long v1 = v0 + vox_per_chunk < nvox ? v0 + vox_per_chunk : nvox;

    float acc[TAPS][CI_PER];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int k = 0; k < CI_PER; ++k) acc[tp][k] = 0.f;
    float accb = 0.f;
    const bool vec4 = (CI_PER % 4 == 0) && (g.Cin % 4 == 0) && (ci0 + CI_PER <= g.Cin);

    for (long v = v0; v < v1; ++v) {
        long r = v;
        const int t = (int)(r % g.To); r /= g.To;
        const int wo = (int)(r % g.Wo); r /= g.Wo;
        const int h = (int)(r % g.Ho);
        const int n = (int)(r / g.Ho);
        float d = 0.f;
        if (co_ok) {
            d = dy[v * g.Cout + co];
            if (gate) d = gate[v * g.Cout + co] > 0.f ? d : 0.f;
        }
        accb += d;
#pragma unroll
        for (int a = 0; a < KH; ++a) {
            int ih = h + a - g.ph;
            if (g.reflect_hw) ih = reflect_idx(ih, g.Hi);
            else if (ih < 0 || ih >= g.Hi) continue;
#pragma unroll
            for (int b = 0; b < KW; ++b) {
                int iw = wo + b - g.pw;
                if (g.reflect_hw) iw = reflect_idx(iw, g.Wi);
                else if (iw < 0 || iw >= g.Wi) continue;
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    int it = t + c - g.pt;
                    if (g.reflect_t) it = reflect_idx(it, g.Ti);
                    else if (it < 0 || it >= g.Ti) continue;
                    const float* xp = x + ((((long)n * g.Hi + ih) * g.Wi + iw) * g.Ti + it) * g.Cin + ci0;
                    const int tp = (a * KW + b) * KT + c;
                    bool done = false;
                    if constexpr (CI_PER % 4 == 0) {
                        if (vec4) {
#pragma unroll
                            for (int k = 0; k < CI_PER; k += 4) {
                                const float4 q = *reinterpret_cast<const float4*>(xp + k);
                                acc[tp][k] = fmaf(q.x, d, acc[tp][k]);
                                acc[tp][k + 1] = fmaf(q.y, d, acc[tp][k + 1]);
                                acc[tp][k + 2] = fmaf(q.z, d, acc[tp][k + 2]);
                                acc[tp][k + 3] = fmaf(q.w, d, acc[tp][k + 3]);
                            }
                            done = true;
                        }
                    }
                    if (!done) {
#pragma unroll
                        for (int k = 0; k < CI_PER; ++k)
                            if (ci0 + k < g.Cin) acc[tp][k] = fmaf(xp[k], d, acc[tp][k]);
                    }
                }
            }
        }
    }
    if (co_ok) {
        float* pp = partial + (long)blockIdx.x * TAPS * g.Cin * g.Cout;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int k = 0; k < CI_PER; ++k)
                if (ci0 + k < g.Cin) pp[((long)tp * g.Cin + ci0 + k) * g.Cout + co] = acc[tp][k];
        if (blockIdx.z == 0 && (tid >> 5) == 0) partial_b[(long)blockIdx.x * g.Cout + co] = accb;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             long n, int chunks)
{
    // 32 elements x 8 interleaved partial sums per block, fixed order (bitwise reproducible); fp64 across chunks:
    // the filter gradient is a sum with heavy cancellation
    const int e = threadIdx.x & 31, part = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + e;
    __shared__ double red[8][32];
    double a0 = 0.0, a1 = 0.0;
    if (i < n) {
        int c = part;
        for (; c + 8 < chunks; c += 16) { a0 += (double)partial[(long)c * n + i]; a1 += (double)partial[(long)(c + 8) * n + i]; }
        if (c < chunks) a0 += (double)partial[(long)c * n + i];
    }
    red[part][e] = a0 + a1;
    __syncthreads();
    if (part != 0 || i >= n) return;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][e];
    out[i] = (float)t;
}


// ---------------------------------------------------------------------------------------------------
// backward-filter of the 2-D residual path (3x3x1 'valid' convolutions on [N, H, W, 1, C <= 9]: residConv1..3).  The generic
// kernel above walks a few voxels per block with dependent global loads (latency-bound: 0.17 ms for 37 MFLOP).  Here one
// workgroup takes half a patch: x rows and gated dy rows are staged in LDS once, thread (tap, ci, co) keeps ONE accumulator and
// walks the half-patch's output voxels with two LDS reads (both broadcasts inside a (tap, ci) / co group) and one fma each.
// One slab per workgroup, summed in fixed order by reduce_partials_kernel.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wgrad2d_small_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gate, float* __restrict__ partial,
                                                            float* __restrict__ partial_b, int halves)
{
    extern __shared__ float lds2d[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = blockIdx.x / halves, part = blockIdx.x - n * halves;
    const int r0 = part * g.Ho / halves, r1 = (part + 1) * g.Ho / halves;      // output rows of this workgroup
    const int nrow_o = r1 - r0, nrow_i = nrow_o + 2;
    float* xs = lds2d;                                       // [nrow_i][Wi][Cin]
    float* ds = xs + nrow_i * g.Wi * g.Cin;                  // [nrow_o][Wo][Cout], already gated
    const float* xsrc = x + ((long)n * g.Hi + r0) * g.Wi * g.Cin;
    for (int i = tid; i < nrow_i * g.Wi * g.Cin; i += nthr) xs[i] = xsrc[i];
    const long dbase = ((long)n * g.Ho + r0) * g.Wo * g.Cout;
    for (int i = tid; i < nrow_o * g.Wo * g.Cout; i += nthr) {
        float d = dy[dbase + i];
        if (gate) d = gate[dbase + i] > 0.f ? d : 0.f;
        ds[i] = d;
    }
    __syncthreads();
    const int nout = 9 * g.Cin * g.Cout;
    const long slab = (long)nout + g.Cout;
    float* pp = partial + (long)blockIdx.x * nout;
    if (tid < nout) {
        const int co = tid % g.Cout, rest = tid / g.Cout, ci = rest % g.Cin, tap = rest / g.Cin;
        const int a = tap / 3, b = tap - 3 * a;
        const float* xp = xs + (a * g.Wi + b) * g.Cin + ci;
        const float* dp = ds + co;
        float acc = 0.f;
        for (int h = 0; h < nrow_o; ++h)
            for (int w = 0; w < g.Wo; ++w) acc = fmaf(xp[(h * g.Wi + w) * g.Cin], dp[(h * g.Wo + w) * g.Cout], acc);
        pp[tid] = acc;                                       // == [(tap * Cin + ci) * Cout + co]
    } else if (tid < nout + g.Cout) {
        const int co = tid - nout;
        float acc = 0.f;
        for (int i = 0; i < nrow_o * g.Wo; ++i) acc += ds[i * g.Cout + co];
        partial_b[(long)blockIdx.x * g.Cout + co] = acc;
    }
    (void)slab;
}

static bool small2d(const ConvGeom& g, int& halves, size_t& lds)
{
    if (g.kh != 3 || g.kw != 3 || g.kt != 1 || g.Ti != 1 || g.To != 1 || g.ph || g.pw || g.pt || g.reflect_hw) return false;
    if (9 * g.Cin * g.Cout + g.Cout > 1024 || g.Ho < 2) return false;
    halves = 2;
    const int nrow_o = (g.Ho + 1) / 2 + 1;
    lds = ((size_t)(nrow_o + 2) * g.Wi * g.Cin + (size_t)nrow_o * g.Wo * g.Cout) * sizeof(float);
    return lds <= 64 * 1024;
}

static void wgrad_plan(const ConvGeom& g, int& ci_per, int& gy, int& gz, int& chunks, long& vpc)
{
    const bool k3d = (g.kh == 3 && g.kw == 3 && g.kt == 3);
    const bool k2d = (g.kh == 3 && g.kw == 3 && g.kt == 1);
    if (g.kh == 5) ci_per = 1;
    else if (k3d) ci_per = g.Cin >= 4 ? 4 : 1;
    else if (k2d) ci_per = g.Cin >= 4 ? 4 : 1;
    else ci_per = g.Cin >= 256 ? 32 : (g.Cin >= 4 ? 4 : 1);
    gy = (g.Cout + 31) / 32;
    gz = (g.Cin + 8 * ci_per - 1) / (8 * ci_per);
    const long nvox = (long)g.N * g.Ho * g.Wo * g.To;
    long want = 2048 / ((long)gy * gz);
    if (want < 1) want = 1;
    if (want > nvox) want = nvox;
    vpc = (nvox + want - 1) / want;
    if (vpc < 1) vpc = 1;
    chunks = (int)((nvox + vpc - 1) / vpc);
}

// ---------------------------------------------------------------------------------------------------
// Backward-filter of a ONE-input-channel 'same' 3x3x3 layer with 32 outputs (mainConv1): dw[tap][co] = sum_v x[v + tap] dy[v][co].
// Too thin for a matrix kernel (K = 27 rows, one of them per tap): the fp32-MFMA backward-filter kernel took 132 us on it, the
// 142 MB of dY + ReLU mask it reads cost ~35 us.  Persistent workgroups walk (patch, row) tiles: the three input rows of a tile sit
// zero-padded in LDS (3 KB), 16 voxel streams (8 waves x 2 half-waves) run over the row's voxels with lane = output channel, so
// dY / mask loads are 128-byte rows and every x value is an LDS broadcast; 27 + 1 accumulators per thread, one slab per workgroup.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void wgrad_cin1_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ gate, float* __restrict__ partial,
                                                        float* __restrict__ partial_b)
{
    extern __shared__ float sx[];                                  // [3 rows][W + 2][T + 2], then the exchange area [8 waves][28][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, co = lane & 31;
    const int Wp = g.Wi + 2, Tp = g.Ti + 2, rowf = Wp * Tp;
    float* xch = sx + 3 * rowf;
    float acc[27], bsum = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    const int nvr = g.Wo * g.To, ntile = g.N * g.Ho;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int n = tile / g.Ho, h = tile - n * g.Ho;
        __syncthreads();
        for (int i = tid; i < 3 * rowf; i += 512) {
            const int r = i / rowf, rem = i - r * rowf, w = rem / Tp - 1, t = rem - (rem / Tp) * Tp - 1, ih = h + r - 1;
            const bool ok = ih >= 0 && ih < g.Hi && w >= 0 && w < g.Wi && t >= 0 && t < g.Ti;
            sx[i] = ok ? x[(((long)n * g.Hi + ih) * g.Wi + w) * g.Ti + t] : 0.f;
        }
        __syncthreads();
        const long ob = ((long)n * g.Ho + h) * nvr;
        for (int vi = 2 * wave + half; vi < nvr; vi += 16) {
            const int w = vi / g.To, t = vi - w * g.To;
            float d = dy[(ob + vi) * 32 + co];
            if (gate) d = gate[(ob + vi) * 32 + co] > 0.f ? d : 0.f;
            bsum += d;
            const float* px = sx + w * Tp + t;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) acc[(dh * 3 + dw) * 3 + dt] = fmaf(px[dh * rowf + dw * Tp + dt], d, acc[(dh * 3 + dw) * 3 + dt]);
        }
    }
    // the 16 streams of an output channel meet in a fixed order: half-waves by shuffle, waves through LDS
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] += __shfl_xor(acc[k], 32, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) xch[(wave * 28 + k) * 32 + co] = acc[k];
        xch[(wave * 28 + 27) * 32 + co] = bsum;
    }
    __syncthreads();
    for (int i = tid; i < 28 * 32; i += 512) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += xch[w * 28 * 32 + i];
        if (i < 27 * 32) partial[(long)blockIdx.x * 27 * 32 + i] = v;
        else partial_b[(long)blockIdx.x * 32 + (i - 27 * 32)] = v;
    }
}
static bool cin1_wgrad(const ConvGeom& g)
{
    return g.Cin == 1 && g.Cout == 32 && g.kh == 3 && g.kw == 3 && g.kt == 3 && g.ph == 1 && g.pw == 1 && g.pt == 1 && !g.reflect_hw && !g.reflect_t &&
           g.Ho == g.Hi && g.Wo == g.Wi && g.To == g.Ti && (size_t)(3 * (g.Wi + 2) * (g.Ti + 2) + 8 * 28 * 32) * sizeof(float) <= 64 * 1024;
}
static int cin1_grid(const ConvGeom& g) { const int nt = g.N * g.Ho; return nt < 512 ? nt : 512; }
bool conv3d_direct_wgrad_is_tuned(const ConvGeom& g) { return cin1_wgrad(g); }

size_t wgrad_partial_floats(const ConvGeom& g)
{
    int ci_per, gy, gz, chunks; long vpc;
    wgrad_plan(g, ci_per, gy, gz, chunks, vpc);
    const size_t K = (size_t)g.kh * g.kw * g.kt * g.Cin;
    int halves; size_t lds;
    if (small2d(g, halves, lds) && g.N * halves > chunks) chunks = g.N * halves;
    if (cin1_wgrad(g) && cin1_grid(g) > chunks) chunks = cin1_grid(g);
    return (size_t)chunks * (K * g.Cout + g.Cout);
}

int conv3d_direct_wgrad(const ConvGeom& g, const float* x, const float* dy, const float* gate,
                        float* dw, float* db, float* partial, hipStream_t s)
{
    int ci_per, gy, gz, chunks; long vpc;
    wgrad_plan(g, ci_per, gy, gz, chunks, vpc);
    const long K = (long)g.kh * g.kw * g.kt * g.Cin;
    if (cin1_wgrad(g)) {
        const int slabs = cin1_grid(g);
        float* pb = partial + (size_t)slabs * 27 * 32;
        const size_t lds = (size_t)(3 * (g.Wi + 2) * (g.Ti + 2) + 8 * 28 * 32) * sizeof(float);
        hipLaunchKernelGGL(wgrad_cin1_kernel, dim3((unsigned)slabs), dim3(512), lds, s, g, x, dy, gate, partial, pb);
        int rc = check_launch("wgrad_cin1");
        if (rc) return rc;
        hipStream_t rs = reduce_fork(s);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((27 * 32 + 31) / 32)), dim3(256), 0, rs, partial, dw, (long)27 * 32, slabs);
        if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, rs, pb, db, (long)32, slabs);
        return check_launch("reduce_partials");
    }
    {
        int halves; size_t lds;
        if (small2d(g, halves, lds)) {
            const int slabs = g.N * halves;
            float* pb = partial + (size_t)slabs * K * g.Cout;
            const int nthr = (int)((9 * g.Cin * g.Cout + g.Cout + 63) / 64 * 64);
            hipLaunchKernelGGL(wgrad2d_small_kernel, dim3((unsigned)slabs), dim3((unsigned)nthr), lds, s, g, x, dy, gate, partial, pb, halves);
            int rc = check_launch("wgrad2d_small");
            if (rc) return rc;
            const long nw = K * g.Cout;
            hipStream_t rs = reduce_fork(s);
            hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((nw + 31) / 32)), dim3(256), 0, rs, partial, dw, nw, slabs);
            if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((g.Cout + 31) / 32)), dim3(256), 0, rs, pb, db, (long)g.Cout, slabs);
            return check_launch("reduce_partials");
        }
    }
    float* partial_b = partial + (size_t)chunks * K * g.Cout;
    dim3 grid((unsigned)chunks, (unsigned)gy, (unsigned)gz), block(256);
    const bool k3d = (g.kh == 3 && g.kw == 3 && g.kt == 3);
    const bool k2d = (g.kh == 3 && g.kw == 3 && g.kt == 1);
    const bool k1 = (g.kh == 1 && g.kw == 1 && g.kt == 1);
    const bool k5d = (g.kh == 5 && g.kw == 5 && g.kt == 5);
#define PROBAV_WG(KH, KW, KT, CP) \
    hipLaunchKernelGGL((conv_direct_wgrad_kernel<KH, KW, KT, CP>), grid, block, 0, s, g, x, dy, gate, partial, partial_b, vpc)
    if (k5d) PROBAV_WG(5, 5, 5, 1);
    else if (k3d && ci_per == 4) PROBAV_WG(3, 3, 3, 4);
    else if (k3d) PROBAV_WG(3, 3, 3, 1);
    else if (k2d && ci_per == 4) PROBAV_WG(3, 3, 1, 4);
    else if (k2d) PROBAV_WG(3, 3, 1, 1);
    else if (k1 && ci_per == 32) PROBAV_WG(1, 1, 1, 32);
    else if (k1 && ci_per == 4) PROBAV_WG(1, 1, 1, 4);
    else if (k1) PROBAV_WG(1, 1, 1, 1);
    else { set_error("conv3d_direct_wgrad: unsupported kernel size", hipSuccess); return PROBAV_EINVAL; }
#undef PROBAV_WG
    int rc = check_launch("conv_direct_wgrad");
    if (rc) return rc;
    const long nw = K * g.Cout;
    hipStream_t rs = reduce_fork(s);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((nw + 31) / 32)), block, 0, rs, partial, dw, nw, chunks);
    if (db) hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((g.Cout + 31) / 32)), block, 0, rs, partial_b, db, (long)g.Cout, chunks);
    return check_launch("reduce_partials");
}

}  // namespace probav
