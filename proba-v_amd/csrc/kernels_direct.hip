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
#include <mutex>
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
    float omax = 0.f;                                                          // of the rows of ONE sample: committed when the sample changes and at the end
    int n_prev = -1;
#pragma unroll 1
    for (int r = 0; r < C1_ROWS; ++r) {
        const long R = (long)blockIdx.x * C1_ROWS + r;
        if (R >= nrows) break;
        const int n = (int)(R / g.Ho), h = (int)(R - (long)n * g.Ho);
        if (n != n_prev) { if (amax && n_prev >= 0) amax_commit(omax, amax + n_prev); omax = 0.f; n_prev = n; }
        __syncthreads();                                                       // the previous row's readers are done
        for (int i = tid; i < nin; i += 256) {
            const int dh = (int)__umulhi((unsigned)i, mWT), rem = i - dh * Wp * Tp;
            const int wp = (int)__umulhi((unsigned)rem, mTp), tp = rem - wp * Tp;
            const int ih = h - 1 + dh, iw = wp - 1, it = tp - 1;
            const bool ok = ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi && it >= 0 && it < g.Ti;
            c1_in[i] = ok ? x[(((long)n * g.Hi + ih) * g.Wi + iw) * g.Ti + it] : 0.f;
        }
        __syncthreads();
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
    }
    if (amax && n_prev >= 0) amax_commit(omax, amax + n_prev);                 // (one call per wave and sample run instead of one per row: 11 264 -> 3 300 atomics on the slots' four lines)
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
// upscaleConv1 (models/modelsTF.py:163-164): valid 3x3x3, 32 -> s^2 = 9 channels, on the last reducer's [18][18][3] output -- the depth
// collapses to 1.  0.26 GMAC per batch of 128, 0.2 % of the step's work: the row-tile MFMA kernel spent 44 us on it (a 32x9 product in
// 32x32 tiles, four percent busy) and the generic VALU kernel 77 us on its backward-data.  Two small VALU kernels, one workgroup per
// (sample, output row), input rows and the whole filter in LDS:
//   forward        y[n][h][w][co] = b[co] + sum_{a,b,c,ci} x[n][h+a][w+b][c][ci] W[a][b][c][ci][co]: thread (w, k-part): 1/16 of the 27 x 32
//                  products for all nine outputs, then a butterfly over the 16 parts (fixed order: bitwise reproducible)
//   backward-data  the full correlation of the one-deep 9-channel dY with the flipped filter (the engine passes W^T: [a'][b'][c'][co][ci]):
//                  dX[n][h][w][t][ci] = sum_{a',b',co} dY[n][h+a'-2][w+b'-2][co] W^T[a'][b'][2-t][co][ci]; thread (w, t, 4 ci); the per-sample
//                  amax of dX (the next H3 kernels' scale) comes out of the same pass
// ---------------------------------------------------------------------------------------------------
bool conv3d_up_forward_supported(const ConvGeom& g)
{
    return g.Cin == 32 && g.Cout == 9 && g.kh == 3 && g.kw == 3 && g.kt == 3 && g.ph == 0 && g.pw == 0 && g.pt == 0 && !g.reflect_hw && !g.reflect_t &&
           g.Ti == 3 && g.To == 1 && g.Ho == g.Hi - 2 && g.Wo == g.Wi - 2 && g.Wo == 16;
}
bool conv3d_up_bwd_data_supported(const ConvGeom& g)
{
    return g.Cin == 9 && g.Cout == 32 && g.kh == 3 && g.kw == 3 && g.kt == 3 && g.ph == 2 && g.pw == 2 && g.pt == 2 && !g.reflect_hw && !g.reflect_t &&
           g.Ti == 1 && g.To == 3 && g.Ho == g.Hi + 2 && g.Wo == g.Wi + 2 && g.Wi <= 30;
}
__global__ __launch_bounds__(256) void conv3_up_fwd_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y)
{
    extern __shared__ float up_lds[];
    float* sW = up_lds;                                   // [27 * 32][9]
    float* sX = sW + 27 * 32 * 9;                         // [3 rows][Wi][3][32]
    const int tid = threadIdx.x;
    const int rowf = g.Wi * 3 * 32;                        // floats of one input row
    // once per workgroup (the grid is ~one per CU).  The engine's filters are packed tightly behind one another (weff + w_off: 4-byte aligned
    // only -- upscaleConv1 sits behind residConv1's 81 floats), so the 16-byte copy is taken only when the pointer allows it
    if ((reinterpret_cast<unsigned long>(w) & 15) == 0) {
        for (int i = tid; i < 27 * 32 * 9 / 4; i += 256) reinterpret_cast<float4*>(sW)[i] = reinterpret_cast<const float4*>(w)[i];
    } else {
        for (int i = tid; i < 27 * 32 * 9; i += 256) sW[i] = w[i];
    }
    const int wo = tid >> 4, kp = tid & 15;                // 16 output columns x 16 parts of the 864 products: part kp = (a, b, c, ci) index = kp + 16 j
    for (int row = blockIdx.x; row < g.N * g.Ho; row += gridDim.x) {
    const int n = row / g.Ho, h = row - n * g.Ho;
    __syncthreads();                                       // (the previous row's readers are done)
    const float4* xr = reinterpret_cast<const float4*>(x + ((long)n * g.Hi + h) * rowf);
    for (int i = tid; i < 3 * rowf / 4; i += 256) reinterpret_cast<float4*>(sX)[i] = xr[i];
    __syncthreads();
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (int q = 0; q < 54; ++q) {
        const int k = kp + 16 * q;                         // (tap, ci) = (k >> 5, k & 31); tap = (a * 3 + b) * 3 + c
        const int tap = k >> 5, ci = k & 31;
        const int a = tap / 9, bc = tap - 9 * a, b = bc / 3, c = bc - 3 * b;
        const float xv = sX[a * rowf + ((wo + b) * 3 + c) * 32 + ci];
        const float* wr = sW + k * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        float v = acc[j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
        acc[j] = v;
    }
    if (kp < 9) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) o = kp == j ? acc[j] : o;
        o += bias ? bias[kp] : 0.f;
        if (g.relu) o = fmaxf(o, 0.f);
        y[(((long)n * g.Ho + h) * g.Wo + wo) * 9 + kp] = o;
    }
    }
}
__global__ __launch_bounds__(256) void conv3_up_bwd_data_kernel(ConvGeom g, const float* __restrict__ dy, const float* __restrict__ wT,
                                                               const float* __restrict__ bias, float* __restrict__ dx, unsigned* __restrict__ amax)
{
    extern __shared__ float up_lds[];
    float* sW = up_lds;                                   // [27][9][32]
    float* sD = sW + 27 * 9 * 32;                         // [3 rows][Wi + 4][9] with two zero columns on each side
    const int tid = threadIdx.x;
    const int Wp = g.Wi + 4;
    if ((reinterpret_cast<unsigned long>(wT) & 15) == 0) {        // (see conv3_up_fwd_kernel: weffT + w_off is 4-byte aligned only)
        for (int i = tid; i < 27 * 9 * 32 / 4; i += 256) reinterpret_cast<float4*>(sW)[i] = reinterpret_cast<const float4*>(wT)[i];
    } else {
        for (int i = tid; i < 27 * 9 * 32; i += 256) sW[i] = wT[i];
    }
    __shared__ float red[4];
    for (int row = blockIdx.x; row < g.N * g.Ho; row += gridDim.x) {
    const int n = row / g.Ho, h = row - n * g.Ho;
    __syncthreads();
    for (int i = tid; i < 3 * Wp * 9; i += 256) {
        const int a = i / (Wp * 9), r = i - a * Wp * 9, wp = r / 9, co = r - 9 * wp;
        const int ih = h + a - 2, iw = wp - 2;
        sD[i] = (ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi) ? dy[(((long)n * g.Hi + ih) * g.Wi + iw) * 9 + co] : 0.f;
    }
    __syncthreads();
    float omax = 0.f;
    const int items = g.Wo * 3 * 8;                        // (w, t, group of 4 input channels)
    for (int it = tid; it < items; it += 256) {
        const int cg = it & 7, wt = it >> 3, wo = wt / 3, t = wt - 3 * wo;
        float4 acc = bias ? make_float4(bias[4 * cg], bias[4 * cg + 1], bias[4 * cg + 2], bias[4 * cg + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);      // (the engine's backward-data has none; scalar reads: the pointer is 4-byte aligned only)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float* dp = sD + (a * Wp + wo + b) * 9;
                const float* wp = sW + (((a * 3 + b) * 3 + (2 - t)) * 9) * 32 + 4 * cg;
#pragma unroll
                for (int co = 0; co < 9; ++co) {
                    const float d = dp[co];
                    const float4 q = *reinterpret_cast<const float4*>(wp + co * 32);
                    acc.x = fmaf(d, q.x, acc.x); acc.y = fmaf(d, q.y, acc.y); acc.z = fmaf(d, q.z, acc.z); acc.w = fmaf(d, q.w, acc.w);
                }
            }
        *reinterpret_cast<float4*>(dx + ((((long)n * g.Ho + h) * g.Wo + wo) * 3 + t) * 32 + 4 * cg) = acc;
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = omax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            atomicMax(amax + n, __float_as_uint(m));
        }
    }
    }
}
int conv3d_up_forward(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t s)
{
    if (!conv3d_up_forward_supported(g)) { set_error("conv3d_up_forward: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    const size_t lds = ((size_t)27 * 32 * 9 + (size_t)3 * g.Wi * 3 * 32) * sizeof(float);
    static std::once_flag once;
    std::call_once(once, [] { note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_up_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); });
    hipLaunchKernelGGL(conv3_up_fwd_kernel, dim3((unsigned)(g.N * g.Ho < 512 ? g.N * g.Ho : 512)), dim3(256), lds, s, g, x, w, bias, y);
    return check_launch("conv3_up_fwd");
}
// amax: per-sample slots of the output (may be null; zeroed by the caller)
int conv3d_up_bwd_data(const ConvGeom& g, const float* dy, const float* wT, const float* bias, float* dx, unsigned* amax, hipStream_t s)
{
    if (!conv3d_up_bwd_data_supported(g)) { set_error("conv3d_up_bwd_data: unsupported geometry", hipSuccess); return PROBAV_EINVAL; }
    const size_t lds = ((size_t)27 * 9 * 32 + (size_t)3 * (g.Wi + 4) * 9) * sizeof(float);
    hipLaunchKernelGGL(conv3_up_bwd_data_kernel, dim3((unsigned)(g.N * g.Ho < 768 ? g.N * g.Ho : 768)), dim3(256), lds, s, g, dy, wT, bias, dx, amax);
    return check_launch("conv3_up_bwd_data");
}

// ---------------------------------------------------------------------------------------------------
// backward-filter.  256 threads = 32 output channels x 8 input-channel groups; every thread keeps
// TAPS x CI_PER partial sums in registers while the block walks its chunk of output voxels (all
// threads on the same voxel => bounds tests are scalar and the x loads are 32-lane broadcasts).
// Per-chunk partials go to scratch and are summed in a fixed order by slab_sum_batch_kernel (slab_sum_later), so the
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

// The slabs / chunk partials of the launches below are summed by slab_sum_later (probav_common.h; kernels_small.hip: slab_sum_batch_kernel): a filter gradient and its bias
// gradient are two jobs of one launch, fp64 across the slabs in a fixed order (the filter gradient is a sum with heavy cancellation), bitwise reproducible.
static int sum_partials(const float* p1, float* o1, long n1, const float* p2, float* o2, long n2, int chunks, hipStream_t s)
{
    SlabSumJob jobs[2] = {{p1, o1, n1, (int)n1, chunks}, {p2, o2, n2, (int)n2, chunks}};
    return slab_sum_later(s, jobs, o2 ? 2 : 1);
}


// ---------------------------------------------------------------------------------------------------
// backward-filter of the 2-D residual path (3x3x1 'valid' convolutions on [N, H, W, 1, C <= 9]: residConv1..3).  The generic
// kernel above walks a few voxels per block with dependent global loads (latency-bound: 0.17 ms for 37 MFLOP).  Here one
// workgroup takes half a patch: x rows and gated dy rows are staged in LDS once, thread (tap, ci, co) keeps ONE accumulator and
// walks the half-patch's output voxels with two LDS reads (both broadcasts inside a (tap, ci) / co group) and one fma each.
// One slab per workgroup, summed in fixed order by slab_sum_batch_kernel (slab_sum_later).
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
template <bool GATE>
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
        // four voxels of a stream per round, all eight requests out before the first is used: one request per round left the kernel waiting
        // for memory 68 times in a row (63 us for 142 MB); the clamped index of a slot beyond the row reads a valid voxel and counts as zero
        for (int v0 = 2 * wave + half; v0 < nvr; v0 += 64) {
            float d[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int vi = v0 + 16 * u < nvr ? v0 + 16 * u : v0;
                d[u] = dy[(ob + vi) * 32 + co];
                if constexpr (GATE) m[u] = gate[(ob + vi) * 32 + co]; else m[u] = 1.f;      // (a run-time test around a load makes hipcc branch and drain vmcnt per element)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = v0 + 16 * u < nvr;                   // (predicated, not skipped: a branch here sinks the requests above into it)
                const int vi = live ? v0 + 16 * u : v0;
                const int w = vi / g.To, t = vi - w * g.To;
                const float dd = (live && m[u] > 0.f) ? d[u] : 0.f;
                bsum += dd;
                const float* px = sx + w * Tp + t;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                        for (int dt = 0; dt < 3; ++dt) acc[(dh * 3 + dw) * 3 + dt] = fmaf(px[dh * rowf + dw * Tp + dt], dd, acc[(dh * 3 + dw) * 3 + dt]);
            }
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
        if (gate) hipLaunchKernelGGL(wgrad_cin1_kernel<true>, dim3((unsigned)slabs), dim3(512), lds, s, g, x, dy, gate, partial, pb);
        else hipLaunchKernelGGL(wgrad_cin1_kernel<false>, dim3((unsigned)slabs), dim3(512), lds, s, g, x, dy, gate, partial, pb);
        int rc = check_launch("wgrad_cin1");
        if (rc) return rc;
        return sum_partials(partial, dw, (long)27 * 32, pb, db, 32, slabs, s);
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
            const int cout = g.Cout;
            return sum_partials(partial, dw, nw, pb, db, cout, slabs, s);
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
    const int cout = g.Cout;
    return sum_partials(partial, dw, nw, partial_b, db, cout, chunks, s);
}

// ---------------------------------------------------------------------------------------------------
// The low-frequency residual path (models/modelsTF.py:45-53: residConv1 3x3 valid Cx -> 9 + ReLU, residConv2 and residConv3 3x3 valid 9 -> 9 on the
// temporal-mean image) as ONE launch each way.  0.46 MMAC per patch forward: as three launches of the generic direct kernel it was 21 + 35 + 40 us on the side
// stream, and its reverse pass three backward-filter launches, two backward-data launches and three slab sums (31 + 55 + 22 + 67 + 55 us of kernels that run
// beside the first residual blocks' reverse pass and stretch it: VERDICT r5 #4).  One workgroup per patch, the whole chain in LDS:
//   forward   mean image -> r1 -> r2 -> r3; a thread owns an output voxel and its nine channels (the filter values are wave-uniform: scalar loads);
//   backward  d r3 -> d r2 -> d r1 (full correlations with the same filters, d r1 gated by r1 > 0), then the three backward-filters and bias sums of the patch:
//             one output (tap, cin, cout) per thread, ONE slab per patch [dw3 | db3 | dw2 | db2 | dw1 | db1], summed over the patches by slab_sum_later (fp64, fixed order).
// ---------------------------------------------------------------------------------------------------
constexpr int RP_C = 9;                    // scale^2: the reference graph only closes for scale = 3 (probav_engine_create)
static size_t rp_fwd_lds(int Hin, int Cx) { return ((size_t)Hin * Hin * Cx + (size_t)(Hin - 2) * (Hin - 2) * RP_C + (size_t)(Hin - 4) * (Hin - 4) * RP_C) * sizeof(float); }
static size_t rp_bwd_lds(int Hin, int Cx)
{
    const size_t H1 = Hin - 2, H2 = Hin - 4, H3 = Hin - 6;
    return ((size_t)Hin * Hin * Cx + 2 * H1 * H1 * RP_C + 2 * H2 * H2 * RP_C + H3 * H3 * RP_C + 2 * 9 * RP_C * RP_C) * sizeof(float);
}
static long rp_slab_floats(int Cx) { return 2L * (9 * RP_C * RP_C + RP_C) + 9L * Cx * RP_C + RP_C; }
bool resid_path_supported(int Hin, int Cx, int C)
{
    return C == RP_C && Hin >= 7 && Cx >= 1 && Cx <= 4 && rp_bwd_lds(Hin, Cx) <= 150 * 1024;
}
size_t resid_path_slab_floats(int N, int Cx) { return (size_t)N * (size_t)rp_slab_floats(Cx); }

template <bool RELU, bool KEEP>
__device__ __forceinline__ void rp_conv(const float* src, int Hi, int Cin, const float* __restrict__ w, const float* __restrict__ bias,
                                        float* keep, float* __restrict__ out, int tid, int nthr)
{
    const int Ho = Hi - 2;
    for (int v = tid; v < Ho * Ho; v += nthr) {
        const int h = v / Ho, x = v - h * Ho;
        float acc[RP_C];
#pragma unroll
        for (int j = 0; j < RP_C; ++j) acc[j] = bias[j];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const float* xp = src + ((h + a) * Hi + (x + b)) * Cin;
                const float* wp = w + (a * 3 + b) * Cin * RP_C;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float xv = xp[ci];
#pragma unroll
                    for (int j = 0; j < RP_C; ++j) acc[j] = fmaf(xv, wp[ci * RP_C + j], acc[j]);
                }
            }
#pragma unroll
        for (int j = 0; j < RP_C; ++j) {
            const float o = RELU ? fmaxf(acc[j], 0.f) : acc[j];
            if (KEEP) keep[v * RP_C + j] = o;
            out[v * RP_C + j] = o;
        }
    }
}

__global__ __launch_bounds__(512) void resid_path_fwd_kernel(int Hin, int Cx, const float* __restrict__ mn,
                                                             const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const float* __restrict__ w3, const float* __restrict__ b3,
                                                             float* __restrict__ r1, float* __restrict__ r2, float* __restrict__ r3)
{
    extern __shared__ float rp_lds[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H1 = Hin - 2, H2 = Hin - 4, H3 = Hin - 6;
    float* in0 = rp_lds;
    float* a1 = in0 + Hin * Hin * Cx;
    float* a2 = a1 + H1 * H1 * RP_C;
    for (int i = tid; i < Hin * Hin * Cx; i += 512) in0[i] = mn[(long)n * Hin * Hin * Cx + i];
    __syncthreads();
    rp_conv<true, true>(in0, Hin, Cx, w1, b1, a1, r1 + (long)n * H1 * H1 * RP_C, tid, 512);
    __syncthreads();
    rp_conv<false, true>(a1, H1, RP_C, w2, b2, a2, r2 + (long)n * H2 * H2 * RP_C, tid, 512);
    __syncthreads();
    rp_conv<false, false>(a2, H2, RP_C, w3, b3, nullptr, r3 + (long)n * H3 * H3 * RP_C, tid, 512);
}

// d src[y][x][ci] = sum over taps (a, b) and co of d dst[y - a][x - b][co] w[a][b][ci][co]   (dst = the 'valid' convolution's output, Ho = Hi - 2)
template <bool GATED>
__device__ __forceinline__ void rp_bwd_data(const float* dd, int Ho, const float* wl, const float* act, float* ds, int tid, int nthr)
{
    const int Hi = Ho + 2;
    for (int o = tid; o < Hi * Hi * RP_C; o += nthr) {
        const int ci = o % RP_C, v = o / RP_C, y = v / Hi, x = v - y * Hi;
        float acc = 0.f;
        for (int a = 0; a < 3; ++a) {
            const int yy = y - a;
            if (yy < 0 || yy >= Ho) continue;
            for (int b = 0; b < 3; ++b) {
                const int xx = x - b;
                if (xx < 0 || xx >= Ho) continue;
                const float* dp = dd + (yy * Ho + xx) * RP_C;
                const float* wp = wl + ((a * 3 + b) * RP_C + ci) * RP_C;
#pragma unroll
                for (int co = 0; co < RP_C; ++co) acc = fmaf(dp[co], wp[co], acc);
            }
        }
        ds[o] = (!GATED || act[o] > 0.f) ? acc : 0.f;
    }
}

__global__ __launch_bounds__(1024) void resid_path_bwd_kernel(int Hin, int Cx, const float* __restrict__ mn, const float* __restrict__ r1, const float* __restrict__ r2,
                                                              const float* __restrict__ dtail, const float* __restrict__ w2, const float* __restrict__ w3,
                                                              float* __restrict__ slabs)
{
    extern __shared__ float rp_lds[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H1 = Hin - 2, H2 = Hin - 4, H3 = Hin - 6;
    float* in0 = rp_lds;                       // [Hin][Hin][Cx]   the temporal-mean image
    float* a1 = in0 + Hin * Hin * Cx;          // [H1][H1][9]      r1 (behind its ReLU)
    float* a2 = a1 + H1 * H1 * RP_C;           // [H2][H2][9]      r2
    float* d3 = a2 + H2 * H2 * RP_C;           // [H3][H3][9]      d loss / d r3
    float* d2 = d3 + H3 * H3 * RP_C;           // [H2][H2][9]      d loss / d r2
    float* d1 = d2 + H2 * H2 * RP_C;           // [H1][H1][9]      d loss / d (residConv1's pre-activation)
    float* W2 = d1 + H1 * H1 * RP_C;           // [3][3][9][9]
    float* W3 = W2 + 9 * RP_C * RP_C;
    for (int i = tid; i < Hin * Hin * Cx; i += 1024) in0[i] = mn[(long)n * Hin * Hin * Cx + i];
    for (int i = tid; i < H1 * H1 * RP_C; i += 1024) a1[i] = r1[(long)n * H1 * H1 * RP_C + i];
    for (int i = tid; i < H2 * H2 * RP_C; i += 1024) a2[i] = r2[(long)n * H2 * H2 * RP_C + i];
    for (int i = tid; i < H3 * H3 * RP_C; i += 1024) d3[i] = dtail[(long)n * H3 * H3 * RP_C + i];
    for (int i = tid; i < 9 * RP_C * RP_C; i += 1024) { W2[i] = w2[i]; W3[i] = w3[i]; }
    __syncthreads();
    rp_bwd_data<false>(d3, H3, W3, nullptr, d2, tid, 1024);
    __syncthreads();
    rp_bwd_data<true>(d2, H2, W2, a1, d1, tid, 1024);
    __syncthreads();
    // the patch's slab: [dw3 9*9*9 | db3 9 | dw2 9*9*9 | db2 9 | dw1 9*Cx*9 | db1 9]
    const int nw = 9 * RP_C * RP_C, nw1 = 9 * Cx * RP_C, total = 2 * (nw + RP_C) + nw1 + RP_C;
    float* sl = slabs + (long)n * total;
    for (int o = tid; o < total; o += 1024) {
        int k = o;
        const float *A, *D; int Wi, Cin;
        if (k < nw + RP_C) { A = a2; D = d3; Wi = H2; Cin = RP_C; }
        else if (k < 2 * (nw + RP_C)) { k -= nw + RP_C; A = a1; D = d2; Wi = H1; Cin = RP_C; }
        else { k -= 2 * (nw + RP_C); A = in0; D = d1; Wi = Hin; Cin = Cx; }
        const int Wo = Wi - 2, nf = 9 * Cin * RP_C;
        float acc = 0.f;
        if (k < nf) {
            const int co = k % RP_C, rest = k / RP_C, ci = rest % Cin, tap = rest / Cin, a = tap / 3, b = tap - 3 * a;
            const float* xp = A + (a * Wi + b) * Cin + ci;
            const float* dp = D + co;
            for (int h = 0; h < Wo; ++h)
                for (int x = 0; x < Wo; ++x) acc = fmaf(xp[(h * Wi + x) * Cin], dp[(h * Wo + x) * RP_C], acc);
        } else {
            const float* dp = D + (k - nf);
            for (int i = 0; i < Wo * Wo; ++i) acc += dp[i * RP_C];
        }
        sl[o] = acc;
    }
}

int resid_path_forward(int N, int Hin, int Cx, const float* mn, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                       float* r1, float* r2, float* r3, hipStream_t s)
{
    if (!resid_path_supported(Hin, Cx, RP_C) || N < 1) { set_error("resid_path_forward: unsupported shape", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] { note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(resid_path_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); });
    hipLaunchKernelGGL(resid_path_fwd_kernel, dim3((unsigned)N), dim3(512), rp_fwd_lds(Hin, Cx), s, Hin, Cx, mn, w1, b1, w2, b2, w3, b3, r1, r2, r3);
    return check_launch("resid_path_fwd");
}

int resid_path_backward(int N, int Hin, int Cx, const float* mn, const float* r1, const float* r2, const float* dtail, const float* w2, const float* w3,
                        float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, float* slabs, hipStream_t s)
{
    if (!resid_path_supported(Hin, Cx, RP_C) || N < 1) { set_error("resid_path_backward: unsupported shape", hipSuccess); return PROBAV_EINVAL; }
    static std::once_flag once;
    std::call_once(once, [] { note_attr_error(hipFuncSetAttribute(reinterpret_cast<const void*>(resid_path_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); });
    hipLaunchKernelGGL(resid_path_bwd_kernel, dim3((unsigned)N), dim3(1024), rp_bwd_lds(Hin, Cx), s, Hin, Cx, mn, r1, r2, dtail, w2, w3, slabs);
    const int rc = check_launch("resid_path_bwd");
    if (rc) return rc;
    const long nw = 9L * RP_C * RP_C, nw1 = 9L * Cx * RP_C, total = rp_slab_floats(Cx);
    const SlabSumJob jobs[6] = {{slabs, dw3, total, (int)nw, N}, {slabs + nw, db3, total, RP_C, N},
                                {slabs + nw + RP_C, dw2, total, (int)nw, N}, {slabs + 2 * nw + RP_C, db2, total, RP_C, N},
                                {slabs + 2 * (nw + RP_C), dw1, total, (int)nw1, N}, {slabs + 2 * (nw + RP_C) + nw1, db1, total, RP_C, N}};
    return slab_sum_later(s, jobs, 6);
}

}  // namespace probav
