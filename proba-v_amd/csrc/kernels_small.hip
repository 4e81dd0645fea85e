// Small fused kernels around the convolution stack (gfx950):
//   weight-norm reparameterisation fwd/bwd   (tensorflow_addons WeightNormalization; models/modelsTF.py:191-197)
//   head  : mean over T + instance normalisation        (models/modelsTF.py:23-27, 199-200)
//   tail  : depth_to_space x2 + Add + denormalise       (models/modelsTF.py:38-41, 52, 71-73, 202-203)
//   reflect-pad gradient fold                           (tf.pad 'reflect', models/modelsTF.py:157-158)
//   clip + round-half-even                              (test.py:117-119, models/testClass.py:27-28)
//   shift-compensated L1 / L2 / cPSNR fwd + bwd         (models/loss.py:37-84, 140-187, 226-238)
#include "probav_common.h"
#include <cstdlib>
#include <atomic>
#include <vector>
#include <functional>
#include <stdio.h>
#include <string.h>

namespace probav {

static thread_local char g_err[512] = "";
void set_error(const char* what, hipError_t e)
{
    if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    else snprintf(g_err, sizeof(g_err), "%s", what);
}
const char* last_error() { return g_err; }
static std::atomic<int> g_attr_err{(int)hipSuccess};                 // first failed hipFuncSetAttribute, sticky (process-wide: the attribute is)
void note_attr_error(hipError_t e)
{
    int ok = (int)hipSuccess;
    if (e != hipSuccess) g_attr_err.compare_exchange_strong(ok, (int)e);
}
int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(what, e); return PROBAV_EHIP; }
    const int a = g_attr_err.load();
    if (a != (int)hipSuccess) { set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for a kernel of this library", (hipError_t)a); return PROBAV_EHIP; }
    return PROBAV_OK;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Weight normalisation.  One 64-lane wave per (layer, output channel): the ||v||^2 reduction over
// taps*Cin is a strided read folded with a wavefront shuffle reduction; the same wave then writes the
// effective kernel in both layouts the convolutions consume:
//   weff  [tap][ci][co]                      forward / backward-filter
//   weffT [taps-1-tap][co][ci]               backward-data (flipped taps, channels swapped)
// ---------------------------------------------------------------------------------------------------
// the layer whose channel range [n_off, next n_off) holds `chan` (offsets ascending).  One wave, up to 64 layers: lane i looks at layer i, a ballot
// counts the layers that start at or before the channel -- one round trip instead of a chain of up to nl dependent loads (which was most of
// these kernels' 40 us).  More layers than lanes: the chain, from layer 63 on.
__device__ __forceinline__ int find_by_offset(const WnLayer* L, int nl, int chan, bool rows)
{
    const int lane = threadIdx.x & 63;
    const int mine = lane < nl ? (rows ? L[lane].r_off : L[lane].n_off) : 0x7fffffff;
    int i = __popcll(__ballot(mine <= chan)) - 1;                    // (layer 0 starts at 0: at least one)
    if (nl > 64) while (i + 1 < nl && chan >= (rows ? L[i + 1].r_off : L[i + 1].n_off)) ++i;
    return __builtin_amdgcn_readfirstlane(i);
}
__device__ __forceinline__ int find_layer(const WnLayer* L, int nl, int chan, int& co)
{
    const int i = find_by_offset(L, nl, chan, false);
    co = chan - L[i].n_off;
    return i;
}

// Optimizer coefficients of the fused update (the rule of nadam_kernel below; Keras Nadam / Adam / SGD by coefficients)
struct OptCoef { float lr, b1, b2, eps, c_g, c_m, c_v; };
__device__ __forceinline__ float opt_update(float theta, float g, float& m, float& v, const OptCoef& c)
{
    m = c.b1 * m + (1.f - c.b1) * g;
    v = c.b2 * v + (1.f - c.b2) * g * g;
    return theta - c.lr * (c.c_g * g + c.c_m * m) / (sqrtf(v * c.c_v) + c.eps);
}

// UPDATE: the optimizer step of this column's parameters (g[co], bias[co], v[:, co]) runs first, in the same wave, and the
// reparameterisation that follows is that of the UPDATED parameters: the next forward pass finds its effective weights ready
// (reference: optimizer.apply_gradients, then WeightNormalization's kernel recomputed at the next call --
// models/trainClass.py:132, models/modelsTF.py:191-197).  The arithmetic of every element is the one of nadam_kernel.
template <bool UPDATE>
__global__ __launch_bounds__(64) void wn_forward_kernel(const WnLayer* __restrict__ layers, int nl,
                                                       float* __restrict__ params, float* __restrict__ weff,
                                                       float* __restrict__ weffT, float* __restrict__ inv_norm,
                                                       unsigned* __restrict__ amax, const float* __restrict__ grad,
                                                       float* __restrict__ om, float* __restrict__ ov, OptCoef oc)
{
    int co;
    const int li = find_layer(layers, nl, blockIdx.x, co);
    const WnLayer L = layers[li];
    float* v = params + L.v_off;
    const int lane = threadIdx.x;
    float gain = params[L.g_off + co];
    if constexpr (UPDATE) {
        {   // the gain g[co] and the bias[co]: every lane computes both (the values are needed below), lanes 0 and 1 store them
            const int ig = L.g_off + co, ib = L.b_off + co;
            float mg = om[ig], vg = ov[ig], mb = om[ib], vb = ov[ib];
            gain = opt_update(gain, grad[ig], mg, vg, oc);
            const float bnew = opt_update(params[ib], grad[ib], mb, vb, oc);
            if (lane == 0) { params[ig] = gain; om[ig] = mg; ov[ig] = vg; }
            if (lane == 1) { params[ib] = bnew; om[ib] = mb; ov[ib] = vb; }
        }
        for (int k = lane; k < L.K; k += 64) {                      // the filter column; each lane re-reads only what it wrote itself
            const int i = L.v_off + k * L.Cout + co;
            float mi = om[i], vi = ov[i];
            params[i] = opt_update(params[i], grad[i], mi, vi, oc);
            om[i] = mi; ov[i] = vi;
        }
    }
    // The column lives in registers between the norm and the scaled stores: all of its loads are in flight together (a loop over a runtime K
    // waited for each in turn -- 14 round trips for a 3x3x3x32 column, twice).  Columns longer than WN_Q * 64 (none in this network) take the loop.
    constexpr int WN_Q = 14;
    float ss = 0.f, wmax = 0.f;
    if (L.K <= WN_Q * 64) {
        float q[WN_Q];
#pragma unroll
        for (int j = 0; j < WN_Q; ++j) { const int k = lane + 64 * j; q[j] = k < L.K ? v[(long)k * L.Cout + co] : 0.f; }
#pragma unroll
        for (int j = 0; j < WN_Q; ++j) ss = fmaf(q[j], q[j], ss);              // (the same order as the loop: zeros beyond the column add nothing)
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));            // tf.nn.l2_normalize epsilon
        const float scale = gain * inv;
        if (lane == 0) inv_norm[L.n_off + co] = inv;
#pragma unroll
        for (int j = 0; j < WN_Q; ++j) {
            const int k = lane + 64 * j;
            if (k < L.K) {
                const float w = q[j] * scale;
                weff[L.w_off + (long)k * L.Cout + co] = w;
                const int tap = k / L.Cin, ci = k - tap * L.Cin;
                weffT[L.w_off + ((long)(L.taps - 1 - tap) * L.Cout + co) * L.Cin + ci] = w;
                wmax = fmaxf(wmax, fabsf(w));
            }
        }
    } else {
        for (int k = lane; k < L.K; k += 64) { const float q = v[(long)k * L.Cout + co]; ss = fmaf(q, q, ss); }
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        const float scale = gain * inv;
        if (lane == 0) inv_norm[L.n_off + co] = inv;
        for (int k = lane; k < L.K; k += 64) {
            const float q = v[(long)k * L.Cout + co] * scale;
            weff[L.w_off + (long)k * L.Cout + co] = q;
            const int tap = k / L.Cin, ci = k - tap * L.Cin;
            weffT[L.w_off + ((long)(L.taps - 1 - tap) * L.Cout + co) * L.Cin + ci] = q;
            wmax = fmaxf(wmax, fabsf(q));
        }
    }
    if (amax) {     // largest |effective weight| and |bias| of the layer (slots li and nl + li) and of this output column: operand scales of the H3 kernels
#pragma unroll
        for (int o = 32; o; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
        // (the layer's own two slots, li and nl + li, are filled by wn_rowmax_kernel from the per-column slots and the biases: 2 x 3 920 atomicMax calls on
        // three cache lines queued for 24 us -- tools/atomic_probe.hip: 3 ns apiece, one after the other)
        if (lane == 0) amax[2 * nl + L.n_off + co] = __float_as_uint(wmax);
    }
}

// largest |effective weight| per INPUT channel (the output columns of the backward-data matrices weffT, and the rows of W1 the fused
// pointwise backward scales its dX by): one wave per (layer, input channel), after wn_forward_kernel on the same stream
__global__ __launch_bounds__(64) void wn_rowmax_kernel(const WnLayer* __restrict__ layers, int nl, const float* __restrict__ weff,
                                                      unsigned* __restrict__ arow, unsigned* __restrict__ amax, const float* __restrict__ params)
{
    const int i = find_by_offset(layers, nl, (int)blockIdx.x, true);
    const WnLayer L = layers[i];
    const int ci = blockIdx.x - L.r_off, lane = threadIdx.x;
    float m = 0.f;
    const int ne = L.taps * L.Cout;
    for (int e0 = 0; e0 < ne; e0 += 64 * 8) {                         // eight independent loads per round trip
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + 64 * j + lane;
            const int ec = e < ne ? e : 0;
            const int tap = ec / L.Cout, co = ec - tap * L.Cout;
            q[j] = weff[L.w_off + ((long)tap * L.Cin + ci) * L.Cout + co];
            q[j] = e < ne ? fabsf(q[j]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, q[j]);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) arow[blockIdx.x] = __float_as_uint(m);
    if (ci == 0) {      // the layer's first wave also fills the layer's own slots: largest |effective weight| (= the largest of its per-column slots, written by
                        // wn_forward_kernel before this launch) and largest |bias| (the parameters hold the updated biases by now)
        unsigned wm = 0u; float bm = 0.f;
        for (int c = lane; c < L.Cout; c += 64) { const unsigned q = amax[2 * nl + L.n_off + c]; wm = q > wm ? q : wm; bm = fmaxf(bm, fabsf(params[L.b_off + c])); }
#pragma unroll
        for (int o = 32; o; o >>= 1) { const unsigned q = (unsigned)__shfl_xor((int)wm, o, 64); wm = q > wm ? q : wm; bm = fmaxf(bm, __shfl_xor(bm, o, 64)); }
        if (lane == 0) { amax[i] = wm; amax[nl + i] = __float_as_uint(bm); }
    }
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ xall, size_t n, unsigned* __restrict__ slots)
{
    const float* x = xall + (size_t)blockIdx.y * n;                 // sample blockIdx.y: n values (n % 4 == 0 or the base stays 16-byte aligned only for sample 0: see the scalar path)
    unsigned* slot = slots + blockIdx.y;
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    if ((n & 3) == 0) {
        const size_t n4 = n >> 2;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {                  // four independent loads in flight
            const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(x)[i + stride];
            const float4 c = reinterpret_cast<const float4*>(x)[i + 2 * stride], d = reinterpret_cast<const float4*>(x)[i + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                                          // one atomic per workgroup, skipped when the slot already holds more
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(slot, __float_as_uint(m));
    }
}
int amax_tensor(const float* x, size_t per_sample, int N, unsigned* slots, hipStream_t s)
{
    if (N < 1) return PROBAV_OK;
    size_t blocks = (per_sample / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    const size_t cap = N >= 2048 ? 1 : 2048 / (size_t)N;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, s, x, per_sample, slots);
    return check_launch("amax");
}

// largest |w[r][c]| over the rows r of a row-major [rows][cols] matrix -> slots[c] (plain store; cols <= 256): the per-column filter
// scales of the H3 kernels for the single-operator entry points (the engine gets them from wn_forward)
__global__ __launch_bounds__(256) void amax_columns_kernel(const float* __restrict__ w, long rows, int cols, unsigned* __restrict__ slots)
{
    __shared__ float part[256];
    const int per = 256 / cols, c = threadIdx.x % cols, r0 = threadIdx.x / cols;      // `per` row streams per column
    float m = 0.f;
    if (r0 < per) for (long r = r0; r < rows; r += per) m = fmaxf(m, fabsf(w[r * cols + c]));
    part[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < cols) {
        for (int k = 1; k < per; ++k) m = fmaxf(m, part[k * cols + threadIdx.x]);
        slots[threadIdx.x] = __float_as_uint(m);
    }
}
int amax_columns(const float* w, long rows, int cols, unsigned* slots, hipStream_t s)
{
    if (cols < 1 || cols > 256) { set_error("amax_columns: 1 <= cols <= 256", hipSuccess); return PROBAV_EINVAL; }
    hipLaunchKernelGGL(amax_columns_kernel, dim3(1), dim3(256), 0, s, w, rows, cols, slots);
    return check_launch("amax_columns");
}

// d loss/d g = sum(dw * v) / ||v|| ;  d loss/d v = g/||v|| * (dw - v * sum(dw * v) / ||v||^2)
__global__ __launch_bounds__(64) void wn_backward_kernel(const WnLayer* __restrict__ layers, int nl,
                                                        const float* __restrict__ params, const float* __restrict__ dweff,
                                                        const float* __restrict__ inv_norm, float* __restrict__ grads)
{
    int co;
    const WnLayer L = layers[find_layer(layers, nl, blockIdx.x, co)];
    const float* v = params + L.v_off;
    const float* dw = dweff + L.w_off;
    const int lane = threadIdx.x;
    float dot = 0.f;
    for (int k = lane; k < L.K; k += 64) dot = fmaf(dw[(long)k * L.Cout + co], v[(long)k * L.Cout + co], dot);
    dot = wave_sum(dot);
    const float inv = inv_norm[L.n_off + co];
    const float gg = params[L.g_off + co];
    const bool clamped = inv >= 1e6f;                               // ||v||^2 < 1e-12: norm is the constant 1e-6
    const float proj = clamped ? 0.f : dot * inv * inv;
    if (lane == 0) grads[L.g_off + co] = dot * inv;
    for (int k = lane; k < L.K; k += 64) {
        const long i = (long)k * L.Cout + co;
        grads[L.v_off + i] = gg * inv * (dw[i] - v[i] * proj);
    }
}

int wn_forward(const WnLayer* d_layers, int nlayers, int cout_total, int cin_total, const float* params,
               float* weff, float* weffT, float* inv_norm, unsigned* amax, hipStream_t s)
{
    hipLaunchKernelGGL(wn_forward_kernel<false>, dim3(cout_total), dim3(64), 0, s, d_layers, nlayers, const_cast<float*>(params), weff, weffT, inv_norm, amax,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, OptCoef());
    int rc = check_launch("wn_forward");
    if (rc || !amax) return rc;
    hipLaunchKernelGGL(wn_rowmax_kernel, dim3(cin_total), dim3(64), 0, s, d_layers, nlayers, weff, amax + 2 * nlayers + cout_total, amax, params);
    return check_launch("wn_rowmax");
}
int optimizer_wn_step(const WnLayer* d_layers, int nlayers, int cout_total, int cin_total, float* params, const float* grad, float* m, float* v,
                      float lr, float b1, float b2, float eps, float c_g, float c_m, float c_v,
                      float* weff, float* weffT, float* inv_norm, unsigned* amax, hipStream_t s)
{
    OptCoef oc = {lr, b1, b2, eps, c_g, c_m, c_v};
    hipLaunchKernelGGL(wn_forward_kernel<true>, dim3(cout_total), dim3(64), 0, s, d_layers, nlayers, params, weff, weffT, inv_norm, amax, grad, m, v, oc);
    int rc = check_launch("optimizer_wn_step");
    if (rc || !amax) return rc;
    hipLaunchKernelGGL(wn_rowmax_kernel, dim3(cin_total), dim3(64), 0, s, d_layers, nlayers, weff, amax + 2 * nlayers + cout_total, amax, params);
    return check_launch("wn_rowmax");
}
int wn_backward(const WnLayer* d_layers, int nlayers, int cout_total, const float* params,
                const float* dweff, const float* inv_norm, float* grads, hipStream_t s)
{
    hipLaunchKernelGGL(wn_backward_kernel, dim3(cout_total), dim3(64), 0, s, d_layers, nlayers, params, dweff, inv_norm, grads);
    return check_launch("wn_backward");
}

// ---------------------------------------------------------------------------------------------------
// head: x [N,H,W,T,C] -> xn = (x - mean)/std  [N,H,W,T,C] ,  mn = (mean_T(x) - mean)/std  [N,H,W,C]     (C = 1, or 3 for isGrayScale=False)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                  float* __restrict__ mn, int nhw, int T, int C, float mean, float stdv, unsigned* __restrict__ zero, int nzero)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // (pixel, channel)
    for (int z = i; z < nzero; z += gridDim.x * 256) zero[z] = 0u;      // the forward pass's per-sample amax slots (a 6-us memset launch of their own before)
    if (i >= nhw * C) return;
    const int px = i / C, c = i - px * C;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
        const long o = ((long)px * T + t) * C + c;
        const float q = x[o];
        s += q;
        xn[o] = (q - mean) / stdv;
    }
    mn[i] = (s / (float)T - mean) / stdv;
}
int head_forward(const float* x, float* xn, float* mn, int nhw, int T, int C, float mean, float stdv, hipStream_t s, unsigned* zero, int nzero)
{
    hipLaunchKernelGGL(head_kernel, dim3((nhw * C + 255) / 256), dim3(256), 0, s, x, xn, mn, nhw, T, C, mean, stdv, zero, nzero);
    return check_launch("head");
}

// tail: y[n, s*h+i, s*w+j] = (up[n,h,w,i*s+j] + r3[n,h,w,i*s+j]) * std + mean
__global__ __launch_bounds__(256) void tail_fwd_kernel(const float* __restrict__ up, const float* __restrict__ r3,
                                                      float* __restrict__ y, int N, int P, int sc, float mean, float stdv)
{
    const int S = P * sc;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * S * S) return;
    const int X = (int)(i % S), Y = (int)((i / S) % S), n = (int)(i / ((long)S * S));
    const long src = (((long)n * P + Y / sc) * P + X / sc) * (sc * sc) + (Y % sc) * sc + (X % sc);
    y[i] = (up[src] + r3[src]) * stdv + mean;
}
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dtail,
                                                      int N, int P, int sc, float stdv, unsigned* __restrict__ zero, int nzero)
{
    const int S = P * sc, C = sc * sc;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (long z = i; z < nzero; z += (long)gridDim.x * 256) zero[z] = 0u;      // the backward pass's per-sample amax slots
    if (i >= (long)N * P * P * C) return;
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % P); r /= P;
    const int h = (int)(r % P);
    const int n = (int)(r / P);
    dtail[i] = dy[((long)n * S + h * sc + c / sc) * S + w * sc + c % sc] * stdv;
}
int tail_forward(const float* up, const float* r3, float* y, int N, int P, int sc, float mean, float stdv, hipStream_t s)
{
    const long n = (long)N * P * sc * P * sc;
    hipLaunchKernelGGL(tail_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, up, r3, y, N, P, sc, mean, stdv);
    return check_launch("tail_fwd");
}
int tail_backward(const float* dy, float* dtail, int N, int P, int sc, float stdv, hipStream_t s, unsigned* zero, int nzero)
{
    const long n = (long)N * P * P * sc * sc;
    hipLaunchKernelGGL(tail_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, dtail, N, P, sc, stdv, zero, nzero);
    return check_launch("tail_bwd");
}

// Gradient of tf.pad(x, 1 on H and W, 'reflect'): padded row 0 mirrors row 1, padded row H+1 mirrors row H-2.
template <int VEC>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx,
                                                          int N, int H, int W, int TC, unsigned* __restrict__ amax)
{
    // VEC = 4: TC % 4 == 0 and 16-byte aligned tensors; indices stay in 32 bits inside a sample (64-bit divisions cost more than the traffic)
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int n = blockIdx.y;                                       // one sample per grid row: its largest |dx| goes to amax[n]
    const int TCv = TC / VEC, per = H * W * TCv;
    const float* src = dpad + (long)n * (H + 2) * (W + 2) * TC;
    float* dst = dx + (long)n * H * W * TC;
    float m = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        const int r = i / TCv, e = i - r * TCv;
        const int h = r / W, w = r - h * W;
        int hs[2], ws[2], nh = 1, nw = 1;
        hs[0] = h + 1; ws[0] = w + 1;
        if (h == 1) hs[nh++] = 0;
        if (h == H - 2) hs[nh++] = H + 1;          // H >= 4, so h == 1 and h == H-2 never coincide
        if (w == 1) ws[nw++] = 0;
        if (w == W - 2) ws[nw++] = W + 1;
        vec_t s = 0.f;
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b)
                s += *reinterpret_cast<const vec_t*>(src + ((long)(hs[a] * (W + 2) + ws[b]) * TCv + e) * VEC);
        *reinterpret_cast<vec_t*>(dst + (long)i * VEC) = s;
#pragma unroll
        for (int c = 0; c < VEC; ++c) m = fmaxf(m, fabsf(s[c]));
    }
    if (amax) {                                 // largest |dx| of the sample (H3 operand scale of the next layer's kernels): one guarded atomic per workgroup
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float wm[4];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            atomicMax(amax + n, __float_as_uint(m));
        }
    }
}
// nothing but a dependent chain of 32x32x16 fp16 MFMAs: what the matrix pipe of THIS device sustains (probav_mfma_probe)
typedef _Float16 probe_f16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_probe_kernel(const probe_f16x8* __restrict__ seed, float* __restrict__ sink, int iters)
{
    probe_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const probe_f16x8 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    sink[blockIdx.x * 256 + threadIdx.x] = t;
}
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_probe16_kernel(const probe_f16x8* __restrict__ seed, float* __restrict__ sink, int iters)
{
    probe_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const probe_f16x8 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int mfma_probe(const void* seed, float* sink, int iters, int launches, hipStream_t s, int shape)
{
    for (int l = 0; l < launches; ++l) {
        if (shape == 1) hipLaunchKernelGGL(mfma_probe16_kernel, dim3(256), dim3(256), 0, s, (const probe_f16x8*)seed, sink, iters);
        else hipLaunchKernelGGL(mfma_probe_kernel, dim3(256), dim3(256), 0, s, (const probe_f16x8*)seed, sink, iters);
    }
    return check_launch("mfma_probe");
}

static thread_local ReduceSide* g_reduce_side = nullptr;
void reduce_side_activate(ReduceSide* ctx) { g_reduce_side = ctx; }
hipStream_t reduce_fork(hipStream_t s)
{
    ReduceSide* c = g_reduce_side;
    if (!c || !c->side || s == c->side) return s;
    hipEvent_t ev = c->ev[c->k++ & 7];
    if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(c->side, ev, 0) != hipSuccess) { (void)hipGetLastError(); c->last = nullptr; return s; }
    c->last = s;
    return c->side;
}
// The caller guarantees that nothing has been enqueued on s since its last reduce_fork(s): the side stream, being in order, already waits for
// that point -- no second event (an event record between two kernels of the launch stream costs it about 6 us of bubble).
hipStream_t reduce_fork_adjacent(hipStream_t s)
{
    ReduceSide* c = g_reduce_side;
    if (!c || !c->side || s == c->side) return s;
    if (c->k == 0 || c->last != s || c->defer) return reduce_fork(s);   // (deferred launches: the previous fork point is no longer adjacent)   // (nothing forked yet in this pass, or the last fork did not take: there is no earlier point)
    return c->side;
}
struct PendingList { std::vector<std::function<int(hipStream_t)>> fns; std::vector<SlabSumJob> jobs; };
int reduce_later(hipStream_t s, std::function<int(hipStream_t)> fn)
{
    ReduceSide* c = g_reduce_side;
    if (!c || !c->side || !c->defer || s == c->side) return fn(reduce_fork(s));
    if (!c->pending) c->pending = new PendingList();
    static_cast<PendingList*>(c->pending)->fns.push_back(std::move(fn));
    return PROBAV_OK;
}

// ---- batched slab sums: one launch for a list of (source slabs -> destination) jobs.  A block = 64 consecutive elements of one job x 4 interleaved slab groups (a wave's
// load is 256 consecutive bytes of one slab); a thread sums its group's slabs in two fp64 accumulators (16 independent requests in flight), the groups meet in LDS:
// ((g0 + g1) + g2) + g3.  The order is a function of (element, slabs) only: the same bits whatever else shares the launch. ----
constexpr int SLAB_SUM_MAXJ = 40;
struct SlabSumBatch { SlabSumJob job[SLAB_SUM_MAXJ]; int first[SLAB_SUM_MAXJ + 1]; int njobs; };
__global__ __launch_bounds__(256) void slab_sum_batch_kernel(SlabSumBatch b)
{
    __shared__ double red[4][64];
    int j = 0;
    while (j + 1 < b.njobs && (int)blockIdx.x >= b.first[j + 1]) ++j;
    const SlabSumJob& J = b.job[j];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - b.first[j]) * 64 + e;
    double a0 = 0.0, a1 = 0.0;
    if (i < J.count) {
        const float* p = J.src + i;
        int c = g;
#pragma unroll 8
        for (; c + 4 < J.slabs; c += 8) { a0 += (double)p[(long)c * J.stride]; a1 += (double)p[(long)(c + 4) * J.stride]; }
        if (c < J.slabs) a0 += (double)p[(long)c * J.stride];
    }
    red[g][e] = a0 + a1;
    __syncthreads();
    if (g == 0 && i < J.count) J.dst[i] = (float)(((red[0][e] + red[1][e]) + red[2][e]) + red[3][e]);
}
static int slab_sum_launch(const SlabSumJob* jobs, int njobs, hipStream_t s)
{
    for (int j0 = 0; j0 < njobs; j0 += SLAB_SUM_MAXJ) {
        SlabSumBatch b;
        b.njobs = njobs - j0 < SLAB_SUM_MAXJ ? njobs - j0 : SLAB_SUM_MAXJ;
        int blocks = 0;
        for (int j = 0; j < b.njobs; ++j) { b.job[j] = jobs[j0 + j]; b.first[j] = blocks; blocks += (jobs[j0 + j].count + 63) / 64; }
        b.first[b.njobs] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(slab_sum_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
        const int rc = check_launch("slab_sum_batch");
        if (rc) return rc;
    }
    return PROBAV_OK;
}
int slab_sum_later(hipStream_t s, const SlabSumJob* jobs, int njobs)
{
    ReduceSide* c = g_reduce_side;
    if (!c || !c->side || !c->defer || s == c->side) return slab_sum_launch(jobs, njobs, reduce_fork(s));
    if (!c->pending) c->pending = new PendingList();
    PendingList* q = static_cast<PendingList*>(c->pending);
    q->jobs.insert(q->jobs.end(), jobs, jobs + njobs);
    return PROBAV_OK;
}
int reduce_flush(hipStream_t s)
{
    ReduceSide* c = g_reduce_side;
    if (!c || !c->pending) return PROBAV_OK;
    PendingList* q = static_cast<PendingList*>(c->pending);
    if (q->fns.empty() && q->jobs.empty()) return PROBAV_OK;
    // The small launches fork to the side stream (one event for all of them).  The slab sums do NOT: a flush's worth of them is one kernel that streams 150 MB at the memory's
    // rate (40 us) and goes to the launch stream itself -- measured against the same kernel on the side stream: -0.8 % of the step (a fork is an event record between two
    // chip-filling kernels, and what the low-priority stream has not finished is waited for at the join); flushes that only carry slab sums record no event at all.
    int rc = PROBAV_OK;
    if (!q->fns.empty()) {
        hipStream_t rs = reduce_fork(s);
        for (auto& fn : q->fns) { rc = fn(rs); if (rc) break; }      // (a failed launch ends the flush: the launches behind it belong to a pass that is being abandoned)
    }
    if (!rc && !q->jobs.empty()) rc = slab_sum_launch(q->jobs.data(), (int)q->jobs.size(), s);
    q->fns.clear(); q->jobs.clear();
    return rc;
}
void reduce_free_pending(ReduceSide* ctx)
{
    if (ctx && ctx->pending) { delete static_cast<PendingList*>(ctx->pending); ctx->pending = nullptr; }
}
void reduce_drop_pending()
{
    ReduceSide* c = g_reduce_side;
    if (c && c->pending) { static_cast<PendingList*>(c->pending)->fns.clear(); static_cast<PendingList*>(c->pending)->jobs.clear(); }
}
int reduce_join(hipStream_t s)
{
    ReduceSide* c = g_reduce_side;
    { const int rc = reduce_flush(s); if (rc) return rc; }
    if (!c || !c->side || c->k == 0) return PROBAV_OK;
    if (hipEventRecord(c->joined, c->side) != hipSuccess || hipStreamWaitEvent(s, c->joined, 0) != hipSuccess) {
        set_error("reduce_join: event record / wait", hipGetLastError());
        return PROBAV_EHIP;
    }
    c->k = 0;
    c->last = nullptr;
    return PROBAV_OK;
}

// The same fold, one workgroup per output row (n, h): which padded rows fold onto it is a property of the workgroup (row h + 1, and row 0 / H + 1 behind
// rows 1 / H - 2), the row's 16-byte elements are walked in order (every request of a wave is one contiguous run), one multiply-high finds the column.
__global__ __launch_bounds__(256) void reflect_fold_rows_kernel(const float4* __restrict__ dpad, float4* __restrict__ dx, int H, int W, int TCv, unsigned mTCv,
                                                               unsigned* __restrict__ amax)
{
    const int h = blockIdx.x, n = blockIdx.y;
    const int rowv = (W + 2) * TCv;                                 // 16-byte elements of a padded row
    const float4* r0 = dpad + ((long)n * (H + 2) + h + 1) * rowv;
    const int extra = h == 1 ? -(h + 1) : (h == H - 2 ? 2 : 0);     // the second padded row that folds onto this one, relative to r0 (0: none)
    const float4* r1 = r0 + (long)extra * rowv;
    float4* dst = dx + ((long)n * H + h) * W * TCv;
    float m = 0.f;
    const int nel = W * TCv;
    for (int i0 = threadIdx.x; i0 < nel; i0 += 256 * 4) {           // four elements per thread and round: their requests go out together
        float4 a[4], b[4];
        int c1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k < nel ? i0 + 256 * k : i0;
            const int w = (int)__umulhi((unsigned)i, mTCv), e = i - w * TCv;
            const int c0 = (w + 1) * TCv + e;
            c1[k] = w == 1 ? e : (w == W - 2 ? (W + 1) * TCv + e : -1);          // the second padded column
            a[k] = r0[c0];
            b[k] = r1[c0];                                                          // (extra == 0: the same element again, unused)
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i >= nel) break;
            float4 v = a[k];
            if (extra) { v.x += b[k].x; v.y += b[k].y; v.z += b[k].z; v.w += b[k].w; }
            if (c1[k] >= 0) {
                float4 q = r0[c1[k]];
                if (extra) { const float4 c = r1[c1[k]]; q.x += c.x; q.y += c.y; q.z += c.z; q.w += c.w; }
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            dst[i] = v;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float wm[4];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            atomicMax(amax + n, __float_as_uint(m));                   // (no guarding read of the slot: x6_device.h, amax_commit)
        }
    }
}

int reflect_fold(const float* dpad, float* dx, int N, int H, int W, int TC, unsigned* amax, hipStream_t s)
{
    if (H < 4 || W < 4) { set_error("reflect_fold: H, W must be >= 4", hipSuccess); return PROBAV_EINVAL; }
    if (TC % 4 == 0 && ((reinterpret_cast<uintptr_t>(dpad) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 && H <= 65535 && N <= 65535) {
        const int TCv = TC / 4;
        const unsigned mTCv = (unsigned)((0x100000000ull + (unsigned)TCv - 1) / (unsigned)TCv);     // floor(i / TCv) = umulhi(i, m) for i < 2^16 * ... (i < W * TCv here)
        if ((long)W * TCv < (1l << 20) && TCv >= 2 && TCv < 4096) {
            hipLaunchKernelGGL(reflect_fold_rows_kernel, dim3((unsigned)H, (unsigned)N), dim3(256), 0, s, reinterpret_cast<const float4*>(dpad),
                               reinterpret_cast<float4*>(dx), H, W, TCv, mTCv, amax);
            return check_launch("reflect_fold");
        }
    }
    const long per = (long)H * W * TC;
    long blocks = (per + 255) / 256;
    const long cap = N >= 4096 ? 1 : 4096 / N;
    if (blocks > cap) blocks = cap;
    const bool v4 = TC % 4 == 0 && ((reinterpret_cast<uintptr_t>(dpad) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    if (v4) {
        blocks = (per / 4 + 255) / 256;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(reflect_fold_kernel<4>, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, s, dpad, dx, N, H, W, TC, amax);
    } else
        hipLaunchKernelGGL(reflect_fold_kernel<1>, dim3((unsigned)blocks, (unsigned)N), dim3(256), 0, s, dpad, dx, N, H, W, TC, amax);
    return check_launch("reflect_fold");
}

// General fold: dx[i] = sum of dpad over every padded index that mirrors onto i, per dimension (pads 0..2; needs extent > 2 * pad).
__device__ __forceinline__ int fold_srcs(int i, int n, int p, int (&u)[3])
{
    int c = 0;
    u[c++] = i + p;
    for (int j = 1; j <= p; ++j) {
        if (i == j) u[c++] = p - j;                    // padded index p - j mirrors onto j
        if (i == n - 1 - j) u[c++] = n - 1 + p + j;    // padded index n - 1 + p + j mirrors onto n - 1 - j
    }
    return c;
}
__global__ __launch_bounds__(256) void reflect_fold3_kernel(const float* __restrict__ dpad, float* __restrict__ dx, int N, int H, int W,
                                                           int T, int C, int ph, int pw, int pt)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * H * W * T * C) return;
    long r = i;
    const int c = (int)(r % C); r /= C;
    const int t = (int)(r % T); r /= T;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    int hs[3], ws[3], ts[3];
    const int nh = fold_srcs(h, H, ph, hs), nw = fold_srcs(w, W, pw, ws), nt = fold_srcs(t, T, pt, ts);
    const int Hp = H + 2 * ph, Wp = W + 2 * pw, Tp = T + 2 * pt;
    float s = 0.f;
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b)
            for (int d = 0; d < nt; ++d)
                s += dpad[((((long)n * Hp + hs[a]) * Wp + ws[b]) * Tp + ts[d]) * C + c];
    dx[i] = s;
}
int reflect_fold3(const float* dpad, float* dx, int N, int H, int W, int T, int C, int ph, int pw, int pt, hipStream_t s)
{
    if (ph < 0 || pw < 0 || pt < 0 || ph > 2 || pw > 2 || pt > 2 || H <= 2 * ph || W <= 2 * pw || T <= 2 * pt) {
        set_error("reflect_fold3: pads must be 0..2 and smaller than half the extent", hipSuccess); return PROBAV_EINVAL;
    }
    const long n = (long)N * H * W * T * C;
    hipLaunchKernelGGL(reflect_fold3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dpad, dx, N, H, W, T, C, ph, pw, pt);
    return check_launch("reflect_fold3");
}

// tf.clip_by_value(x, lo, hi) then tf.round (half to even == rintf in the default rounding mode)
__global__ __launch_bounds__(256) void clip_round_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, float lo, float hi)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rintf(fminf(fmaxf(in[i], lo), hi));
}
int clip_round(const float* in, float* out, size_t n, float lo, float hi, hipStream_t s)
{
    hipLaunchKernelGGL(clip_round_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, lo, hi);
    return check_launch("clip_round");
}

// ---------------------------------------------------------------------------------------------------
// Shift-compensated losses.  One 256-thread block per sample; each of its four waves owns a subset
// of the (2*border+1)^2 candidate registrations and evaluates them with wavefront shuffle reductions
// only (no block barrier per shift).  Sums run in fp64: the path is ~1.7 MFLOP per sample, the
// reference needs ~600 tiny TF ops for it, here it is one launch.
//   n = sum M ; b = sum(H - P*M)/n ; C = (P + b)*M ; l1 = sum|H - C|/n ; l2 = sum (H - C)^2/n
// (HR is NOT masked -- reference quirk, models/loss.py:146,151)
// ---------------------------------------------------------------------------------------------------
// One workgroup per (sample, shift row i): its four waves take the shifts (i, j), j = wave, wave + 4.  Candidates go to a scratch
// table [B][49][2] (fp64); shift_select_kernel picks the per-sample minima (first minimum in shift order wins ties), batch_mean_kernel
// the batch means.  (One workgroup per sample looping over all 49 shifts left half of the CUs idle: 131 us at batch 128.)
__global__ __launch_bounds__(256) void shift_loss_fwd_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred, int S, int border,
    double* __restrict__ cand)
{
    const int b = blockIdx.x, i = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = S - 2 * border, ns = 2 * border + 1, nshift = ns * ns;
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    for (int j = wave; j < ns; j += 4) {
        const int sft = i * ns + j;
        // Both passes walk the crop four pixels per lane and round, requested together (one pixel per round was one memory round trip per
        // pixel: 112 of them in a row per wave); the sums take the pixels in the same order as before, so the bits are the same.
        constexpr int U = 4;
        const int n = L * L;
        double cnt = 0.0, dsum = 0.0;
        for (int k0 = lane; k0 < n; k0 += 64 * U) {
            float m[U], h[U], q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 64 * u < n ? k0 + 64 * u : k0;
                const int r = k / L, c = k - r * L;
                m[u] = M[(i + r) * S + j + c] ? 1.f : 0.f;
                h[u] = H[(i + r) * S + j + c];
                q[u] = P[(border + r) * S + border + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + 64 * u < n) { cnt += (double)m[u]; dsum += (double)(h[u] - q[u] * m[u]); }
        }
        cnt = wave_sum(cnt); dsum = wave_sum(dsum);
        const double bias = dsum / cnt;
        double s1 = 0.0, s2 = 0.0;
        for (int k0 = lane; k0 < n; k0 += 64 * U) {
            float h[U], q[U]; bool mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 64 * u < n ? k0 + 64 * u : k0;
                const int r = k / L, c = k - r * L;
                mk[u] = M[(i + r) * S + j + c] != 0;
                h[u] = H[(i + r) * S + j + c];
                q[u] = P[(border + r) * S + border + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + 64 * u < n) {
                    const double m = mk[u] ? 1.0 : 0.0;
                    const double e = (double)h[u] - ((double)q[u] + bias) * m;
                    s1 += fabs(e);
                    s2 += e * e;
                }
        }
        s1 = wave_sum(s1) / cnt; s2 = wave_sum(s2) / cnt;
        if (lane == 0) { cand[((long)b * nshift + sft) * 2] = s1; cand[((long)b * nshift + sft) * 2 + 1] = s2; }
    }
}

__global__ __launch_bounds__(64) void batch_mean_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                       float* __restrict__ ma, float* __restrict__ mb, int n)
{
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) { s1 += (double)a[i]; s2 += (double)b2[i]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) { *ma = (float)(s1 / n); *mb = (float)(s2 / n); }
}

// per-sample minima over the shifts (ascending shift order, strict <: the first minimum wins ties); one thread per sample
// One workgroup of up to 1024 threads: the batch means (batch_mean_kernel's sums, in its order: a wave's lanes take the samples 64 apart, then the butterfly)
// are taken in the same launch -- a 6-us launch less between the forward and the backward pass.
__global__ __launch_bounds__(1024) void shift_select_mean_kernel(const double* __restrict__ cand, int B, int nshift, float max_val,
                                                                float* __restrict__ l1_out, float* __restrict__ l2_out, float* __restrict__ cpsnr_out,
                                                                int* __restrict__ arg_l1, int* __restrict__ arg_l2, float* __restrict__ ma, float* __restrict__ mb)
{
    __shared__ float sl1[1024], sl2[1024];
    const int b = threadIdx.x;
    if (b < B) {
        const double2* c = reinterpret_cast<const double2*>(cand) + (long)b * nshift;
        double best1 = 1e300, best2 = 1e300;
        int a1 = 0, a2 = 0;
#pragma unroll 7
        for (int sft = 0; sft < nshift; ++sft) {
            const double2 v = c[sft];
            if (v.x < best1) { best1 = v.x; a1 = sft; }
            if (v.y < best2) { best2 = v.y; a2 = sft; }
        }
        sl1[b] = l1_out[b] = (float)best1;
        sl2[b] = l2_out[b] = (float)best2;
        cpsnr_out[b] = (float)(10.0 * log10((double)max_val * (double)max_val / best2));
        arg_l1[b] = a1;
        arg_l2[b] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = threadIdx.x; i < B; i += 64) { s1 += (double)sl1[i]; s2 += (double)sl2[i]; }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (threadIdx.x == 0) { *ma = (float)(s1 / B); *mb = (float)(s2 / B); }
    }
}

__global__ __launch_bounds__(64) void shift_select_kernel(const double* __restrict__ cand, int B, int nshift, float max_val,
                                                         float* __restrict__ l1_out, float* __restrict__ l2_out, float* __restrict__ cpsnr_out,
                                                         int* __restrict__ arg_l1, int* __restrict__ arg_l2)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double2* c = reinterpret_cast<const double2*>(cand) + (long)b * nshift;
    double best1 = 1e300, best2 = 1e300;
    int a1 = 0, a2 = 0;
#pragma unroll 7
    for (int sft = 0; sft < nshift; ++sft) {
        const double2 v = c[sft];
        if (v.x < best1) { best1 = v.x; a1 = sft; }
        if (v.y < best2) { best2 = v.y; a2 = sft; }
    }
    l1_out[b] = (float)best1;
    l2_out[b] = (float)best2;
    cpsnr_out[b] = (float)(10.0 * log10((double)max_val * (double)max_val / best2));
    arg_l1[b] = a1;
    arg_l2[b] = a2;
}

int shift_loss_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border,
                       float* l1, float* l2, float* cpsnr, int* arg_l1, int* arg_l2, float* mean_l1, float* mean_l2,
                       float max_val, hipStream_t s)
{
    if (B <= 0 || S <= 2 * border) { set_error("shift_loss_forward: bad shape", hipSuccess); return PROBAV_EINVAL; }
    const int ns = 2 * border + 1;
    // candidate table owned by the library, grown on demand (never inside a captured region: the first call of a shape allocates)
    static double* cand = nullptr;
    static size_t cand_n = 0;
    const size_t need = (size_t)B * ns * ns * 2;
    if (need > cand_n) {
        if (cand) (void)hipFree(cand);
        cand = nullptr; cand_n = 0;
        hipError_t err = hipMalloc((void**)&cand, need * sizeof(double));
        if (err != hipSuccess) { set_error("shift_loss_forward: scratch allocation", err); return PROBAV_EHIP; }
        cand_n = need;
    }
    hipLaunchKernelGGL(shift_loss_fwd_kernel, dim3(B, ns), dim3(256), 0, s, hr, mask, pred, S, border, cand);
    if (B <= 1024) {
        hipLaunchKernelGGL(shift_select_mean_kernel, dim3(1), dim3((unsigned)((B + 63) / 64 * 64)), 0, s, cand, B, ns * ns, max_val, l1, l2, cpsnr, arg_l1, arg_l2, mean_l1, mean_l2);
        return check_launch("shift_loss_forward");
    }
    hipLaunchKernelGGL(shift_select_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, cand, B, ns * ns, max_val, l1, l2, cpsnr, arg_l1, arg_l2);
    hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(64), 0, s, l1, l2, mean_l1, mean_l2, B);
    return check_launch("shift_loss_forward");
}

// Gradient of mean_B min_shift loss w.r.t. pred for the arg-min shift:
//   L1: dP_k = -(M_k/n) (s_k - sum(s M)/n),  s = sign(H - C)      L2: s = 2 (H - C)
__global__ __launch_bounds__(256) void shift_loss_bwd_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred,
    const int* __restrict__ arg, int S, int border, int which, const float* __restrict__ upstream, float inv_b,
    float* __restrict__ dpred)
{
    const float scale = (upstream ? upstream[0] : 1.f) * inv_b;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = S - 2 * border, ns = 2 * border + 1;
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    float* G = dpred + (long)b * S * S;
    const int sft = arg[b], i = sft / ns, j = sft - i * ns;
    __shared__ double red[2][4];
    double cnt = 0.0, dsum = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const float m = M[(i + r) * S + j + c] ? 1.f : 0.f;
        cnt += (double)m;
        dsum += (double)(H[(i + r) * S + j + c] - P[(border + r) * S + border + c] * m);
    }
    cnt = wave_sum(cnt); dsum = wave_sum(dsum);
    if (lane == 0) { red[0][wave] = cnt; red[1][wave] = dsum; }
    __syncthreads();
    cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    dsum = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double bias = dsum / cnt;
    __syncthreads();
    double ssum = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
        const double e = (double)H[(i + r) * S + j + c] - ((double)P[(border + r) * S + border + c] + bias) * m;
        const double sg = which == 1 ? (e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0)) : 2.0 * e;
        ssum += sg * m;
    }
    ssum = wave_sum(ssum);
    if (lane == 0) red[0][wave] = ssum;
    __syncthreads();
    ssum = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    for (int k = tid; k < S * S; k += 256) {
        const int Y = k / S, X = k - Y * S, r = Y - border, c = X - border;
        float gk = 0.f;
        if (r >= 0 && r < L && c >= 0 && c < L) {
            const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
            const double e = (double)H[(i + r) * S + j + c] - ((double)P[Y * S + X] + bias) * m;
            const double sg = which == 1 ? (e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0)) : 2.0 * e;
            gk = (float)(-(m / cnt) * (sg - ssum / cnt) * (double)scale);
        }
        G[k] = gk;
    }
}
int shift_loss_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, int B, int S,
                        int border, int which, const float* upstream, float* dpred, hipStream_t s)
{
    if (which != 1 && which != 2) { set_error("shift_loss_backward: which must be 1 (L1) or 2 (L2)", hipSuccess); return PROBAV_EINVAL; }
    hipLaunchKernelGGL(shift_loss_bwd_kernel, dim3(B), dim3(256), 0, s, hr, mask, pred, arg, S, border, which, upstream, 1.0f / (float)B, dpred);
    return check_launch("shift_loss_backward");
}

// ---------------------------------------------------------------------------------------------------
// cfg loss = sobel_l1_mix: shiftCompensatedL1EdgeLoss (models/loss.py:86-97, 126-137, 214-219).  Per sample and candidate shift
//   D = H - (P + b) M   (b, M as above; H un-masked),  l1 = sum |D| / n,
//   sob = sum (|Gy| + |Gx|) / n,  (Gy, Gx) = tf.image.sobel_edges(H) - sobel_edges(C) = sobel_edges(D): 3x3 cross-correlations
//         [[-1,-2,-1],[0,0,0],[1,2,1]] and its transpose on the REFLECT-padded crop,
//   loss = pi * l1 + (1 - pi) * sob;  minimum over the shifts, mean over the batch.
// One 256-thread block per sample walks the shifts; D lives in LDS (the Sobel taps read it with mirrored indices).  fp64 sums.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ __forceinline__ double block_sum(double v, double* red, int tid)
{
    v = wave_sum(v);
    __syncthreads();                       // red may still be read by the previous reduction
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void shift_l1edge_fwd_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred, int S, int border, float pi,
    float* __restrict__ loss_out, int* __restrict__ arg_out)
{
    extern __shared__ float sD[];                                   // [L][L]
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = S - 2 * border, ns = 2 * border + 1, nshift = ns * ns;
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    double best = 1e300;
    int abest = 0;
    for (int sft = 0; sft < nshift; ++sft) {
        const int i = sft / ns, j = sft - i * ns;
        double cnt = 0.0, dsum = 0.0;
        for (int k = tid; k < L * L; k += 256) {
            const int r = k / L, c = k - r * L;
            const float m = M[(i + r) * S + j + c] ? 1.f : 0.f;
            cnt += (double)m;
            dsum += (double)(H[(i + r) * S + j + c] - P[(border + r) * S + border + c] * m);
        }
        cnt = block_sum(cnt, red, tid);
        dsum = block_sum(dsum, red, tid);
        const double bias = dsum / cnt;
        double s1 = 0.0;
        for (int k = tid; k < L * L; k += 256) {
            const int r = k / L, c = k - r * L;
            const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
            const double e = (double)H[(i + r) * S + j + c] - ((double)P[(border + r) * S + border + c] + bias) * m;
            sD[k] = (float)e;
            s1 += fabs(e);
        }
        s1 = block_sum(s1, red, tid);                               // (its barriers also publish sD)
        double sg = 0.0;
        for (int k = tid; k < L * L; k += 256) {
            const int r = k / L, c = k - r * L;
            const int r0 = mirror(r - 1, L) * L, r1 = r * L, r2 = mirror(r + 1, L) * L;
            const int c0 = mirror(c - 1, L), c2 = mirror(c + 1, L);
            const double gy = ((double)sD[r2 + c0] + 2.0 * sD[r2 + c] + sD[r2 + c2]) - ((double)sD[r0 + c0] + 2.0 * sD[r0 + c] + sD[r0 + c2]);
            const double gx = ((double)sD[r0 + c2] + 2.0 * sD[r1 + c2] + sD[r2 + c2]) - ((double)sD[r0 + c0] + 2.0 * sD[r1 + c0] + sD[r2 + c0]);
            sg += fabs(gy) + fabs(gx);
        }
        sg = block_sum(sg, red, tid);
        const double loss = ((double)pi * s1 + (1.0 - (double)pi) * sg) / cnt;
        if (loss < best) { best = loss; abest = sft; }              // first minimum in shift order wins ties
    }
    if (tid == 0) { loss_out[b] = (float)best; arg_out[b] = abest; }
}

// gradient of mean_B min_shift (pi l1 + (1-pi) sob) w.r.t. pred at the arg-min shift:
//   G = dloss/dD = [pi sign(D) + (1-pi) Sobel^T(sign(Gy), sign(Gx))] / n      (Sobel^T folds the mirrored pad back onto the crop)
//   dP_k = -M_k (G_k - sum(G M) / n)                                          (the second term is the brightness bias)
__global__ __launch_bounds__(256) void shift_l1edge_bwd_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred, const int* __restrict__ arg,
    int S, int border, float pi, const float* __restrict__ upstream, float inv_b, float* __restrict__ dpred)
{
    extern __shared__ float sm[];
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = S - 2 * border, ns = 2 * border + 1, Lp = L + 2;
    float* sD = sm;                     // [L][L]
    float* sY = sD + L * L;             // sign(Gy) [L][L]
    float* sX = sY + L * L;             // sign(Gx)
    float* sG = sX + L * L;             // gradient w.r.t. the padded crop [Lp][Lp], then folded
    const float scale = (upstream ? upstream[0] : 1.f) * inv_b;
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    float* G = dpred + (long)b * S * S;
    const int sft = arg[b], i = sft / ns, j = sft - i * ns;
    double cnt = 0.0, dsum = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const float m = M[(i + r) * S + j + c] ? 1.f : 0.f;
        cnt += (double)m;
        dsum += (double)(H[(i + r) * S + j + c] - P[(border + r) * S + border + c] * m);
    }
    cnt = block_sum(cnt, red, tid);
    dsum = block_sum(dsum, red, tid);
    const double bias = dsum / cnt;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
        sD[k] = (float)((double)H[(i + r) * S + j + c] - ((double)P[(border + r) * S + border + c] + bias) * m);
    }
    __syncthreads();
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const int r0 = mirror(r - 1, L) * L, r1 = r * L, r2 = mirror(r + 1, L) * L;
        const int c0 = mirror(c - 1, L), c2 = mirror(c + 1, L);
        const double gy = ((double)sD[r2 + c0] + 2.0 * sD[r2 + c] + sD[r2 + c2]) - ((double)sD[r0 + c0] + 2.0 * sD[r0 + c] + sD[r0 + c2]);
        const double gx = ((double)sD[r0 + c2] + 2.0 * sD[r1 + c2] + sD[r2 + c2]) - ((double)sD[r0 + c0] + 2.0 * sD[r1 + c0] + sD[r2 + c0]);
        sY[k] = gy > 0.0 ? 1.f : (gy < 0.0 ? -1.f : 0.f);
        sX[k] = gx > 0.0 ? 1.f : (gx < 0.0 ? -1.f : 0.f);
    }
    __syncthreads();
    // adjoint of the two correlations on the PADDED crop: position (u, v) in [-1, L] x [-1, L] collects the outputs q it feeds
    for (int k = tid; k < Lp * Lp; k += 256) {
        const int u = k / Lp - 1, v = k - (k / Lp) * Lp - 1;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
                const int qr = u - (a - 1), qc = v - (bb - 1);      // output whose tap (a, bb) reads (u, v)
                if (qr < 0 || qr >= L || qc < 0 || qc >= L) continue;
                const float ky = (a == 0 ? -1.f : (a == 2 ? 1.f : 0.f)) * (bb == 1 ? 2.f : 1.f);
                const float kx = (bb == 0 ? -1.f : (bb == 2 ? 1.f : 0.f)) * (a == 1 ? 2.f : 1.f);
                acc += ky * sY[qr * L + qc] + kx * sX[qr * L + qc];
            }
        sG[k] = acc;
    }
    __syncthreads();
    // fold the mirrored pad back (pad row -1 mirrors row 1, row L mirrors row L-2; same for columns) and mix with the L1 term
    double gm = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        float acc = 0.f;
        const int us[3] = {r, r == 1 ? -1 : -2, r == L - 2 ? L : -2};       // padded rows that mirror onto r (-2 = none)
        const int vs[3] = {c, c == 1 ? -1 : -2, c == L - 2 ? L : -2};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
                if (us[a] != -2 && vs[bb] != -2) acc += sG[(us[a] + 1) * Lp + vs[bb] + 1];
        const float d = sD[k];
        const float g = pi * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + (1.f - pi) * acc;
        sY[k] = g;                                                   // sY is dead: reuse for G * n
        gm += (double)g * (M[(i + r) * S + j + c] ? 1.0 : 0.0);
    }
    gm = block_sum(gm, red, tid);
    for (int k = tid; k < S * S; k += 256) {
        const int Y = k / S, X = k - Y * S, r = Y - border, c = X - border;
        float gk = 0.f;
        if (r >= 0 && r < L && c >= 0 && c < L) {
            const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
            gk = (float)(-(m / cnt) * ((double)sY[r * L + c] - gm / cnt) * (double)scale);
        }
        G[k] = gk;
    }
}

int shift_l1edge_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border, float pi,
                         float* loss, int* arg, float* mean, float* scratch_mean2, hipStream_t s)
{
    if (B <= 0 || S <= 2 * border + 2) { set_error("shift_l1edge_forward: bad shape", hipSuccess); return PROBAV_EINVAL; }
    const int L = S - 2 * border;
    hipLaunchKernelGGL(shift_l1edge_fwd_kernel, dim3(B), dim3(256), (size_t)L * L * sizeof(float), s, hr, mask, pred, S, border, pi, loss, arg);
    hipLaunchKernelGGL(batch_mean_kernel, dim3(1), dim3(64), 0, s, loss, loss, mean, scratch_mean2, B);
    return check_launch("shift_l1edge_forward");
}
int shift_l1edge_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, int B, int S, int border, float pi,
                          const float* upstream, float* dpred, hipStream_t s)
{
    const int L = S - 2 * border;
    const size_t lds = ((size_t)3 * L * L + (size_t)(L + 2) * (L + 2)) * sizeof(float);
    if (B <= 0 || L < 3 || lds > 64 * 1024) { set_error("shift_l1edge_backward: bad shape", hipSuccess); return PROBAV_EINVAL; }
    hipLaunchKernelGGL(shift_l1edge_bwd_kernel, dim3(B), dim3(256), lds, s, hr, mask, pred, arg, S, border, pi, upstream, 1.0f / (float)B, dpred);
    return check_launch("shift_l1edge_backward");
}

// ---------------------------------------------------------------------------------------------------
// cfg loss = l1msssim: shiftCompensatedRevSSIM (models/loss.py:99-124, 189-212), restated with the reference's quirks:
//   weights of scale s: w1d = exp(-x / (2 sigma_s^2)), x = linspace(-L/2, L/2, L)  (NOT squared), w = outer(w1d, w1d) * M, normalised
//   per sample;  mu, variance "sigma" and cov are w-weighted moments of H and C = (P + b) M;  luminance / contrast / structure use
//   C1, C1, C3;  pcs = prod_s contrast_s structure_s;  ssim term = 1 - sum_{s,b} luminance_{s,b} pcs_b / B;
//   mixed with the w-weighted L1: loss = eta * ssim + (1 - eta) * sum_{s,b,px} w |H - C| / (B * numBytes).
// The loss is ONE scalar per shift for the whole batch, and the minimum over the shifts is taken of that scalar (the batch shares
// the shift).  Kernel 1 (grid B x shifts) leaves 7 weighted sums per (shift, sample, scale); kernel 2 combines them and picks the
// arg-min shift; the backward kernel differentiates that shift.  All sums in fp64.
// ---------------------------------------------------------------------------------------------------
#define RS_NS 5
#define RS_NM 7          // Sw, SwH, SwC, SwHH, SwCC, SwHC, Sw|H-C|

__device__ __forceinline__ double rs_w1d(int k, int L, int s)
{
    const double sig[RS_NS] = {0.5, 1.0, 2.0, 4.0, 8.0};
    const double x = -0.5 * L + (double)k * (double)L / (double)(L - 1);      // tf.linspace(-L/2, L/2, L)
    return exp(-x / (2.0 * sig[s] * sig[s]));
}

__global__ __launch_bounds__(256) void revssim_moments_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred, int S, int border,
    double* __restrict__ mom /* [shift][B][5][7] */, int B)
{
    extern __shared__ double sW[];                                  // [5][L]
    __shared__ double red[4];
    const int b = blockIdx.x, sft = blockIdx.y, tid = threadIdx.x;
    const int L = S - 2 * border, ns = 2 * border + 1;
    const int i = sft / ns, j = sft - i * ns;
    for (int k = tid; k < RS_NS * L; k += 256) sW[k] = rs_w1d(k % L, L, k / L);
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    double cnt = 0.0, dsum = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const float m = M[(i + r) * S + j + c] ? 1.f : 0.f;
        cnt += (double)m;
        dsum += (double)(H[(i + r) * S + j + c] - P[(border + r) * S + border + c] * m);
    }
    cnt = block_sum(cnt, red, tid);
    dsum = block_sum(dsum, red, tid);
    const double bias = dsum / cnt;
    double acc[RS_NS][RS_NM];
#pragma unroll
    for (int s = 0; s < RS_NS; ++s)
#pragma unroll
        for (int q = 0; q < RS_NM; ++q) acc[s][q] = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
        const double h = (double)H[(i + r) * S + j + c];
        const double cc = ((double)P[(border + r) * S + border + c] + bias) * m;
#pragma unroll
        for (int s = 0; s < RS_NS; ++s) {
            const double w = sW[s * L + r] * sW[s * L + c] * m;
            acc[s][0] += w; acc[s][1] += w * h; acc[s][2] += w * cc; acc[s][3] += w * h * h; acc[s][4] += w * cc * cc;
            acc[s][5] += w * h * cc; acc[s][6] += w * fabs(h - cc);
        }
    }
    double* out = mom + ((long)sft * B + b) * RS_NS * RS_NM;
#pragma unroll
    for (int s = 0; s < RS_NS; ++s)
#pragma unroll
        for (int q = 0; q < RS_NM; ++q) {
            const double v = block_sum(acc[s][q], red, tid);
            if (tid == 0) out[s * RS_NM + q] = v;
        }
}

struct RsTerms { double lum[RS_NS], con[RS_NS], str[RS_NS], muH[RS_NS], muS[RS_NS], sH[RS_NS], sS[RS_NS], cov[RS_NS], l1[RS_NS]; };

__device__ __forceinline__ void rs_terms(const double* m, double C1, double C3, RsTerms& t)
{
#pragma unroll
    for (int s = 0; s < RS_NS; ++s) {
        const double* q = m + s * RS_NM;
        const double iw = 1.0 / q[0];
        t.muH[s] = q[1] * iw; t.muS[s] = q[2] * iw;
        t.sH[s] = q[3] * iw - t.muH[s] * t.muH[s];
        t.sS[s] = q[4] * iw - t.muS[s] * t.muS[s];
        t.cov[s] = q[5] * iw - t.muS[s] * t.muH[s];
        t.l1[s] = q[6] * iw;
        t.lum[s] = (2.0 * t.muH[s] * t.muS[s] + C1) / (t.muH[s] * t.muH[s] + t.muS[s] * t.muS[s] + C1);
        t.con[s] = (2.0 * t.sH[s] * t.sS[s] + C1) / (t.sH[s] * t.sH[s] + t.sS[s] * t.sS[s] + C1);
        t.str[s] = (2.0 * t.cov[s] + C3) / (t.sH[s] * t.sS[s] + C3);
    }
}

__global__ __launch_bounds__(64) void revssim_select_kernel(const double* __restrict__ mom, int B, int nshift, float max_val, float eta,
                                                            float* __restrict__ loss_out, int* __restrict__ arg_out)
{
    const double C1 = (0.01 * max_val) * (0.01 * max_val), C3 = 0.5 * (0.03 * max_val) * (0.03 * max_val);
    __shared__ double sl[64];
    const int lane = threadIdx.x;
    double mine = 1e300;
    int marg = 0;
    for (int sft = lane; sft < nshift; sft += 64) {                   // one lane per shift: a fixed-order sum over the batch
        double ssim = 0.0, l1 = 0.0;
        for (int b = 0; b < B; ++b) {
            RsTerms t;
            rs_terms(mom + ((long)sft * B + b) * RS_NS * RS_NM, C1, C3, t);
            double pcs = 1.0, lsum = 0.0;
#pragma unroll
            for (int s = 0; s < RS_NS; ++s) { pcs *= t.con[s] * t.str[s]; lsum += t.lum[s]; l1 += t.l1[s]; }
            ssim += lsum * pcs;
        }
        const double loss = (double)eta * (1.0 - ssim / B) + (1.0 - (double)eta) * (l1 / B) / (double)max_val;
        if (loss < mine) { mine = loss; marg = sft; }
    }
    sl[lane] = mine;
    __shared__ int sa[64];
    sa[lane] = marg;
    __syncthreads();
    if (lane == 0) {
        for (int k = 1; k < 64; ++k)
            if (sl[k] < mine || (sl[k] == mine && sa[k] < marg)) { mine = sl[k]; marg = sa[k]; }
        loss_out[0] = (float)mine;
        arg_out[0] = marg;
    }
}

__global__ __launch_bounds__(256) void revssim_bwd_kernel(
    const float* __restrict__ hr, const uint8_t* __restrict__ mask, const float* __restrict__ pred, const int* __restrict__ arg,
    const double* __restrict__ mom, int S, int border, int B, float max_val, float eta, const float* __restrict__ upstream,
    float* __restrict__ dpred)
{
    extern __shared__ double sm2[];
    double* sW = sm2;                                                // [5][L]
    double* sG = sW + RS_NS * (S - 2 * border);                      // [L][L] dLoss/dC
    __shared__ double red[4];
    __shared__ double cA[RS_NS], cB[RS_NS], cC[RS_NS], cMuS[RS_NS], cMuH[RS_NS], cIw[RS_NS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = S - 2 * border, ns = 2 * border + 1;
    const int sft = arg[0], i = sft / ns, j = sft - i * ns;
    const double C1 = (0.01 * max_val) * (0.01 * max_val), C3 = 0.5 * (0.03 * max_val) * (0.03 * max_val);
    const double up = upstream ? (double)upstream[0] : 1.0;
    for (int k = tid; k < RS_NS * L; k += 256) sW[k] = rs_w1d(k % L, L, k / L);
    if (tid == 0) {
        RsTerms t;
        const double* m = mom + ((long)sft * B + b) * RS_NS * RS_NM;
        rs_terms(m, C1, C3, t);
        double lsum = 0.0;
#pragma unroll
        for (int s = 0; s < RS_NS; ++s) lsum += t.lum[s];
#pragma unroll
        for (int s = 0; s < RS_NS; ++s) {
            double pex = 1.0;                                        // product of contrast * structure over the OTHER scales
#pragma unroll
            for (int q = 0; q < RS_NS; ++q) if (q != s) pex *= t.con[q] * t.str[q];
            const double pcs = pex * t.con[s] * t.str[s];
            const double dl = t.muH[s] * t.muH[s] + t.muS[s] * t.muS[s] + C1;
            const double dlum = (2.0 * t.muH[s] * dl - (2.0 * t.muH[s] * t.muS[s] + C1) * 2.0 * t.muS[s]) / (dl * dl);   // d lum / d muS
            const double dc = t.sH[s] * t.sH[s] + t.sS[s] * t.sS[s] + C1;
            const double dcon = (2.0 * t.sH[s] * dc - (2.0 * t.sH[s] * t.sS[s] + C1) * 2.0 * t.sS[s]) / (dc * dc);       // d con / d sS
            const double ds = t.sH[s] * t.sS[s] + C3;
            const double dstr_cov = 2.0 / ds, dstr_sS = -(2.0 * t.cov[s] + C3) * t.sH[s] / (ds * ds);
            const double dT_con = lsum * pex * t.str[s], dT_str = lsum * pex * t.con[s];
            cA[s] = pcs * dlum;                                      // dT / d muS
            cB[s] = dT_con * dcon + dT_str * dstr_sS;                // dT / d sS
            cC[s] = dT_str * dstr_cov;                               // dT / d cov
            cMuS[s] = t.muS[s]; cMuH[s] = t.muH[s]; cIw[s] = 1.0 / m[s * RS_NM];
        }
    }
    const float* H = hr + (long)b * S * S;
    const uint8_t* M = mask + (long)b * S * S;
    const float* P = pred + (long)b * S * S;
    float* G = dpred + (long)b * S * S;
    double cnt = 0.0, dsum = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const float m = M[(i + r) * S + j + c] ? 1.f : 0.f;
        cnt += (double)m;
        dsum += (double)(H[(i + r) * S + j + c] - P[(border + r) * S + border + c] * m);
    }
    cnt = block_sum(cnt, red, tid);                                  // (barriers inside also publish sW and the c* scalars)
    dsum = block_sum(dsum, red, tid);
    const double bias = dsum / cnt;
    double gm = 0.0;
    for (int k = tid; k < L * L; k += 256) {
        const int r = k / L, c = k - r * L;
        const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
        const double h = (double)H[(i + r) * S + j + c];
        const double cc = ((double)P[(border + r) * S + border + c] + bias) * m;
        const double sg = h - cc > 0.0 ? 1.0 : (h - cc < 0.0 ? -1.0 : 0.0);
        double g = 0.0;
#pragma unroll
        for (int s = 0; s < RS_NS; ++s) {
            const double w = sW[s * L + r] * sW[s * L + c] * m * cIw[s];        // normalised weight
            g += -((double)eta / B) * w * (cA[s] + 2.0 * cB[s] * (cc - cMuS[s]) + cC[s] * (h - cMuH[s]))
                 - ((1.0 - (double)eta) / ((double)max_val * B)) * w * sg;
        }
        sG[k] = g;
        gm += g * m;
    }
    gm = block_sum(gm, red, tid);
    for (int k = tid; k < S * S; k += 256) {
        const int Y = k / S, X = k - Y * S, r = Y - border, c = X - border;
        float gk = 0.f;
        if (r >= 0 && r < L && c >= 0 && c < L) {
            const double m = M[(i + r) * S + j + c] ? 1.0 : 0.0;
            gk = (float)(m * (sG[r * L + c] - gm / cnt) * up);
        }
        G[k] = gk;
    }
}

size_t revssim_scratch_bytes(int B, int border) { const int ns = 2 * border + 1; return (size_t)ns * ns * B * RS_NS * RS_NM * sizeof(double); }

int revssim_forward(const float* hr, const uint8_t* mask, const float* pred, int B, int S, int border, float max_val, float eta,
                    double* scratch, float* loss, int* arg, hipStream_t s)
{
    const int L = S - 2 * border, ns = 2 * border + 1;
    if (B <= 0 || L < 2) { set_error("revssim_forward: bad shape", hipSuccess); return PROBAV_EINVAL; }
    hipLaunchKernelGGL(revssim_moments_kernel, dim3(B, ns * ns), dim3(256), (size_t)RS_NS * L * sizeof(double), s, hr, mask, pred, S, border, scratch, B);
    hipLaunchKernelGGL(revssim_select_kernel, dim3(1), dim3(64), 0, s, scratch, B, ns * ns, max_val, eta, loss, arg);
    return check_launch("revssim_forward");
}
int revssim_backward(const float* hr, const uint8_t* mask, const float* pred, const int* arg, const double* scratch, int B, int S,
                     int border, float max_val, float eta, const float* upstream, float* dpred, hipStream_t s)
{
    const int L = S - 2 * border;
    const size_t lds = ((size_t)RS_NS * L + (size_t)L * L) * sizeof(double);
    if (B <= 0 || L < 2 || lds > 64 * 1024) { set_error("revssim_backward: bad shape", hipSuccess); return PROBAV_EINVAL; }
    hipLaunchKernelGGL(revssim_bwd_kernel, dim3(B), dim3(256), lds, s, hr, mask, pred, arg, scratch, S, border, B, max_val, eta, upstream, dpred);
    return check_launch("revssim_backward");
}

// ---------------------------------------------------------------------------------------------------
// Keras Nadam (optimizer_v2; train.py:79-81) on the flat parameter buffer, one launch:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   theta -= lr * ( (1-mu_t) g / (1-Pi_t) + mu_{t+1} m / (1-Pi_t mu_{t+1}) ) / ( sqrt(v / (1-b2^t)) + eps )
// The step-dependent scalars (c_g = (1-mu_t)/(1-Pi_t), c_m = mu_{t+1}/(1-Pi_t mu_{t+1}), c_v = 1/(1-b2^t)) are computed by the
// host in double and passed by value.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nadam_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float c_g, float c_m, float c_v)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    theta[i] -= lr * (c_g * g + c_m * mi) / (sqrtf(vi * c_v) + eps);
}
int nadam_step(float* theta, const float* grad, float* m, float* v, long n, float lr, float b1, float b2, float eps,
               float c_g, float c_m, float c_v, hipStream_t s)
{
    if (n <= 0) return PROBAV_OK;
    hipLaunchKernelGGL(nadam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, theta, grad, m, v, n, lr, b1, b2, eps, c_g, c_m, c_v);
    return check_launch("nadam");
}

}  // namespace probav
