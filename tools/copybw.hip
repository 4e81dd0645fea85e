// Diagnostic (not part of the product): what a plain streaming kernel reaches on this device at the sizes of the step's small kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/copybw.hip -o tools/copybw.bin && tools/copybw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_direct.hip"
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy4x4(const float4* __restrict__ a, float4* __restrict__ b, long n)
{
    for (long i0 = (long)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += (long)gridDim.x * 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[i0 + 256 * k < n ? i0 + 256 * k : i0];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i0 + 256 * k < n) b[i0 + 256 * k] = v[k];
    }
}
__global__ __launch_bounds__(256) void write4(float4* __restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ a, float* __restrict__ sink, long n)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) *sink = s;
}
template <class F> static float timeit(F f, int iters = 50)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
int main()
{
    const long MB = 1 << 20;
    float4 *a, *b; float* sink;
    hipMalloc(&a, 1024 * MB); hipMalloc(&b, 1024 * MB); hipMalloc(&sink, 4);
    hipMemset(a, 0, 1024 * MB);
    for (long mb : {18L, 71L, 142L, 512L}) {
        const long n = mb * MB / 16;
        for (int grid : {1024, 2816, 8192, 32768}) {
            const float t1 = timeit([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, a, b, n); });
            const float t2 = timeit([&] { hipLaunchKernelGGL(copy4x4, dim3(grid), dim3(256), 0, 0, a, b, n); });
            const float t3 = timeit([&] { hipLaunchKernelGGL(write4, dim3(grid), dim3(256), 0, 0, b, n); });
            const float t4 = timeit([&] { hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, a, sink, n); });
            printf("%4ld MB, grid %5d: copy %6.1f us (%.2f TB/s moved)  copy x4 %6.1f us (%.2f)  write %6.1f us (%.2f)  read %6.1f us (%.2f)\n", mb, grid,
                   t1, 2.0 * mb * MB / t1 * 1e-6, t2, 2.0 * mb * MB / t2 * 1e-6, t3, 1.0 * mb * MB / t3 * 1e-6, t4, 1.0 * mb * MB / t4 * 1e-6);
        }
    }
    {   // the same copy on a different window of an 12-GB block at every launch (as in a training step: nothing is re-read from where it was a step ago)
        char* big; const long GB = 1l << 30, span = 12 * GB;
        if (hipMalloc(&big, span) == hipSuccess) {
            hipMemset(big, 0, span);
            for (long mb : {71L, 142L}) {
                const long n = mb * MB / 16, stride = 3 * mb * MB;
                long k = 0;
                const float t = timeit([&] { const long off = (k++ * stride) % (span - 2 * mb * MB - stride); hipLaunchKernelGGL(copy4x4, dim3(2816), dim3(256), 0, 0,
                                             reinterpret_cast<const float4*>(big + off), reinterpret_cast<float4*>(big + off + mb * MB), n); }, 60);
                printf("rotating windows of a 12-GB block, %ld MB: copy x4 %6.1f us (%.2f TB/s moved)\n", mb, t, 2.0 * mb * MB / t * 1e-6);
                // read what the PREVIOUS launch wrote (producer -> consumer, as between two kernels of the step)
                k = 0;
                const float t2 = timeit([&] { const long off = (k++ * mb * MB) % (span - 3 * mb * MB); hipLaunchKernelGGL(copy4x4, dim3(2816), dim3(256), 0, 0,
                                              reinterpret_cast<const float4*>(big + off), reinterpret_cast<float4*>(big + off + mb * MB), n); }, 60);
                printf("a chain through the block (each launch reads what the last one wrote), %ld MB: copy x4 %6.1f us (%.2f TB/s moved)\n", mb, t2, 2.0 * mb * MB / t2 * 1e-6);
            }
            {   // the step's reflect_fold on rotating windows (cold inputs, as behind its producer's 85 MB of stores in a 16-GB workspace)
                using namespace probav;
                unsigned* am2; hipMalloc(&am2, 4096); hipMemset(am2, 0, 4096);
                long k = 0; const long win = 256 * MB;
                const float t = timeit([&] { const long off = (k++ * win) % (span - 2 * win); reflect_fold(reinterpret_cast<const float*>(big + off), reinterpret_cast<float*>(big + off + 128 * MB), 128, 22, 22, 288, am2, 0); }, 60);
                printf("reflect_fold on rotating windows: %.1f us\n", t);
                k = 0;
                const float t3 = timeit([&] { const long off = (k++ * win) % (span - 2 * win);
                                              hipLaunchKernelGGL(write4, dim3(2816), dim3(256), 0, 0, reinterpret_cast<float4*>(big + off), 85 * MB / 16);
                                              reflect_fold(reinterpret_cast<const float*>(big + off), reinterpret_cast<float*>(big + off + 128 * MB), 128, 22, 22, 288, am2, 0); }, 60);
                const float t4 = timeit([&] { const long off = (k++ * win) % (span - 2 * win);
                                              hipLaunchKernelGGL(write4, dim3(2816), dim3(256), 0, 0, reinterpret_cast<float4*>(big + off), 85 * MB / 16); }, 60);
                printf("an 85-MB write + reflect_fold of it, rotating windows: %.1f us (the write alone %.1f)\n", t3, t4);
            }
            hipFree(big);
        }
    }
    {   // a streaming kernel BETWEEN matrix-dense kernels (as the small kernels of the step sit between the block kernels): does the chip serve it at the same rate?
        void* seed; float* sk; hipMalloc(&seed, 64 * 16); hipMemset(seed, 0x3c, 64 * 16); hipMalloc(&sk, 4096);
        std::vector<unsigned short> hs(512); unsigned long long q = 88172645463325252ull;
        for (auto& u : hs) { q ^= q << 13; q ^= q >> 7; q ^= q << 17; _Float16 hf = (_Float16)((float)((q >> 11) & 0xffffff) / 8388608.f - 1.f); u = *reinterpret_cast<unsigned short*>(&hf); }
        hipMemcpy(seed, hs.data(), 1024, hipMemcpyHostToDevice);
        const long mb = 71, n = mb * MB / 16;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int dense_us : {0, 100, 300, 1000, 3000}) {
            const int iters = dense_us * 50;                       // ~20 ns per 32x32x16 MFMA of the chain
            double tot = 0; int cnt = 0;
            for (int rep = 0; rep < 30; ++rep) {
                if (iters) probav::mfma_probe(seed, sk, iters, 1, 0, 0);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(copy4x4, dim3(2816), dim3(256), 0, 0, a, b, n);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 5) { tot += ms * 1e3; ++cnt; }
            }
            printf("copy of %ld MB right behind %4d us of dependent fp16 MFMAs on every CU: %.1f us (events around the copy alone)\n", mb, dense_us, tot / cnt);
        }
    }
    {   // the step's own streaming kernels at their sizes, back to back (inputs resident in the memory-side cache, as right behind their producer)
        using namespace probav;
        const int N = 128, H = 22, W = 22, TC = 9 * 32;
        unsigned* am; hipMalloc(&am, 65536); hipMemset(am, 0, 65536);
        const float t = timeit([&] { reflect_fold(reinterpret_cast<const float*>(a), reinterpret_cast<float*>(b), N, H, W, TC, am, 0); });
        const double mb = (double)N * ((H + 2) * (W + 2) + H * W) * TC * 4 / MB;
        printf("reflect_fold (%d x %d x %d x %d): %.1f us for %.0f MB = %.2f TB/s\n", N, H, W, TC, t, mb, mb * MB / t * 1e-6);
        {   // the same with DATA in the input and the amax slots cleared before every launch, as in a step: now the workgroups' atomicMax calls are live
            std::vector<float> hv((size_t)N * (H + 2) * (W + 2) * TC);
            unsigned long long q = 88172645463325252ull;
            for (auto& v : hv) { q ^= q << 13; q ^= q >> 7; q ^= q << 17; v = (float)((q >> 11) & 0xffffff) / 8388608.f - 1.f; }
            hipMemcpy(a, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
            const float t2 = timeit([&] { hipMemsetAsync(am, 0, 65536, 0); reflect_fold(reinterpret_cast<const float*>(a), reinterpret_cast<float*>(b), N, H, W, TC, am, 0); });
            const float t5 = timeit([&] { hipMemsetAsync(am, 0, 65536, 0); });
            const float t6 = timeit([&] { hipMemsetAsync(am, 0, 65536, 0); reflect_fold(reinterpret_cast<const float*>(a), reinterpret_cast<float*>(b), N, H, W, TC, nullptr, 0); });
            printf("reflect_fold on random data, amax slots cleared before each launch: %.1f us (the clear alone %.1f; without amax reporting %.1f)\n", t2, t5, t6);
        }
        float* fa = reinterpret_cast<float*>(a); float* fb = reinterpret_cast<float*>(b);
        float *w, *bias, *dw, *db, *part;
        hipMalloc(&w, 27 * 32 * 32 * 4); hipMalloc(&bias, 128); hipMalloc(&dw, 27 * 32 * 32 * 4); hipMalloc(&db, 128); hipMemset(w, 0, 27 * 32 * 32 * 4); hipMemset(bias, 0, 128);
        ConvGeom g1{N, 22, 22, 9, 1, 22, 22, 9, 32, 3, 3, 3, 1, 1, 1, 0, 1, 0};                      // mainConv1
        hipMalloc(&part, wgrad_partial_floats(g1) * 4 + 4096);
        const double vox = (double)N * 22 * 22 * 9;
        float t1 = timeit([&] { conv3d_cin1_forward(g1, fa, w, bias, fb, am, 0); });
        printf("conv3_cin1_fwd (mainConv1 forward): %.1f us; it writes %.0f MB (%.2f TB/s)\n", t1, vox * 32 * 4 / MB, vox * 32 * 4 / t1 * 1e-6);
        t1 = timeit([&] { conv3d_direct_wgrad(g1, fa, fb, fb, dw, db, part, 0); });
        printf("wgrad_cin1 (mainConv1 backward-filter, gated) + its sums: %.1f us; it reads %.0f MB (%.2f TB/s)\n", t1, vox * 65 * 4 / MB, vox * 65 * 4 / t1 * 1e-6);
        ConvGeom gu{N, 18, 18, 3, 32, 16, 16, 1, 9, 3, 3, 3, 0, 0, 0, 0, 0, 0};                      // upscaleConv1
        if (conv3d_up_forward_supported(gu)) { t1 = timeit([&] { conv3d_up_forward(gu, fa, w, bias, fb, 0); }); printf("conv3_up_fwd: %.1f us; it reads %.1f MB\n", t1, (double)N * 18 * 18 * 3 * 32 * 4 / MB); }
        ConvGeom gub{N, 16, 16, 1, 9, 18, 18, 3, 32, 3, 3, 3, 2, 2, 2, 0, 0, 0};
        if (conv3d_up_bwd_data_supported(gub)) { t1 = timeit([&] { conv3d_up_bwd_data(gub, fa, w, nullptr, fb, am, 0); }); printf("conv3_up_bwd_data: %.1f us; it writes %.1f MB\n", t1, (double)N * 18 * 18 * 3 * 32 * 4 / MB); }
    }
    return 0;
}
