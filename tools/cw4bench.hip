// Diagnostic (not part of the product): the one-wave-per-SIMD 3x3x3 convolution (kernels_cw4.hip) against the eight-wave piece-ring kernel (conv3_pp_kernel) on random data --
// element-wise agreement of the output and of its amax slots at several batch sizes / depths, with and without skip and bias, then both timed at the benchmark's shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I proba-v_amd/csrc -I include tools/cw4bench.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/cw4bench.bin
#include "kernels_x6.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
using namespace probav;

static unsigned long long g_s = 88172645463325252ull;
static float rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.f - 0.5f; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int BMAX = 128;
    const long VMAX = 22 * 22 * 9, nvmax = (long)BMAX * VMAX;
    std::vector<float> hx((size_t)nvmax * 32), hs((size_t)nvmax * 32);
    for (auto& v : hx) v = rnd();
    for (auto& v : hs) v = 3.f * rnd();
    float *x, *sk, *y, *wf, *bias;
    hipMalloc(&x, nvmax * 32 * 4); hipMalloc(&sk, nvmax * 32 * 4); hipMalloc(&y, nvmax * 32 * 4);
    hipMemcpy(x, hx.data(), nvmax * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(sk, hs.data(), nvmax * 32 * 4, hipMemcpyHostToDevice);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    std::vector<unsigned> slots(8192, 0x3f800000u);
    for (int i = 0; i < 2048; ++i) { const float f = 0.5f * (1.f + (i % 7)); slots[i] = *reinterpret_cast<const unsigned*>(&f); }             // activations: per sample, different binades
    for (int i = 2048; i < 4096; ++i) { const float f = 0.01f * (1.f + (i % 13)); slots[i] = *reinterpret_cast<const unsigned*>(&f); }        // filters: per output channel
    hipMalloc(&wf, X6_CONV_FRAG_WORDS * 4); hipMalloc(&bias, 32 * 4);
    {
        std::vector<unsigned> hw(X6_CONV_FRAG_WORDS);
        for (auto& u : hw) {
            unsigned short hh[2];
            for (int q = 0; q < 2; ++q) { const float f = 2.f * rnd(); _Float16 hf = (_Float16)f; hh[q] = *reinterpret_cast<unsigned short*>(&hf); }
            u = hh[0] | ((unsigned)hh[1] << 16);
        }
        hipMemcpy(wf, hw.data(), X6_CONV_FRAG_WORDS * 4, hipMemcpyHostToDevice);
        std::vector<float> hb(32);
        for (auto& v : hb) v = 1e-3f * rnd();
        hipMemcpy(bias, hb.data(), 32 * 4, hipMemcpyHostToDevice);
    }
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = am_ + 4096;
    struct Case { int B, H, W, T, Cin, Cout, skip, bias, relu; };
    const Case cases[] = {
        {1, 22, 22, 9, 25, 32, 1, 1, 0}, {2, 22, 22, 9, 25, 32, 1, 1, 0}, {3, 22, 22, 9, 25, 32, 0, 0, 1}, {128, 22, 22, 9, 25, 32, 1, 1, 0},
        {1, 22, 22, 9, 32, 25, 0, 0, 0}, {2, 22, 22, 9, 32, 25, 0, 1, 0}, {5, 22, 22, 9, 32, 25, 1, 1, 1}, {128, 22, 22, 9, 32, 25, 0, 0, 0},
        {2, 22, 22, 9, 32, 32, 1, 1, 0}, {3, 22, 22, 7, 25, 32, 1, 1, 0}, {3, 22, 22, 7, 32, 25, 0, 0, 0}, {2, 16, 16, 9, 25, 32, 1, 1, 0}, {4, 10, 22, 9, 32, 25, 0, 0, 0},
        {100, 22, 22, 9, 25, 32, 1, 1, 0}, {128, 22, 22, 9, 25, 32, 0, 0, 0}, {64, 22, 22, 9, 25, 32, 0, 1, 0}, {6, 22, 22, 7, 25, 32, 0, 0, 1},      // (128 x T 9 and T 7: strips of 4 m + 1 tiles, the lone last tile's k-split)
    };
    int bad = 0;
    for (const Case& c : cases) {
        ConvGeom g{c.B, c.H, c.W, c.T, c.Cin, c.H, c.W, c.T, c.Cout, 3, 3, 3, 1, 1, 1, 0, c.relu, 0};
        const size_t ny = (size_t)c.B * c.H * c.W * c.T * c.Cout;
        std::vector<float> ya(ny), yb(ny);
        std::vector<unsigned> sa(c.B), sb(c.B);
        if (!cw4_conv_supported(g, nullptr)) { printf("B %3d %dx%dx%d %d->%d: not taken by the new kernel\n", c.B, c.H, c.W, c.T, c.Cin, c.Cout); continue; }
        for (int k = 0; k < 2; ++k) {
            cw4_set_enabled(k);
            hipMemset(y, 0xff, nvmax * 32 * 4);
            hipMemcpy(am_, slots.data(), 8192 * 4, hipMemcpyHostToDevice);
            hipMemset(am_ + 4096, 0, 2048 * 4);
            if (x6_conv_strip_forward(g, x, nullptr, wf, c.bias ? bias : nullptr, c.skip ? sk : nullptr, y, 2, am, 0)) { printf("launch failed: %s\n", last_error()); return 1; }
            hipDeviceSynchronize();
            if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
            hipMemcpy(k ? yb.data() : ya.data(), y, ny * 4, hipMemcpyDeviceToHost);
            hipMemcpy(k ? sb.data() : sa.data(), am_ + 4096, c.B * 4, hipMemcpyDeviceToHost);
        }
        double m = 0, d = 0; size_t nbad = 0, first = 0;
        for (size_t i = 0; i < ny; ++i) m = std::max(m, (double)std::fabs(ya[i]));
        for (size_t i = 0; i < ny; ++i) { const double e = std::isnan(yb[i]) ? 1e30 : std::fabs((double)ya[i] - yb[i]); d = std::max(d, e); if (e > 2e-6 * m) { if (!nbad) first = i; ++nbad; } }
        double ds = 0;
        for (int i = 0; i < c.B; ++i) { const float fa = *reinterpret_cast<float*>(&sa[i]), fb = *reinterpret_cast<float*>(&sb[i]); ds = std::max(ds, (double)std::fabs(fa - fb) / std::max(1e-30, (double)fa)); }
        const bool ok = nbad == 0 && ds < 1e-5;
        printf("B %3d %2dx%2dx%d %d->%d skip %d bias %d relu %d: max |new - old| / max |old| %.2e (max |old| %.4g), amax slots %.1e  %s\n", c.B, c.H, c.W, c.T, c.Cin, c.Cout, c.skip, c.bias, c.relu,
               m > 0 ? d / m : d, m, ds, ok ? "ok" : "MISMATCH");
        if (!ok) {
            ++bad;
            int shown = 0;
            for (size_t i = first; i < ny && shown < 8; ++i) if (std::isnan(yb[i]) || std::fabs((double)ya[i] - yb[i]) > 2e-6 * m) {
                const size_t v = i / c.Cout; const int ch = (int)(i % c.Cout);
                const int t = (int)(v % c.T), w = (int)((v / c.T) % c.W), h = (int)((v / c.T / c.W) % c.H), n = (int)(v / c.T / c.W / c.H);
                printf("    y[n %d h %d w %d t %d ch %d]: old %.6g new %.6g   (%zu mismatching in all)\n", n, h, w, t, ch, ya[i], yb[i], nbad); ++shown;
            }
        }
    }
    // timing at the benchmark's shapes, alternating
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    hipMemcpy(am_, slots.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int dir = 0; dir < 2; ++dir) {
        ConvGeom g{BMAX, 22, 22, 9, dir ? 32 : 25, 22, 22, 9, dir ? 25 : 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        const double gflop = (double)nvmax * 2e-9 * 27 * 25 * 32;
        for (int pass = 0; pass < 4; ++pass)
            for (int k = 0; k < 2; ++k) {
                cw4_set_enabled(k);
                auto run = [&] { x6_conv_strip_forward(g, x, nullptr, wf, bias, dir ? nullptr : sk, y, 2, am, 0); };
                for (int i = 0; i < 3; ++i) run();
                hipDeviceSynchronize();
                hipEventRecord(ea, 0);
                for (int i = 0; i < iters; ++i) run();
                hipEventRecord(eb, 0); hipEventSynchronize(eb);
                float ms = 0; hipEventElapsedTime(&ms, ea, eb);
                const double us = ms * 1e3 / iters;
                if (pass) printf("pass %d  %s  %s  %8.1f us per launch  %7.1f TFLOP/s algorithmic fp32\n", pass, dir ? "backward-data 32->25" : "forward 25->32 + skip", k ? "conv3_w4 (one wave per SIMD)" : "conv3_pp (eight waves)     ", us, gflop / us * 1e3);
            }
    }
    printf(bad ? "FAILED: %d case(s) mismatch\n" : "all cases agree\n", bad);
    return bad ? 2 : 0;
}
