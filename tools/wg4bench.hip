// Diagnostic (not part of the product): the one-wave-per-SIMD backward-filter kernel (kernels_wg4.hip) against the general form (conv3_wgrad_x6_kernel<25, false, H3>) on random
// data -- element-wise agreement of dW and db at several batch sizes / depths / heights, then both timed at the benchmark's shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I proba-v_amd/csrc -I include tools/wg4bench.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/wg4bench.bin
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
    std::vector<float> hx((size_t)nvmax * 32), hd((size_t)nvmax * 32), hg((size_t)nvmax * 32);
    for (auto& v : hx) v = rnd();
    for (auto& v : hd) v = 3.f * rnd();
    for (auto& v : hg) v = rnd() + 0.2f;                                 // the layer's output: 70 % of the gates open
    float *x, *d, *gt, *dw, *db, *part;
    hipMalloc(&x, nvmax * 32 * 4); hipMalloc(&d, nvmax * 32 * 4); hipMalloc(&gt, nvmax * 32 * 4);
    hipMemcpy(x, hx.data(), nvmax * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), nvmax * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(gt, hg.data(), nvmax * 32 * 4, hipMemcpyHostToDevice);
    const long nwmax = 27 * 32 * 32;
    hipMalloc(&dw, nwmax * 4); hipMalloc(&db, 32 * 4);
    hipMalloc(&part, 2 * 256 * (nwmax + 32) * 4 + 4096);
    unsigned* am_; hipMalloc(&am_, 4096 * 4);
    std::vector<unsigned> slots(4096);
    for (int i = 0; i < 2048; ++i) { const float f = 0.5f; slots[i] = *reinterpret_cast<const unsigned*>(&f); }
    for (int i = 2048; i < 4096; ++i) { const float f = 1.5f; slots[i] = *reinterpret_cast<const unsigned*>(&f); }
    hipMemcpy(am_, slots.data(), 4096 * 4, hipMemcpyHostToDevice);
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = nullptr;
    struct Case { int B, H, T, red; };                                  // T = output depth; red: a reducer layer (32 channels, mirrored rows / columns, input depth T + 2, gate)
    const Case cases[] = {{1, 22, 9, 0}, {2, 22, 9, 0}, {3, 22, 9, 0}, {128, 22, 9, 0}, {100, 22, 9, 0}, {5, 22, 7, 0}, {2, 10, 9, 0}, {7, 3, 9, 0}, {256, 1, 7, 0}, {64, 22, 9, 0},
                          {50, 22, 9, 0}, {37, 22, 7, 0}, {37, 21, 9, 0}, {1, 22, 7, 1}, {3, 22, 5, 1}, {37, 22, 7, 1}, {2, 22, 3, 1}, {128, 22, 7, 1}, {128, 22, 5, 1}, {128, 22, 3, 1}, {5, 2, 7, 1}, {100, 22, 3, 1}};
    int bad = 0;
    for (const Case& c : cases) {
        ConvGeom g = c.red ? ConvGeom{c.B, c.H, 22, c.T + 2, 32, c.H, 22, c.T, 32, 3, 3, 3, 1, 1, 0, 1, 1, 0} : ConvGeom{c.B, c.H, 22, c.T, 25, c.H, 22, c.T, 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        const float* gate = c.red ? gt : nullptr;
        const long nw = 27L * g.Cin * 32;
        if (!wg4_wgrad_supported(g, gate)) { printf("B %3d %dx22x%d%s: not taken by the new kernel\n", c.B, c.H, c.T, c.red ? " reducer" : ""); continue; }
        std::vector<float> wa(nw), wb(nw), ba(32), bb(32);
        for (int k = 0; k < 2; ++k) {
            wg4_set_enabled(k);
            hipMemset(dw, 0xff, nw * 4); hipMemset(db, 0xff, 32 * 4);
            if (x6_conv_wgrad(g, x, d, gate, dw, db, part, 2, am, 0)) { printf("launch failed: %s\n", last_error()); return 1; }
            hipDeviceSynchronize();
            if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
            hipMemcpy(k ? wb.data() : wa.data(), dw, nw * 4, hipMemcpyDeviceToHost);
            hipMemcpy(k ? bb.data() : ba.data(), db, 32 * 4, hipMemcpyDeviceToHost);
        }
        double m = 0, dd = 0, mb = 0, ddb = 0; long nbad = 0, first = -1;
        for (long i = 0; i < nw; ++i) m = std::max(m, (double)std::fabs(wa[i]));
        for (long i = 0; i < nw; ++i) { const double e = std::isnan(wb[i]) ? 1e30 : std::fabs((double)wa[i] - wb[i]); dd = std::max(dd, e); if (e > 2e-6 * m) { if (first < 0) first = i; ++nbad; } }
        for (int i = 0; i < 32; ++i) { mb = std::max(mb, (double)std::fabs(ba[i])); ddb = std::max(ddb, std::isnan(bb[i]) ? 1e30 : std::fabs((double)ba[i] - bb[i])); }
        const bool ok = nbad == 0 && ddb <= 2e-6 * std::max(mb, 1.0) * std::sqrt((double)c.B * c.H * 22 * c.T);
        printf("B %3d %2dx22x%d%s: max |new - old| / max |old| dW %.2e (max %.4g), db %.2e (max %.4g)  %s\n", c.B, c.H, c.T, c.red ? " reducer" : "", m > 0 ? dd / m : dd, m, mb > 0 ? ddb / mb : ddb, mb, ok ? "ok" : "MISMATCH");
        if (!ok) {
            ++bad;
            int shown = 0;
            for (long i = std::max(first, 0L); i < nw && shown < 8; ++i) if (std::isnan(wb[i]) || std::fabs((double)wa[i] - wb[i]) > 2e-6 * m) {
                printf("    dW[tap %ld ci %ld co %ld]: old %.6g new %.6g   (%ld mismatching in all)\n", i / (32 * g.Cin), (i / 32) % g.Cin, i % 32, wa[i], wb[i], nbad); ++shown;
            }
        }
    }
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    for (int shape = 0; shape < 4; ++shape) {
        const int To = shape == 0 ? 9 : 9 - 2 * shape;
        ConvGeom g = shape ? ConvGeom{BMAX, 22, 22, To + 2, 32, 22, 22, To, 32, 3, 3, 3, 1, 1, 0, 1, 1, 0} : ConvGeom{BMAX, 22, 22, 9, 25, 22, 22, 9, 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        const float* gate = shape ? gt : nullptr;
        const double gflop = (double)BMAX * 22 * 22 * To * 2e-9 * 27 * g.Cin * 32;
        for (int pass = 0; pass < 3; ++pass)
            for (int k = 0; k < 2; ++k) {
                wg4_set_enabled(k);
                auto run = [&] { x6_conv_wgrad(g, x, d, gate, dw, db, part, 2, am, 0); };
                for (int i = 0; i < 3; ++i) run();
                hipDeviceSynchronize();
                hipEventRecord(ea, 0);
                for (int i = 0; i < iters; ++i) run();
                hipEventRecord(eb, 0); hipEventSynchronize(eb); hipDeviceSynchronize();
                float ms = 0; hipEventElapsedTime(&ms, ea, eb);
                const double us = ms * 1e3 / iters;
                if (pass) printf("pass %d  %s depth %d  %s  %8.1f us per launch (+ its slab sum on the forked stream)  %7.1f TFLOP/s algorithmic fp32\n", pass, shape ? "reducer 32->32" : "normConv 25->32", To,
                                 k ? "conv3_wgrad_w4 (one wave per SIMD)" : "conv3_wgrad_x6 (eight waves)      ", us, gflop / us * 1e3);
            }
    }
    printf(bad ? "FAILED: %d case(s) mismatch\n" : "all cases agree\n", bad);
    return bad ? 2 : 0;
}
