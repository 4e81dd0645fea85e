// Diagnostic (not part of the product): phase stamps / ablations of conv3_wgrad_w4_kernel (kernels_wg4.hip built as probav::diag with -DWG4_STAMP) at the benchmark's shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 -DWG4_DIAG -DWG4_STAMP [-DWG4_ABL_...] \
//         -I proba-v_amd/csrc -I include tools/wg4diag.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/wg4diag.bin
#include "../proba-v_amd/csrc/kernels_wg4.hip"
#include <vector>
#include <cstdio>
#include <algorithm>
#include <cmath>
using namespace probav;

static unsigned long long g_s = 88172645463325252ull;
static float rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.f - 0.5f; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int B = 128;
    const long V = 22 * 22 * 9, nv = (long)B * V;
    std::vector<float> hx((size_t)nv * 32);
    for (auto& v : hx) v = rnd();
    float *x, *d, *dw, *db, *part;
    hipMalloc(&x, nv * 25 * 4); hipMalloc(&d, nv * 32 * 4);
    hipMemcpy(x, hx.data(), nv * 25 * 4, hipMemcpyHostToDevice); hipMemcpy(d, hx.data(), nv * 32 * 4, hipMemcpyHostToDevice);
    const long nw = 27 * 25 * 32;
    hipMalloc(&dw, nw * 4); hipMalloc(&db, 32 * 4); hipMalloc(&part, 256 * (nw + 32) * 4 + 4096);
    unsigned* am_; hipMalloc(&am_, 4096 * 4);
    { std::vector<unsigned> one(4096, 0x3f800000u); hipMemcpy(am_, one.data(), 4096 * 4, hipMemcpyHostToDevice); }
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = nullptr;
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    ConvGeom g{B, 22, 22, 9, 25, 22, 22, 9, 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};
    auto run = [&] { return diag::wg4_conv_wgrad(g, x, d, nullptr, dw, db, part, am, 0); };
    {   // agreement with the general form (the library's kernel) on this input
        std::vector<float> wa(nw), wb(nw);
        wg4_set_enabled(0);
        x6_conv_wgrad(g, x, d, nullptr, dw, db, part, 2, am, 0); hipDeviceSynchronize();
        hipMemcpy(wa.data(), dw, nw * 4, hipMemcpyDeviceToHost);
        hipMemset(dw, 0xff, nw * 4);
        run(); hipDeviceSynchronize();
        hipMemcpy(wb.data(), dw, nw * 4, hipMemcpyDeviceToHost);
        double m = 0, dd = 0;
        for (long i = 0; i < nw; ++i) m = std::max(m, (double)std::fabs(wa[i]));
        for (long i = 0; i < nw; ++i) dd = std::max(dd, std::isnan(wb[i]) ? 1e30 : std::fabs((double)wa[i] - wb[i]));
        printf("max |diag - general| / max |general| = %.2e\n", dd / m);
    }
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed: %s\n", last_error()); return 1; }
        hipDeviceSynchronize();
        hipEventRecord(ea, 0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(eb, 0); hipEventSynchronize(eb); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, ea, eb);
        printf("pass %d: %.1f us per launch + slab sum (stamped build)\n", pass, ms * 1e3 / iters);
    }
#ifdef WG4_STAMP
    std::vector<unsigned long long> st(1024 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(diag::g_wg4_stamps), st.size() * 8);
    const char* nm[8] = {"whole kernel (cycles)", "whole kernel (100 MHz ticks)", "prologue", "prologue: entry -> LDS cleared (issue)", "waits at the barriers", "epilogue (slab stores)", "prologue: entry -> behind the clear barrier", "prologue: three rows + dY units cut"};
    auto stat = [&](int k, double& md, double& mx, double& mn) { std::vector<double> v; for (int wv = 0; wv < 1024; ++wv) if (st[wv * 8]) v.push_back((double)st[wv * 8 + k]); std::sort(v.begin(), v.end()); md = v[v.size() / 2]; mx = v.back(); mn = v[0]; };
    for (int k = 0; k < 8; ++k) { double md, mx, mn; stat(k, md, mx, mn); printf("  slot %d  %-32s median %10.0f  min %10.0f  max %10.0f   per row %8.0f\n", k, nm[k], md, mn, mx, md / 11.0); }
    { double c, t, a, b; stat(0, c, a, b); stat(1, t, a, b); printf("  in-kernel clock %.2f GHz; kernel %.1f us per wave\n", c / t * 0.1, t * 0.01); }
    for (int w = 0; w < 4; ++w) printf("  workgroup 0 wave %d: whole %llu, rows %llu, barrier waits %llu\n", w, st[w * 8], st[w * 8 + 3], st[w * 8 + 4]);
#endif
    return 0;
}
