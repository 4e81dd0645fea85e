// Diagnostic (not part of the product): the one-wave-per-SIMD fused pointwise backward (kernels_pw4.hip) against the general eight-wave form (pw_bwd_x6_kernel<H3>) on random data --
// element-wise agreement of dX, dW1, dW2, db1, db2 at several batch sizes, then both timed at the benchmark's shape (batch 128, T = 9).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I proba-v_amd/csrc -I include tools/pw4bench.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/pw4bench.bin
#include "kernels_x6.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
using namespace probav;

static unsigned long long g_s = 88172645463325252ull;
static float rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.f - 0.5f; }

struct Out { std::vector<float> dX, dW1, dW2, db1, db2; };
static double reldiff(const std::vector<float>& a, const std::vector<float>& b, double* amax_out = nullptr)
{
    double m = 0, d = 0;
    for (size_t i = 0; i < a.size(); ++i) { m = std::max(m, (double)std::fabs(a[i])); d = std::max(d, (double)std::fabs(a[i] - b[i])); if (std::isnan(b[i])) d = 1e30; }
    if (amax_out) *amax_out = m;
    return m > 0 ? d / m : d;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int D = 25, BMAX = 128;
    const long V = 22 * 22 * 9, nvmax = (long)BMAX * V;
    std::vector<float> hx((size_t)nvmax * 32), hd((size_t)nvmax * D), ho((size_t)nvmax * 32);
    for (auto& v : hx) v = rnd();
    for (auto& v : hd) v = rnd();
    for (auto& v : ho) v = rnd();
    float *x, *dT, *dOut, *dX, *w, *b1, *dW1, *dW2, *db1, *db2, *slabs;
    hipMalloc(&x, nvmax * 32 * 4); hipMalloc(&dT, nvmax * D * 4); hipMalloc(&dOut, nvmax * 32 * 4); hipMalloc(&dX, nvmax * 32 * 4);
    hipMemcpy(x, hx.data(), nvmax * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(dT, hd.data(), nvmax * D * 4, hipMemcpyHostToDevice);
    hipMemcpy(dOut, ho.data(), nvmax * 32 * 4, hipMemcpyHostToDevice);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> one(8192, 0x3f800000u); hipMemcpy(am_, one.data(), 8192 * 4, hipMemcpyHostToDevice); }
    const size_t fw = (size_t)H3_PW_FRAG_WORDS;
    hipMalloc(&w, 3 * fw * 4); hipMalloc(&b1, 256 * 4);
    {
        std::vector<unsigned> hw(3 * fw);
        for (auto& u : hw) {
            unsigned short hh[2];
            for (int q = 0; q < 2; ++q) { const float f = 2.f * rnd(); _Float16 hf = (_Float16)f; hh[q] = *reinterpret_cast<unsigned short*>(&hf); }
            u = hh[0] | ((unsigned)hh[1] << 16);
        }
        hipMemcpy(w, hw.data(), 3 * fw * 4, hipMemcpyHostToDevice);
        std::vector<float> hb(256);
        for (auto& v : hb) v = 4.f * rnd();
        hipMemcpy(b1, hb.data(), 256 * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&dW1, 8192 * 4); hipMalloc(&dW2, 256 * D * 4); hipMalloc(&db1, 256 * 4); hipMalloc(&db2, D * 4);
    hipMalloc(&slabs, mfma_pw_backward_slab_floats(D) * 4);
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.y = am_ + 4096;
    pam.w2c = am_ + 2100; pam.w1r = am_ + 2200;
    auto run = [&](int newk, int B, long vps) {
        pw4_set_enabled(newk);
        return x6_pw_backward(x, dT, dOut, w, w + fw, w + 2 * fw, b1, dX, dW1, dW2, db1, db2, slabs, (long)B * vps, vps, D, 2, pam, 0);
    };
    auto fetch = [&](int B, long vps) {
        Out o; o.dX.resize((size_t)B * vps * 32); o.dW1.resize(8192); o.dW2.resize(256 * D); o.db1.resize(256); o.db2.resize(D);
        hipDeviceSynchronize();
        hipMemcpy(o.dX.data(), dX, o.dX.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o.dW1.data(), dW1, 8192 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o.dW2.data(), dW2, 256 * D * 4, hipMemcpyDeviceToHost); hipMemcpy(o.db1.data(), db1, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o.db2.data(), db2, D * 4, hipMemcpyDeviceToHost);
        return o;
    };
    int bad = 0;
    const int cases[][2] = {{1, 4356}, {2, 4356}, {5, 4356}, {3, 100}, {7, 32}, {128, 4356}, {100, 4356}, {128, 4352}};
    for (auto& cs : cases) {
        const int B = cs[0]; const long vps = cs[1];
        hipMemset(dX, 0xff, nvmax * 32 * 4);
        if (run(0, B, vps)) { printf("general form failed: %s\n", last_error()); return 1; }
        Out a = fetch(B, vps);
        hipMemset(dX, 0xff, nvmax * 32 * 4); hipMemset(dW1, 0xff, 8192 * 4); hipMemset(dW2, 0xff, 256 * D * 4); hipMemset(db1, 0xff, 256 * 4); hipMemset(db2, 0xff, D * 4);
        if (run(1, B, vps)) { printf("new kernel failed: %s\n", last_error()); return 1; }
        Out b = fetch(B, vps);
        if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
        double m[5];
        const double e[5] = {reldiff(a.dX, b.dX, m), reldiff(a.dW1, b.dW1, m + 1), reldiff(a.dW2, b.dW2, m + 2), reldiff(a.db1, b.db1, m + 3), reldiff(a.db2, b.db2, m + 4)};
        const bool ok = e[0] < 2e-6 && e[1] < 2e-5 && e[2] < 2e-5 && e[3] < 2e-5 && e[4] < 2e-5;
        printf("B %3d vps %5ld: max |new - old| / max |old|:  dX %.2e  dW1 %.2e  dW2 %.2e  db1 %.2e  db2 %.2e   (max |old| %.3g %.3g %.3g %.3g %.3g)  %s\n", B, vps, e[0], e[1], e[2], e[3], e[4],
               m[0], m[1], m[2], m[3], m[4], ok ? "ok" : "MISMATCH");
        if (!ok) {
            ++bad;
            // where: first few mismatching elements of each tensor
            auto where = [&](const char* nm, const std::vector<float>& p, const std::vector<float>& q, double mx, int width) {
                int shown = 0;
                for (size_t i = 0; i < p.size() && shown < 6; ++i) if (std::fabs(p[i] - q[i]) > 1e-5 * mx || std::isnan(q[i])) { printf("    %s[%zu] (row %zu, col %zu): old %.6g new %.6g\n", nm, i, i / width, i % width, p[i], q[i]); ++shown; }
            };
            where("dX", a.dX, b.dX, m[0], 32); where("dW1", a.dW1, b.dW1, m[1], 256); where("dW2", a.dW2, b.dW2, m[2], D); where("db1", a.db1, b.db1, m[3], 256); where("db2", a.db2, b.db2, m[4], D);
        }
    }
    // timing at the benchmark's shape, alternating
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    const double gflop = (double)nvmax * 2e-9 * 29184;
    for (int pass = 0; pass < 4; ++pass)
        for (int k = 0; k < 2; ++k) {
            for (int i = 0; i < 3; ++i) run(k, 128, V);
            hipDeviceSynchronize();
            hipEventRecord(ea, 0);
            for (int i = 0; i < iters; ++i) run(k, 128, V);
            hipEventRecord(eb, 0); hipEventSynchronize(eb);
            float ms = 0; hipEventElapsedTime(&ms, ea, eb);
            const double us = ms * 1e3 / iters;
            if (pass) printf("pass %d  %s  %8.1f us per launch (incl. the slab sum)  %7.1f TFLOP/s algorithmic fp32\n", pass, k ? "pw_bwd_w4 (one wave per SIMD)" : "pw_bwd_x6<H3> (general form)", us, gflop / us * 1e3);
        }
    printf(bad ? "FAILED: %d case(s) mismatch\n" : "all cases agree\n", bad);
    return bad ? 2 : 0;
}
