// Diagnostic (not part of the product): the one-wave-per-SIMD fused pointwise forward (kernels_pf4.hip) against pw_fwd_x6_kernel<H3> -- the 32x32x16 arrangement whose products
// and order it repeats: the output and the hidden tile must agree BIT FOR BIT -- on several shapes, then timed against pw_fwd_h3k_kernel at the benchmark's shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I proba-v_amd/csrc -I include tools/pf4bench.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/pf4bench.bin
#include "kernels_x6.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
using namespace probav;
static unsigned long long g_s = 88172645463325252ull;
static float rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.f - 0.5f; }
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int BMAX = 128;
    const long V = 22 * 22 * 9, nvmax = (long)BMAX * V;
    std::vector<float> hx((size_t)nvmax * 32);
    for (auto& v : hx) v = rnd();
    float *x, *dec, *hd, *w, *b1, *b2;
    hipMalloc(&x, nvmax * 32 * 4); hipMalloc(&dec, nvmax * 32 * 4); hipMalloc(&hd, (size_t)8 * V * 256 * 4);
    hipMemcpy(x, hx.data(), nvmax * 32 * 4, hipMemcpyHostToDevice);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    std::vector<unsigned> slots(8192, 0x3f800000u);
    for (int i = 0; i < 2048; ++i) { const float f = 0.5f * (1.f + (i % 7)); memcpy(&slots[i], &f, 4); }
    for (int i = 2100; i < 2132; ++i) { const float f = 0.02f * (1.f + (i % 5)); memcpy(&slots[i], &f, 4); }
    hipMemcpy(am_, slots.data(), 8192 * 4, hipMemcpyHostToDevice);
    const size_t fw = (size_t)H3_PW_FRAG_WORDS;
    hipMalloc(&w, 2 * fw * 4); hipMalloc(&b1, 256 * 4); hipMalloc(&b2, 32 * 4);
    {
        std::vector<unsigned> hw(2 * fw);
        for (auto& u : hw) { unsigned short hh[2]; for (int q = 0; q < 2; ++q) { _Float16 hf = (_Float16)(2.f * rnd()); memcpy(&hh[q], &hf, 2); } u = hh[0] | ((unsigned)hh[1] << 16); }
        hipMemcpy(w, hw.data(), 2 * fw * 4, hipMemcpyHostToDevice);
        std::vector<float> hb(256); for (auto& v : hb) v = 4.f * rnd();
        hipMemcpy(b1, hb.data(), 256 * 4, hipMemcpyHostToDevice);
        std::vector<float> hb2(32); for (auto& v : hb2) v = 1e-3f * rnd();
        hipMemcpy(b2, hb2.data(), 32 * 4, hipMemcpyHostToDevice);
    }
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.y = am_ + 4096; pam.w2c = am_ + 2100; pam.w1r = am_ + 2200;
    const int cases[][3] = {{1, 4356, 25}, {2, 4356, 25}, {5, 4356, 25}, {3, 100, 25}, {7, 32, 25}, {8, 4356, 25}, {6, 700, 32}, {4, 40, 7}, {128, 4356, 25}};
    int bad = 0;
    for (auto& cs : cases) {
        const int B = cs[0]; const long vps = cs[1]; const int D = cs[2];
        const long nv = (long)B * vps;
        const bool dump = nv <= 8 * V;
        std::vector<float> ya((size_t)nv * D), yb((size_t)nv * D), ha, hb_;
        std::vector<unsigned> sa(B), sb(B);
        for (int k = 0; k < 2; ++k) {
            pf4_set_enabled(k);
            x6_pw_dump_from_forward_kernel(k);                          // k = 0: the dump of pw_fwd_x6_kernel<H3>; k = 1: of the new kernel
            hipMemset(dec, 0xff, nvmax * 32 * 4);
            if (dump) hipMemset(hd, 0xff, (size_t)nv * 256 * 4);
            hipMemset(am_ + 4096, 0, 2048 * 4);
            if (x6_pw_forward(x, w, w + fw, b1, b2, dec, nv, vps, D, 2, pam, 0, dump ? hd : nullptr)) { printf("launch failed: %s\n", last_error()); return 1; }
            hipDeviceSynchronize();
            if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
            hipMemcpy(k ? yb.data() : ya.data(), dec, (size_t)nv * D * 4, hipMemcpyDeviceToHost);
            hipMemcpy(k ? sb.data() : sa.data(), am_ + 4096, B * 4, hipMemcpyDeviceToHost);
            if (dump) { auto& h = k ? hb_ : ha; h.resize((size_t)nv * 256); hipMemcpy(h.data(), hd, (size_t)nv * 256 * 4, hipMemcpyDeviceToHost); }
        }
        size_t ny = 0, nh = 0, firsty = 0; double dmax = 0, m = 0;
        for (size_t i = 0; i < ya.size(); ++i) { m = std::max(m, (double)std::fabs(ya[i])); if (memcmp(&ya[i], &yb[i], 4)) { if (!ny) firsty = i; ++ny; dmax = std::max(dmax, (double)std::fabs(ya[i] - yb[i])); } }
        for (size_t i = 0; i < ha.size(); ++i) if (memcmp(&ha[i], &hb_[i], 4)) ++nh;
        size_t ns = 0; for (int i = 0; i < B; ++i) ns += sa[i] != sb[i];
        const bool ok = dump ? (ny == 0 && nh == 0 && ns == 0) : (dmax <= 2e-6 * m);      // (without a dump the other kernel is pw_fwd_h3k_kernel: another summation order)
        printf("B %3d vps %5ld D %2d: output values that differ %zu of %zu (max |diff| %.3g of %.3g), hidden values that differ %zu%s, amax slots %zu  %s\n", B, vps, D, ny, ya.size(), dmax, m, nh, dump ? "" : " (not dumped)", ns, ok ? "ok" : "MISMATCH");
        if (ns) for (int i = 0; i < B && i < 8; ++i) { float fa, fb; memcpy(&fa, &sa[i], 4); memcpy(&fb, &sb[i], 4); printf("    slot %d: %.9g against %.9g\n", i, fa, fb); }
        if (!ok) { ++bad; if (ny) { const size_t v = firsty / D; printf("    first: voxel %zu (sample %zu, voxel %zu of it) channel %zu: %.9g against %.9g\n", v, v / vps, v % vps, firsty % D, ya[firsty], yb[firsty]); } }
    }
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    x6_pw_dump_from_forward_kernel(0);
    const double gflop = (double)nvmax * 2e-9 * 14592;
    for (int pass = 0; pass < 4; ++pass)
        for (int k = 0; k < 2; ++k) {
            pf4_set_enabled(k);
            auto run = [&] { x6_pw_forward(x, w, w + fw, b1, b2, dec, nvmax, V, 25, 2, pam, 0, nullptr); };
            for (int i = 0; i < 3; ++i) run();
            hipDeviceSynchronize();
            hipEventRecord(ea, 0);
            for (int i = 0; i < iters; ++i) run();
            hipEventRecord(eb, 0); hipEventSynchronize(eb);
            float ms = 0; hipEventElapsedTime(&ms, ea, eb);
            const double us = ms * 1e3 / iters;
            if (pass) printf("pass %d  %s  %8.1f us per launch  %7.1f TFLOP/s algorithmic fp32\n", pass, k ? "pw_fwd_w4 (one wave per SIMD)" : "pw_fwd_h3k (eight waves)    ", us, gflop / us * 1e3);
        }
    printf(bad ? "FAILED: %d case(s) mismatch\n" : "all cases agree bit for bit\n", bad);
    return bad ? 2 : 0;
}
