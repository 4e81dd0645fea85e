// Diagnostic (not part of the product): phase stamps of pw_bwd_w4_kernel (kernels_pw4.hip built as probav::diag with -DPW4_STAMP) at the benchmark's shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DPW4_DIAG -DPW4_STAMP [-DPW4_...ablation] -I proba-v_amd/csrc -I include \
//         tools/pw4diag.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/pw4diag.bin
#include "../proba-v_amd/csrc/kernels_pw4.hip"
#include <vector>
#include <cstdio>
#include <algorithm>
using namespace probav;

static unsigned long long g_s = 88172645463325252ull;
static float rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (float)((g_s >> 11) & 0xffffff) / 16777216.f - 0.5f; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int D = 25, B = 128;
    const long V = 22 * 22 * 9, nv = (long)B * V;
    std::vector<float> hx((size_t)nv * 32);
    for (auto& v : hx) v = rnd();
    float *x, *dT, *dOut, *dX, *w, *b1, *dW1, *dW2, *db1, *db2, *slabs;
    hipMalloc(&x, nv * 32 * 4); hipMalloc(&dT, nv * D * 4); hipMalloc(&dOut, nv * 32 * 4); hipMalloc(&dX, nv * 32 * 4);
    hipMemcpy(x, hx.data(), nv * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(dT, hx.data(), nv * D * 4, hipMemcpyHostToDevice); hipMemcpy(dOut, hx.data(), nv * 32 * 4, hipMemcpyHostToDevice);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> one(8192, 0x3f800000u); hipMemcpy(am_, one.data(), 8192 * 4, hipMemcpyHostToDevice); }
    const size_t fw = (size_t)H3_PW_FRAG_WORDS;
    hipMalloc(&w, 3 * fw * 4); hipMalloc(&b1, 256 * 4);
    {
        std::vector<unsigned> hw(3 * fw);
        for (auto& u : hw) { unsigned short hh[2]; for (int q = 0; q < 2; ++q) { _Float16 hf = (_Float16)(2.f * rnd()); hh[q] = *reinterpret_cast<unsigned short*>(&hf); } u = hh[0] | ((unsigned)hh[1] << 16); }
        hipMemcpy(w, hw.data(), 3 * fw * 4, hipMemcpyHostToDevice);
        std::vector<float> hb(256); for (auto& v : hb) v = 4.f * rnd();
        hipMemcpy(b1, hb.data(), 256 * 4, hipMemcpyHostToDevice);
    }
    hipMalloc(&dW1, 8192 * 4); hipMalloc(&dW2, 256 * D * 4); hipMalloc(&db1, 256 * 4); hipMalloc(&db2, D * 4);
    hipMalloc(&slabs, mfma_pw_backward_slab_floats(D) * 4);
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.y = am_ + 4096; pam.w2c = am_ + 2100; pam.w1r = am_ + 2200;
    auto run = [&] { return diag::pw4_backward(x, dT, dOut, w, w + fw, w + 2 * fw, b1, dX, dW1, dW2, db1, db2, slabs, nv, V, D, pam, 0); };
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed: %s\n", last_error()); return 1; }
        hipDeviceSynchronize();
        hipEventRecord(ea, 0);
        for (int i = 0; i < iters; ++i) run();
        hipEventRecord(eb, 0); hipEventSynchronize(eb);
        float ms = 0; hipEventElapsedTime(&ms, ea, eb);
        printf("pass %d: %.1f us per launch (incl. the slab sum)\n", pass, ms * 1e3 / iters);
    }
#ifdef PW4_STAMP
    std::vector<unsigned long long> st(1024 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(diag::g_pw4_stamps), st.size() * 8);
    const char* nm[10] = {"(b) + relu", "(e) + gate", "(d) + gate, cut H'", "(c) + cut dH'", "(a) + cut dH', next operands", "boundary: (e), (d), staging", "boundary: (c), reads", "boundary: (a), dX rows", "whole run (cycles)", "whole run (100 MHz ticks)"};
    auto med = [&](int k) { std::vector<double> v; for (int wv = 0; wv < 1024; ++wv) if (st[wv * 16 + 8]) v.push_back((double)st[wv * 16 + k]); std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    const double tiles = 137.0 / 8.0;
    for (int k = 0; k < 10; ++k) printf("  slot %d  %-34s median %10.0f   per tile %8.0f   per chunk iteration %7.0f\n", k, nm[k], med(k), med(k) / tiles, med(k) / tiles / 8);
    printf("  in-kernel clock %.2f GHz; run %.1f us\n", med(8) / med(9) * 0.1, med(9) * 0.01);
    {   // outside the loop (100-MHz ticks -> us): prologue (entry -> first barrier passed), wait at the barrier behind the loop, the LDS sums and slab stores
        std::vector<double> pro, bw, epi, runs;
        for (int wv = 0; wv < 1024; ++wv) if (st[wv * 16 + 8]) { pro.push_back(st[wv * 16 + 10] * 0.01); bw.push_back((double)(st[wv * 16 + 12] - st[wv * 16 + 11]) * 0.01); epi.push_back((double)(st[wv * 16 + 13] - st[wv * 16 + 12]) * 0.01); runs.push_back(st[wv * 16 + 9] * 0.01); }
        auto md = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
        auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
        printf("  prologue median %.1f us (max %.1f) | run min %.1f median %.1f max %.1f | wait at the barrier behind the loop median %.1f (max %.1f) | sums + slab stores median %.1f (max %.1f)\n",
               md(pro), mx(pro), mn(runs), md(runs), mx(runs), md(bw), mx(bw), md(epi), mx(epi));
    }
#endif
    return 0;
}
