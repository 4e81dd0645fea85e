// Diagnostic (not part of the product): phase stamps of conv3_w4_kernel (kernels_cw4.hip built as probav::diag with -DCW4_STAMP) at the benchmark's shapes.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -mllvm -pragma-unroll-threshold=4000000 -DCW4_DIAG -DCW4_STAMP [-DCW4_...ablation] \
//         -I proba-v_amd/csrc -I include tools/cw4diag.hip -L proba-v_amd/csrc -lprobav_hip -Wl,-rpath,'$ORIGIN/../proba-v_amd/csrc' -o tools/cw4diag.bin
#include "../proba-v_amd/csrc/kernels_cw4.hip"
#include <vector>
#include <cstdio>
#include <algorithm>
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
    float *x, *sk, *y, *wf, *bias;
    hipMalloc(&x, nv * 32 * 4); hipMalloc(&sk, nv * 32 * 4); hipMalloc(&y, nv * 32 * 4);
    hipMemcpy(x, hx.data(), nv * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(sk, hx.data(), nv * 32 * 4, hipMemcpyHostToDevice);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> one(8192, 0x3f800000u); hipMemcpy(am_, one.data(), 8192 * 4, hipMemcpyHostToDevice); }
    hipMalloc(&wf, X6_CONV_FRAG_WORDS * 4); hipMalloc(&bias, 32 * 4); hipMemset(bias, 0, 32 * 4);
    {
        std::vector<unsigned> hw(X6_CONV_FRAG_WORDS);
        for (auto& u : hw) { unsigned short hh[2]; for (int q = 0; q < 2; ++q) { _Float16 hf = (_Float16)(2.f * rnd()); hh[q] = *reinterpret_cast<unsigned short*>(&hf); } u = hh[0] | ((unsigned)hh[1] << 16); }
        hipMemcpy(wf, hw.data(), X6_CONV_FRAG_WORDS * 4, hipMemcpyHostToDevice);
    }
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = am_ + 4096;
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    for (int dir = 0; dir < 2; ++dir) {
        ConvGeom g{B, 22, 22, 9, dir ? 32 : 25, 22, 22, 9, dir ? 25 : 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        auto run = [&] { return diag::cw4_conv_forward(g, x, wf, bias, dir ? nullptr : sk, y, am, 0); };
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed: %s\n", last_error()); return 1; }
            hipDeviceSynchronize();
            hipEventRecord(ea, 0);
            for (int i = 0; i < iters; ++i) run();
            hipEventRecord(eb, 0); hipEventSynchronize(eb);
            float ms = 0; hipEventElapsedTime(&ms, ea, eb);
            printf("%s pass %d: %.1f us per launch (stamped build)\n", dir ? "backward-data 32->25" : "forward 25->32", pass, ms * 1e3 / iters);
        }
#ifdef CW4_STAMP
        std::vector<unsigned long long> st(1024 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(diag::g_cw4_stamps), st.size() * 8);
        const char* nm[8] = {"whole kernel (cycles)", "whole kernel (100 MHz ticks)", "prologue: the rest", "prologue: constants", "prologue: requests issued", "prologue: ring cleared, barrier", "prologue: first rows cut", "prologue: W1 stored, tables"};
        auto stat = [&](int k, double& md, double& mx, double& mn) { std::vector<double> v; for (int wv = 0; wv < 1024; ++wv) if (st[wv * 8]) v.push_back((double)st[wv * 8 + k]); std::sort(v.begin(), v.end()); md = v[v.size() / 2]; mx = v.back(); mn = v[0]; };
        const double rounds = 18.0;
        for (int k = 0; k < 8; ++k) { double md, mx, mn; stat(k, md, mx, mn); printf("  slot %d  %-30s median %10.0f  min %10.0f  max %10.0f   per round %8.0f\n", k, nm[k], md, mn, mx, md / rounds); }
        { double c, t, a, b; stat(0, c, a, b); stat(1, t, a, b); printf("  in-kernel clock %.2f GHz; kernel %.1f us per wave\n", c / t * 0.1, t * 0.01); }
#endif
    }
    return 0;
}
