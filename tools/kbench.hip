// Diagnostic (not part of the product): times the hot H3 kernels in isolation on random data at the benchmark's shapes (batch 128, T = 9).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I proba-v_amd/csrc tools/kbench.hip -o tools/kbench.bin && tools/kbench.bin [iters]
// -DPROBAV_STAMP_CLOCK adds each kernel's in-kernel clock (two stamps per wave, none inside the loops).
// -DKB_OLD builds against the round-1 signatures (per-tensor amax slots), for A/B runs of two source trees.
// Ablation switches of conv3_pp_kernel (timing only, results are wrong): -DPPX_IDLE (the finishing half does nothing: taps alone),
// -DPPX_NOTAPS (no MFMA loop: finishing + staging alone), -DPPX_NOEPI (no epilogue / skip / stores), -DPPX_NOSTAGE (no row staging).
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_mfma.hip"
#include "../proba-v_amd/csrc/kernels_x6.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
using namespace probav;

static int g_pass = 0;
template <class F> static float timeit(const char* name, int iters, double gflop, F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const float us = ms * 1e3f / iters;
    if (g_pass) printf("%-44s %8.1f us   %7.1f TFLOP/s (algorithmic fp32)", name, us, gflop / us * 1e3);
#ifdef PROBAV_STAMP_CLOCK
    if (g_pass) {       // in-kernel clock of the LAST launch: per wave cycles / 100 MHz ticks between the kernel's two stamps; median over the waves that wrote them
        static std::vector<unsigned long long> st(8192 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        std::vector<double> clk, cyc;
        for (size_t w = 0; w < 8192; ++w) if (st[w * 8 + 1] > 0) { clk.push_back((double)st[w * 8] / (double)st[w * 8 + 1] * 0.1); cyc.push_back((double)st[w * 8]); }
        if (!clk.empty()) { std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            printf("   clock %.2f GHz (p10 %.2f, p90 %.2f), %.0f cycles between the stamps", clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], cyc[cyc.size() / 2]); }
        {   // anatomy of the LAST launch (100-MHz ticks -> us): arrival of the waves, prologue (entry -> first stamp), loop, how long the last wave runs beyond the median one
            std::vector<double> en, pr, lp, ex;
            double e0 = 1e30, x1 = 0;
            for (size_t w = 0; w < 8192; ++w) if (st[w * 8 + 1] > 0 && st[w * 8 + 2] > 0) { en.push_back((double)st[w * 8 + 2]); pr.push_back((double)(st[w * 8 + 3] - st[w * 8 + 2])); lp.push_back((double)st[w * 8 + 1]); ex.push_back((double)st[w * 8 + 4]); e0 = std::min(e0, (double)st[w * 8 + 2]); x1 = std::max(x1, (double)st[w * 8 + 4]); }
            if (!en.empty()) {
                auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
                printf("\n      launch anatomy: first entry -> last exit %.1f us | entry: median +%.1f, last +%.1f | prologue median %.1f (max %.1f) | loop median %.1f (max %.1f) | exit: median %.1f before the last",
                       (x1 - e0) * 0.01, (med(en) - e0) * 0.01, (mx(en) - e0) * 0.01, med(pr) * 0.01, mx(pr) * 0.01, med(lp) * 0.01, mx(lp) * 0.01, (x1 - med(ex)) * 0.01);
                // where the time is really lost: a SIMD is idle from the exit of ITS last wave to the exit of the launch's last wave (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID)
                std::map<unsigned long long, double> simd_last, cu_last, xcd_last;
                for (size_t w = 0; w < 8192; ++w) if (st[w * 8 + 1] > 0 && st[w * 8 + 2] > 0) {
                    const unsigned long long id = st[w * 8 + 5], ks = id & ~0xfull & 0xf0000ffffull, kc = ks & ~0x30ull, kx = id >> 32;
                    const double x = (double)st[w * 8 + 4];
                    simd_last[ks] = std::max(simd_last[ks], x); cu_last[kc] = std::max(cu_last[kc], x); xcd_last[kx] = std::max(xcd_last[kx], x);
                }
                double idle = 0; for (auto& kv : simd_last) idle += x1 - kv.second;
                double idc = 0; for (auto& kv : cu_last) idc += x1 - kv.second;
                printf("\n      %zu SIMDs on %zu CUs seen: a SIMD idles %.1f us on average behind its last wave (a CU %.1f us); last exit per XCD, us before the launch's:", simd_last.size(), cu_last.size(),
                       idle / simd_last.size() * 0.01, idc / cu_last.size() * 0.01);
                for (auto& kv : xcd_last) printf(" %.1f", (x1 - kv.second) * 0.01);
                {   // per XCD: median loop time (us) and median in-kernel clock (GHz) of its waves
                    std::map<unsigned long long, std::vector<double>> xl, xc;
                    for (size_t w = 0; w < 8192; ++w) if (st[w * 8 + 1] > 0 && st[w * 8 + 2] > 0) { xl[st[w * 8 + 5] >> 32].push_back((double)st[w * 8 + 1] * 0.01); xc[st[w * 8 + 5] >> 32].push_back((double)st[w * 8] / (double)st[w * 8 + 1] * 0.1); }
                    printf("\n      per XCD, median loop us @ GHz:");
                    for (auto& kv : xl) printf(" %.1f@%.2f", med(kv.second), med(xc[kv.first]));
                }
            }
        }
        std::fill(st.begin(), st.end(), 0ull);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), st.data(), st.size() * 8);
    }
#endif
    if (g_pass) printf("\n");
    if (hipGetLastError() != hipSuccess) printf("   !! HIP error\n");
    return us;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int B = 128, D = 25;
    const long V = 22 * 22 * 9, nvox = (long)B * V;
    std::vector<float> h((size_t)nvox * 32);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((s >> 11) & 0xffffff) / 16777216.f - 0.5f; }
    float *x32, *y32, *z32, *x25, *y25;
    hipMalloc(&x32, nvox * 32 * 4); hipMalloc(&y32, nvox * 32 * 4); hipMalloc(&z32, nvox * 32 * 4); hipMalloc(&x25, nvox * 25 * 4); hipMalloc(&y25, nvox * 25 * 4);
    hipMemcpy(x32, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(y32, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(z32, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(x25, h.data(), nvox * 25 * 4, hipMemcpyHostToDevice); hipMemcpy(y25, h.data(), nvox * 25 * 4, hipMemcpyHostToDevice);
    // amax slots: every slot = 1.0f (the data is in [-0.5, 0.5]; weights are random pieces)
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> one(8192, 0x3f800000u); hipMemcpy(am_, one.data(), 8192 * 4, hipMemcpyHostToDevice); }
    float *w, *b1, *b2, *dW1, *dW2, *db1, *db2, *slabs, *wf, *bias, *dw, *db, *part;
    hipMalloc(&w, 4 * X6_PW_FRAG_WORDS * 4); hipMalloc(&b1, 256 * 4); hipMalloc(&b2, 32 * 4);
    // weight fragments: RANDOM fp16 pieces in (-1, 1) unless KB_CONST_WEIGHTS (constant operands toggle fewer wires: the chip then holds a higher clock
    // and the numbers flatter the kernels -- round 2's tables were taken that way)
    auto fill_frag = [&](float* dst, size_t words) {
#ifdef KB_CONST_WEIGHTS
        hipMemset(dst, 0x3c, words * 4);
#else
        std::vector<unsigned> hw(words);
        for (auto& u : hw) {
            unsigned short hh[2];
            for (int q = 0; q < 2; ++q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const float f = (float)((s >> 11) & 0xffffff) / 8388608.f - 1.f; _Float16 hf = (_Float16)f; hh[q] = *reinterpret_cast<unsigned short*>(&hf); }
            u = hh[0] | ((unsigned)hh[1] << 16);
        }
        hipMemcpy(dst, hw.data(), words * 4, hipMemcpyHostToDevice);
#endif
    };
    fill_frag(w, 4 * X6_PW_FRAG_WORDS); hipMemset(b1, 0, 256 * 4); hipMemset(b2, 0, 32 * 4);
    hipMalloc(&dW1, 8192 * 4); hipMalloc(&dW2, 256 * D * 4); hipMalloc(&db1, 256 * 4); hipMalloc(&db2, D * 4);
    hipMalloc(&slabs, mfma_pw_backward_slab_floats(D) * 4);
    hipMalloc(&wf, X6_CONV_FRAG_WORDS * 4); fill_frag(wf, X6_CONV_FRAG_WORDS);
    hipMalloc(&bias, 32 * 4); hipMemset(bias, 0, 32 * 4);
    hipMalloc(&dw, 27 * 32 * 32 * 4); hipMalloc(&db, 32 * 4);
    ConvGeom gf{B, 22, 22, 9, 25, 22, 22, 9, 32, 3, 3, 3, 1, 1, 1, 0, 0, 0};      // normConv forward
    ConvGeom gb{B, 22, 22, 9, 32, 22, 22, 9, 25, 3, 3, 3, 1, 1, 1, 0, 0, 0};      // its backward-data
    hipMalloc(&part, x6_wgrad_partial_floats(gf) * 4);
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = am_ + 4096;
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.y = am_ + 4096;
#ifndef KB_OLD
    pam.w2c = am_ + 2100; pam.w1r = am_ + 2200;
#endif
    const double gv = (double)nvox * 2e-9;
    for (g_pass = 0; g_pass < 3; ++g_pass) {          // pass 0 warms the clocks up and is not printed
    if (g_pass) printf("-- pass %d\n", g_pass);
#ifdef KB_REDUCERS
    {   // the reducer stages of the 9-frame network (32 -> 32 channels): forward, backward-data (with and without the ReLU mask), backward-filter
        static float* wfr = nullptr; static float* partr = nullptr; static float *x32, *y32, *z32;          // (shadow the benchmark's buffers: the padded 24 x 24 x 9 tensor is larger)
        if (!wfr) {
            hipMalloc(&wfr, X6_CONV_FRAG_WORDS * 4); fill_frag(wfr, X6_CONV_FRAG_WORDS);
            const size_t nb = (size_t)B * 24 * 24 * 9 * 32;
            hipMalloc(&x32, nb * 4); hipMalloc(&y32, nb * 4); hipMalloc(&z32, nb * 4);
            std::vector<float> hb(nb);
            for (size_t i = 0; i < nb; ++i) hb[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
            hipMemcpy(x32, hb.data(), nb * 4, hipMemcpyHostToDevice); hipMemcpy(y32, hb.data(), nb * 4, hipMemcpyHostToDevice); hipMemcpy(z32, hb.data(), nb * 4, hipMemcpyHostToDevice);
        }
        const int hs[4] = {24, 22, 20, 18}, ts[4] = {9, 7, 5, 3};           // extents: padded input of reducer 1, then the outputs of reducers 1..3
        for (int k = 0; k < 3; ++k) {
            ConvGeom f{B, hs[k], hs[k], ts[k], 32, hs[k + 1], hs[k + 1], ts[k + 1], 32, 3, 3, 3, 0, 0, 0, 0, 1, 0};             // valid, ReLU
            ConvGeom bd{B, hs[k + 1], hs[k + 1], ts[k + 1], 32, hs[k], hs[k], ts[k], 32, 3, 3, 3, 2, 2, 2, 0, 0, 0};           // its backward-data ("full")
            if (!partr) hipMalloc(&partr, 2 * x6_wgrad_partial_floats(ConvGeom{B, 24, 24, 9, 32, 22, 22, 7, 32, 3, 3, 3, 0, 0, 0, 0, 1, 0}) * 4);   // (short rows: twice the slabs)
            char nm[96];
            const double gfl = (double)B * hs[k + 1] * hs[k + 1] * ts[k + 1] * 27 * 32 * 32 * 2e-9;
            snprintf(nm, sizeof(nm), "reducer %d forward  (%dx%dx%d out)", k + 1, hs[k + 1], hs[k + 1], ts[k + 1]);
            timeit(nm, iters, gfl, [&] { x6_conv_strip_forward(f, x32, nullptr, wfr, bias, nullptr, y32, 2, am, 0); });
            snprintf(nm, sizeof(nm), "reducer %d backward-data, gated", k + 1);
            timeit(nm, iters, gfl, [&] { x6_conv_strip_forward(bd, y32, z32, wfr, nullptr, nullptr, x32, 2, am, 0); });
            snprintf(nm, sizeof(nm), "reducer %d backward-data, no gate", k + 1);
            timeit(nm, iters, gfl, [&] { x6_conv_strip_forward(bd, y32, nullptr, wfr, nullptr, nullptr, x32, 2, am, 0); });
            snprintf(nm, sizeof(nm), "reducer %d backward-filter, gated", k + 1);
            timeit(nm, iters, gfl, [&] { x6_conv_wgrad(f, x32, y32, z32, dw, db, partr, 2, am, 0); });
        }
    }
    continue;
#endif
#ifdef KB_ONLY_STRIP
    timeit("pstrip<25> normConv forward + skip", iters, gv * 21600, [&] { x6_conv_strip_forward(gf, x25, nullptr, wf, bias, y32, z32, 2, am, 0); });
    timeit("pstrip<32> normConv backward-data", iters, gv * 21600, [&] { x6_conv_strip_forward(gb, x32, nullptr, wf, nullptr, nullptr, y25, 2, am, 0); });
    continue;
#endif
#ifdef KB_ONLY_WG
    timeit("wgrad<25> normConv backward-filter", iters, gv * 21600, [&] { x6_conv_wgrad(gf, x25, y32, nullptr, dw, db, part, 2, am, 0); });
    continue;
#endif
#ifdef KB_OLD
    timeit("pw_fwd  (expConv+ReLU+decConv)", iters, gv * 14592, [&] { x6_pw_forward(x32, w, w + X6_PW_FRAG_WORDS, b1, b2, y25, nvox, D, 2, pam, 0); });
    timeit("pw_bwd  (fused reverse)", iters, gv * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nvox, D, 2, pam, 0); });
#else
    timeit("pw_fwd  (expConv+ReLU+decConv)", iters, gv * 14592, [&] { x6_pw_forward(x32, w, w + X6_PW_FRAG_WORDS, b1, b2, y25, nvox, V, D, 2, pam, 0); });
    timeit("pw_bwd  (fused reverse)", iters, gv * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nvox, V, D, 2, pam, 0); });
#ifdef KB_ONLY_PW
    continue;                                        // (ablation builds of the fused pointwise kernels: -DH3S_... ; only the two lines above)
#endif
    { PwAmax q = pam; q.y = nullptr;
      timeit("pw_bwd  no amax report", iters, gv * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nvox, V, D, 2, q, 0); });
      timeit("pw_fwd  no amax report", iters, gv * 14592, [&] { x6_pw_forward(x32, w, w + X6_PW_FRAG_WORDS, b1, b2, y25, nvox, V, D, 2, q, 0); }); }
    { const long V2 = 4352, nv2 = B * V2;
      timeit("pw_bwd  vps = 4352 (full tiles only)", iters, (double)nv2 * 2e-9 * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nv2, V2, D, 2, pam, 0); });
      timeit("pw_bwd  vps = 2*4356", iters, gv * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nvox, 2 * V, D, 2, pam, 0); }); }
    timeit("pw_fwd  one sample (vps = nvox)", iters, gv * 14592, [&] { x6_pw_forward(x32, w, w + X6_PW_FRAG_WORDS, b1, b2, y25, nvox, nvox, D, 2, pam, 0); });
    timeit("pw_bwd  one sample (vps = nvox)", iters, gv * 29184, [&] { x6_pw_backward(x32, x25, y32, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, z32, dW1, dW2, db1, db2, slabs, nvox, nvox, D, 2, pam, 0); });
#endif
    timeit("pstrip<25> normConv forward + skip", iters, gv * 21600, [&] { x6_conv_strip_forward(gf, x25, nullptr, wf, bias, y32, z32, 2, am, 0); });
    timeit("pstrip<32> normConv backward-data", iters, gv * 21600, [&] { x6_conv_strip_forward(gb, x32, nullptr, wf, nullptr, nullptr, y25, 2, am, 0); });
    timeit("wgrad<25> normConv backward-filter", iters, gv * 21600, [&] { x6_conv_wgrad(gf, x25, y32, nullptr, dw, db, part, 2, am, 0); });
    }
    return 0;
}
