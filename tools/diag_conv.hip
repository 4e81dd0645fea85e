// Diagnostic (not part of the product): where does a workgroup of conv3_mfma_kernel spend its cycles?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DPROBAV_STAMP -I proba-v_amd/csrc tools/diag_conv.hip -o /tmp/diag_conv && /tmp/diag_conv
// Rebuilds the kernel file with in-kernel s_memtime stamps (fill done / taps done / end) and prints their shares.
// Never quote this build's run time: the stamps serialise; read the SHARES.
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_mfma.hip"
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
using namespace probav;

int main(int argc, char** argv)
{
    const int cin = argc > 1 ? atoi(argv[1]) : 25, cout = argc > 2 ? atoi(argv[2]) : 32, B = 128;
    ConvGeom g{B, 22, 22, 9, cin, 22, 22, 9, cout, 3, 3, 3, 1, 1, 1, 0, 0};
    const size_t nin = (size_t)B * 22 * 22 * 9 * cin, nout = (size_t)B * 22 * 22 * 9 * cout, nfrag = mfma_conv_wfrag_floats(cin, cout);
    float *x, *y, *w;
    hipMalloc(&x, nin * 4); hipMalloc(&y, nout * 4); hipMalloc(&w, nfrag * 4);
    std::vector<float> h(nin);
    for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(nfrag);
    for (size_t i = 0; i < nfrag; ++i) hw[i] = (float)((i * 40503u) % 1000) / 5000.f - 0.1f;
    hipMemcpy(w, hw.data(), nfrag * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) mfma_conv_forward(g, x, nullptr, w, nullptr, nullptr, y, Amax(), 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int it = 0; it < 5; ++it) mfma_conv_forward(g, x, nullptr, w, nullptr, nullptr, y, Amax(), 0);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(8192 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const int nwg = B * 22;
    const int nchunk = cin == 25 || cin == 1 ? 1 : cin / 16;
    double fill = 0, taps = 0, epi = 0, life = 0; int n = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nwg && b < 8192; ++b) {
        const unsigned long long* s = &st[b * 8];
        double f = 0, t = 0; unsigned long long prev = s[0];
        for (int c = 0; c < nchunk; ++c) { f += (double)(s[1 + 2 * c] - prev); t += (double)(s[2 + 2 * c] - s[1 + 2 * c]); prev = s[2 + 2 * c]; }
        fill += f; taps += t; epi += (double)(s[7] - prev); life += (double)(s[7] - s[0]); ++n;
        tmin = std::min(tmin, s[0]); tmax = std::max(tmax, s[7]);
    }
    {   // backward-filter of the same layer
        float *dy, *dw, *db, *part;
        hipMalloc(&dy, nout * 4); hipMemcpy(dy, y, nout * 4, hipMemcpyDeviceToDevice);
        hipMalloc(&dw, 27 * cin * cout * 4); hipMalloc(&db, cout * 4);
        if (mfma_wgrad_supported(g)) {
            hipMalloc(&part, mfma_wgrad_partial_floats(g) * 4);
            for (int it = 0; it < 3; ++it) mfma_conv_wgrad(g, x, dy, nullptr, dw, db, part, 0);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int it = 0; it < 5; ++it) mfma_conv_wgrad(g, x, dy, nullptr, dw, db, part, 0);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms2; hipEventElapsedTime(&ms2, e0, e1);
            std::vector<unsigned long long> s2(8192 * 8);
            hipMemcpyFromSymbol(s2.data(), HIP_SYMBOL(g_stamps), s2.size() * 8);
            double f = 0, st = 0, life2 = 0; int m = 0;
            for (int b = 0; b < 512; ++b) { f += (double)s2[b * 8 + 1]; st += (double)s2[b * 8 + 2]; life2 += (double)(s2[b * 8 + 7] - s2[b * 8]); ++m; }
            printf("wgrad Cin %d Cout %d: %.1f us/launch incl. slab reduce (stamped); per workgroup (cycles): fill %.0f  steps %.0f  lifetime %.0f\n", cin, cout, ms2 / 5 * 1e3, f / m, st / m, life2 / m);
        }
    }
    printf("Cin %d Cout %d: %.1f us/launch (stamped build); per workgroup (shader cycles): fill %.0f  taps %.0f  epilogue %.0f  lifetime %.0f ; kernel span %.0f ticks, %d WGs\n",
           cin, cout, ms / 5 * 1e3, fill / n, taps / n, epi / n, life / n, (double)(tmax - tmin), n);
    return 0;
}
