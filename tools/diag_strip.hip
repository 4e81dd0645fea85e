// Diagnostic (not part of the product): time of the strip convolution under compile-time ablations (-DABL_NOMFMA / NOW / NOCUT / NOLDS).
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_mfma.hip"
#include <vector>
#include <cstdio>
using namespace probav;
int main()
{
    const int B = 128;
    for (int cin : {25, 32}) {
        const int cout = cin == 25 ? 32 : 25;
        ConvGeom g{B, 22, 22, 9, cin, 22, 22, 9, cout, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        const size_t nin = (size_t)B * 22 * 22 * 9 * cin, nout = (size_t)B * 22 * 22 * 9 * cout;
        float *x, *y, *wf, *bias, *skp; unsigned* am;
        hipMalloc(&x, nin * 4); hipMalloc(&y, nout * 4); hipMalloc(&skp, nout * 4); hipMemset(skp, 0, nout * 4); hipMalloc(&wf, X6_CONV_FRAG_WORDS * 4); hipMalloc(&bias, 32 * 4); hipMalloc(&am, 4096 * 4);
        std::vector<float> h(nin);
        for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
        hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice);
        hipMemset(wf, 0x3c, X6_CONV_FRAG_WORDS * 4); hipMemset(bias, 0, 32 * 4);
        { std::vector<unsigned> hv(4096, 0x3f800000u); hipMemcpy(am, hv.data(), 4096 * 4, hipMemcpyHostToDevice); }     // per-sample / per-column slots: all 1.0
        for (int arith = 1; arith <= 2; ++arith) {
            Amax m; m.x = am; m.w = am + 1024; m.y = am + 2048;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int it = 0; it < 3; ++it) x6_conv_strip_forward(g, x, nullptr, wf, bias, cin == 25 ? skp : nullptr, y, arith, m, 0);
            hipEventRecord(e0, 0);
            for (int it = 0; it < 20; ++it) x6_conv_strip_forward(g, x, nullptr, wf, bias, cin == 25 ? skp : nullptr, y, arith, m, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("strip cin %d arith %d: %.1f us\n", cin, arith, ms * 1000 / 20);
#ifdef PROBAV_STAMP
            {
                std::vector<unsigned long long> st(8192 * 8);
                hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
                const char* nm[8] = {"t0", "prologue", "taps", "stage store", "barrier waits", "epilogue", "-", "t_end"};
                for (int wave = 0; wave < 8; wave += 1) {
                    double acc[8] = {0}; double life = 0;
                    for (int b = 0; b < 256; ++b) {
                        const unsigned long long* q = &st[(b * 8 + wave) * 8];
                        for (int k = 1; k < 7; ++k) acc[k] += (double)q[k];
                        life += (double)(q[7] - q[0]);
                    }
                    printf("  wave %d: life %.0f cyc/WG |", wave, life / 256);
                    for (int k = 1; k < 6; ++k) printf(" %s %.1f%% (%.0f)", nm[k], 100.0 * acc[k] / life, acc[k] / 256);
                    printf("\n");
                }
            }
#endif
        }
    }
    return 0;
}
