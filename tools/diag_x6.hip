// Diagnostic (not part of the product): per-wave phase shares of the x6 backward-filter kernel.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DPROBAV_STAMP -I proba-v_amd/csrc tools/diag_x6.hip -o /tmp/diag_x6 && /tmp/diag_x6
// Never quote this build's run time: the stamps serialise; read the SHARES.
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_mfma.hip"
#include "../proba-v_amd/csrc/kernels_x6.hip"
#include <vector>
#include <cstdio>
using namespace probav;

int main()
{
    const int B = 128, cin = 25, cout = 32;
    ConvGeom g{B, 22, 22, 9, cin, 22, 22, 9, cout, 3, 3, 3, 1, 1, 1, 0, 0, 0};
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> hv(8192, 0x3f800000u); hipMemcpy(am_, hv.data(), 8192 * 4, hipMemcpyHostToDevice); }      // per-sample / per-column slots: all 1.0
    Amax am; am.x = am_; am.w = am_ + 2048; am.y = am_ + 4096;
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.w2c = am_ + 2100; pam.w1r = am_ + 2200; pam.y = am_ + 4096;
    const int ARITH = 2;
    const size_t nin = (size_t)B * 22 * 22 * 9 * cin, nout = (size_t)B * 22 * 22 * 9 * cout;
    float *x, *dy, *dw, *db, *part;
    hipMalloc(&x, nin * 4); hipMalloc(&dy, nout * 4); hipMalloc(&dw, 27 * cin * cout * 4); hipMalloc(&db, cout * 4);
    hipMalloc(&part, x6_wgrad_partial_floats(g) * 4);
    std::vector<float> h(nout);
    for (size_t i = 0; i < nout; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, h.data(), nout * 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) x6_conv_wgrad(g, x, dy, nullptr, dw, db, part, ARITH, am, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(8192 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* names[8] = {"t0", "prologue", "wait: rows staged", "k-loop", "wait: end of tile", "stage store", "epilogue", "t_end"};
    for (int wave = 0; wave < 8; ++wave) {
        double acc[8] = {0}; double life = 0;
        for (int b = 0; b < 256; ++b) {
            const unsigned long long* s = &st[(b * 8 + wave) * 8];
            for (int k = 1; k < 7; ++k) acc[k] += (double)s[k];
            life += (double)(s[7] - s[0]);
        }
        printf("wave %d (tg %d ksel %d): life %.0f cyc/WG |", wave, wave & 3, wave >> 2, life / 256);
        for (int k = 1; k < 7; ++k) printf(" %s %.1f%%", names[k], 100.0 * acc[k] / life);
        printf("\n");
    }
    for (int dir = 0; dir < 2; ++dir) {   // strip convolution: forward 25 -> 32, then its backward-data 32 -> 25 (alternating-halves form: 2 = taps, 3 = finishing, 4 = barrier)
        float *y, *wf, *bias;
        hipMalloc(&y, nout * 4); hipMalloc(&wf, X6_CONV_FRAG_WORDS * 4); hipMalloc(&bias, 32 * 4);
        { std::vector<unsigned> hw(X6_CONV_FRAG_WORDS); unsigned long long q = 88172645463325252ull;
          for (auto& u : hw) { unsigned short hh[2]; for (int e = 0; e < 2; ++e) { q ^= q << 13; q ^= q >> 7; q ^= q << 17; _Float16 hf = (_Float16)((float)((q >> 11) & 0xffffff) / 8388608.f - 1.f); hh[e] = *reinterpret_cast<unsigned short*>(&hf); } u = hh[0] | ((unsigned)hh[1] << 16); }
          hipMemcpy(wf, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); }
        hipMemset(bias, 0, 32 * 4);
        ConvGeom gb{B, 22, 22, 9, 32, 22, 22, 9, 25, 3, 3, 3, 1, 1, 1, 0, 0, 0};
        for (int it = 0; it < 3; ++it) {
            if (dir == 0) x6_conv_strip_forward(g, x, nullptr, wf, bias, dy, y, ARITH, am, 0);
            else x6_conv_strip_forward(gb, dy, nullptr, wf, nullptr, nullptr, y, ARITH, am, 0);
        }
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        const char* nm[8] = {"t0", "prologue", "taps (+skip loads)", "finishing: epilogue + stage store", "barrier waits", "epilogue", "-", "t_end"};
        for (int wave = 0; wave < 8; wave += 2) {
            double acc[8] = {0}; double life = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned long long* s = &st[(b * 8 + wave) * 8];
                for (int k = 1; k < 7; ++k) acc[k] += (double)s[k];
                life += (double)(s[7] - s[0]);
            }
            printf("strip %s wave %d: life %.0f cyc/WG |", dir ? "32->25 (16x16x32)" : "25->32", wave, life / 256);
            for (int k = 1; k < 6; ++k) printf(" %s %.1f%% (%.0f cyc)", nm[k], 100.0 * acc[k] / life, acc[k] / 256);
            printf("\n");
        }
    }
    {   // fused pointwise forward (stamps: 1 cut of X, 2 W1 reads + (a), 3 bias / ReLU / cut, 4 W2 reads + decay product, 5 epilogue + stores)
        const long nvox = (long)B * 22 * 22 * 9;
        const int D = 25;
        float *xx, *dec, *w, *b1, *b2;
        hipMalloc(&xx, nvox * 32 * 4); hipMalloc(&dec, nvox * D * 4); hipMalloc(&w, 3 * X6_PW_FRAG_WORDS * 4); hipMalloc(&b1, 256 * 4); hipMalloc(&b2, 32 * 4);
        hipMemcpy(xx, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
        hipMemset(w, 0x3c, 3 * X6_PW_FRAG_WORDS * 4); hipMemset(b1, 0, 256 * 4); hipMemset(b2, 0, 32 * 4);
        for (int it = 0; it < 3; ++it) x6_pw_forward(xx, w, w + X6_PW_FRAG_WORDS, b1, b2, dec, nvox, 22 * 22 * 9, D, ARITH, pam, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        const char* nm[8] = {"t0", "cut of X", "W1 reads + (a)", "bias / ReLU / cut", "W2 reads + decay product", "epilogue + stores (+ loop)", "-", "t_end"};
        for (int wave = 0; wave < 8; wave += 3) {
            double acc[8] = {0}; double life = 0;
            for (int b = 0; b < 512; ++b) {
                const unsigned long long* s = &st[(b * 8 + wave) * 8];
                for (int k = 1; k < 7; ++k) acc[k] += (double)s[k];
                life += (double)(s[7] - s[0]);
            }
            printf("pw_fwd wave %d: life %.0f cyc/WG, in-kernel clock %.2f GHz |", wave, life / 512, acc[6] > 0 ? life / acc[6] * 0.1 : 0.0);
            for (int k = 1; k < 6; ++k) printf(" %s %.1f%%", nm[k], 100.0 * acc[k] / life);
            printf("\n");
        }
    }
    {   // fused pointwise backward
        const long nvox = (long)B * 22 * 22 * 9;
        const int D = 25;
        float *xx, *dT, *dO, *dX, *w, *b1, *dW1, *dW2, *db1, *db2, *slabs;
        hipMalloc(&xx, nvox * 32 * 4); hipMalloc(&dT, nvox * D * 4); hipMalloc(&dO, nvox * 32 * 4); hipMalloc(&dX, nvox * 32 * 4);
        hipMalloc(&w, 3 * X6_PW_FRAG_WORDS * 4); hipMalloc(&b1, 256 * 4);
        hipMalloc(&dW1, 8192 * 4); hipMalloc(&dW2, 256 * D * 4); hipMalloc(&db1, 256 * 4); hipMalloc(&db2, D * 4);
        hipMalloc(&slabs, mfma_pw_backward_slab_floats(D) * 4);
        hipMemcpy(xx, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dT, h.data(), nvox * D * 4, hipMemcpyHostToDevice);
        hipMemcpy(dO, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
        hipMemset(w, 0x3c, 3 * X6_PW_FRAG_WORDS * 4); hipMemset(b1, 0, 256 * 4);
        for (int it = 0; it < 3; ++it)
            x6_pw_backward(xx, dT, dO, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, dX, dW1, dW2, db1, db2, slabs, nvox, 22 * 22 * 9, D, ARITH, pam, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
#ifdef PROBAV_STAMP_H3S
        const char* nm[8] = {"t0", "wait: barrier before Y", "Y: (d),(e),(c) of tile i-1, (a),(b) of tile i", "wait: barrier before X", "X: dX sums (half B)", "X: gate + cut + transpose stores", "X: stage store (half A)", "t_end"};
#elif defined(PROBAV_STAMP2)
        const char* nm[8] = {"t0", "(a),(b) + dX reduce of the previous tile", "gate + cut", "(c)", "Tb store, dH' transpose, (d)", "H' transpose, (e)", "-", "t_end"};
#else
        const char* nm[8] = {"t0", "wait: tile staged", "compute", "stage store", "wait: partials", "dX reduce", "epilogue", "t_end"};
#endif
        for (int wave = 0; wave < 8; wave += 3) {
            double acc[8] = {0}; double life = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned long long* s = &st[(b * 8 + wave) * 8];
                for (int k = 1; k < 7; ++k) acc[k] += (double)s[k];
                life += (double)(s[7] - s[0]);
            }
            printf("pw_bwd wave %d: life %.0f cyc/WG (%.0f per tile), in-kernel clock %.2f GHz |", wave, life / 256, life / 256 / 68.06, acc[6] > 0 ? life / acc[6] * 0.1 : 0.0);
            for (int k = 1; k < 7; ++k) printf(" %s %.1f%%", nm[k], 100.0 * acc[k] / life);
            printf("\n");
        }
    }
    return 0;
}
