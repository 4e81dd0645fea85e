// Diagnostic (not part of the product): per-wave phase times of the fused pointwise backward (pw_bwd_h3t_kernel / pw_bwd_h3s_kernel with PROBAV_PW_BWD_H3S=1).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -w -DPROBAV_STAMP -I proba-v_amd/csrc -I include tools/diag_h3t.hip -o tools/diag_h3t.bin && tools/diag_h3t.bin
// Stamps (s_memtime) at the phase boundaries of every iteration: cycles per tile iteration of each of the eight waves, averaged over the workgroups:
// wait at the barrier before Y | Y (30 MFMAs + the work in their gaps) | wait at the barrier behind Y | X (loads, gate and cut, transpose stores).
// The stamps serialise a little (each waits for lgkmcnt(0)): read the SHARES and the differences between waves, not the total.
#include "../proba-v_amd/csrc/kernels_small.hip"
#include "../proba-v_amd/csrc/kernels_mfma.hip"
#include "../proba-v_amd/csrc/kernels_x6.hip"
#include <vector>
#include <cstdio>
using namespace probav;
int main()
{
    const int B = 128, D = 25;
    const long V = 22 * 22 * 9, nvox = (long)B * V;
    std::vector<float> h((size_t)nvox * 32);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((s >> 11) & 0xffffff) / 16777216.f - 0.5f; }
    float *xx, *dT, *dO, *dX, *w, *b1, *dW1, *dW2, *db1, *db2, *slabs;
    hipMalloc(&xx, nvox * 32 * 4); hipMalloc(&dT, nvox * D * 4); hipMalloc(&dO, nvox * 32 * 4); hipMalloc(&dX, nvox * 32 * 4);
    hipMalloc(&w, 3 * X6_PW_FRAG_WORDS * 4); hipMalloc(&b1, 256 * 4);
    hipMalloc(&dW1, 8192 * 4); hipMalloc(&dW2, 256 * D * 4); hipMalloc(&db1, 256 * 4); hipMalloc(&db2, D * 4);
    hipMalloc(&slabs, mfma_pw_backward_slab_floats(D) * 4);
    hipMemcpy(xx, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dT, h.data(), nvox * D * 4, hipMemcpyHostToDevice);
    hipMemcpy(dO, h.data(), nvox * 32 * 4, hipMemcpyHostToDevice);
    { std::vector<unsigned> hw(3 * X6_PW_FRAG_WORDS);
      for (auto& u : hw) { unsigned short hh[2]; for (int q = 0; q < 2; ++q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; _Float16 hf = (_Float16)((float)((s >> 11) & 0xffffff) / 8388608.f - 1.f); hh[q] = *reinterpret_cast<unsigned short*>(&hf); } u = hh[0] | ((unsigned)hh[1] << 16); }
      hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); }
    hipMemset(b1, 0, 256 * 4);
    unsigned* am_; hipMalloc(&am_, 8192 * 4);
    { std::vector<unsigned> one(8192, 0x3f800000u); hipMemcpy(am_, one.data(), 8192 * 4, hipMemcpyHostToDevice); }
    PwAmax pam; pam.x = am_; pam.w1 = am_ + 2048; pam.w2 = am_ + 2049; pam.b1 = am_ + 2050; pam.dt = am_ + 1024; pam.w2c = am_ + 2100; pam.w1r = am_ + 2200; pam.y = am_ + 4096;
    for (int it = 0; it < 4; ++it)
        x6_pw_backward(xx, dT, dO, w, w + X6_PW_FRAG_WORDS, w + 2 * X6_PW_FRAG_WORDS, b1, dX, dW1, dW2, db1, db2, slabs, nvox, V, D, 2, pam, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(8192 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const double iters = 70.5;                                   // nt + 2, nt = 68.5 on average
    printf("wave | life cycles | clock GHz | per iteration: wait before Y | Y | wait behind Y | X | sum\n");
    for (int wave = 0; wave < 8; ++wave) {
        double acc[8] = {0}, life = 0, mn[8], mx[8];
        for (int k = 0; k < 8; ++k) { mn[k] = 1e30; mx[k] = 0; }
        for (int b = 0; b < 256; ++b) {
            const unsigned long long* q = &st[(b * 8 + wave) * 8];
            for (int k = 1; k < 7; ++k) { acc[k] += (double)q[k]; if ((double)q[k] < mn[k]) mn[k] = (double)q[k]; if ((double)q[k] > mx[k]) mx[k] = (double)q[k]; }
            life += (double)(q[7] - q[0]);
        }
        for (int k = 1; k < 7; ++k) acc[k] /= 256;
        life /= 256;
        printf("%d (%s) | %8.0f | %.2f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f      (Y min %.0f max %.0f; X min %.0f max %.0f over the workgroups)\n", wave, wave < 4 ? "A" : "B", life, acc[6] > 0 ? life / acc[6] * 0.1 : 0.0,
               acc[1] / iters, acc[2] / iters, acc[3] / iters, acc[5] / iters, (acc[1] + acc[2] + acc[3] + acc[5]) / iters, mn[2] / iters, mx[2] / iters, mn[5] / iters, mx[5] / iters);
    }
#ifdef PROBAV_STAMP_Y
    printf("inside Y (extra stamps, each a full lgkmcnt(0) wait: the phase is longer than in the build without them), cycles per iteration:\nwave | prologue + (a) | (d),(e) k-block 0 | (d),(e) k-block 1 | (c) | (b)      [-DH3T_PURE: bias+ReLU+cut+store | staging / sums | loads | - | -]\n");
    for (int wave = 0; wave < 8; ++wave) {
        double acc[5] = {0};
        for (int b = 0; b < 256; ++b) for (int q = 0; q < 5; ++q) acc[q] += (double)st[4096 * 8 + (b * 8 + wave) * 8 + q];
        printf("%d (%s) | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f\n", wave, wave < 4 ? "A" : "B", acc[0] / 256 / iters, acc[1] / 256 / iters, acc[2] / 256 / iters, acc[3] / 256 / iters, acc[4] / 256 / iters);
    }
#endif
    return 0;
}
