// Engine: sequences the HIP kernels into the WDSR-B Conv3D network of models/modelsTF.py:15-203
// (forward) and its reverse-mode gradient (what tf.GradientTape computes at models/trainClass.py:126-131),
// and exports the C ABI of include/probav_hip.h.  Host-side C++ only decides shapes, offsets and launch
// order; it never touches tensor data and never synchronises.
#include "probav_common.h"
#include "../../include/probav_hip.h"
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

using namespace probav;

#include "kernels_mfma.h"
#include "kernels_x6.h"

struct LayerRec {
    char name[32];
    WnLayer wn;
    int kh, kw, kt;
};

struct probav_engine {
    probav_net_cfg cfg;
    std::vector<LayerRec> layers;
    int64_t nparams = 0, weff_count = 0, cout_total = 0, cin_total = 0;
    WnLayer* d_layers = nullptr;
    int impl = 4;             // 0 direct kernels, 1 MFMA row-tile kernels, 2 MFMA + strip convolution where it applies,
                              // 3 = 2 with the x6 kernels (fp32 products as six bf16-piece MFMA products) where they exist,
                              // 4 = 3 with the H3 arithmetic (three products of scaled fp16 piece pairs) where it exists
    int iMain = -1, iResid1 = -1, iResid2 = -1, iResid3 = -1, iUp = -1;
    std::vector<int> iExp, iDec, iNorm, iRed;
    ReduceSide side = {};                           // side stream of the slab sums (probav_common.h), created on first use
    int side_mode = 2;                              // probav_engine_side_stream(): 0 off, 1 small work only, 2 + the backward-filter kernels
    bool side_tried = false;
    struct RedSpec { int k, p, pt, refl, refl_t; };   // one valid convReducer: kernel size, H/W pad, depth pad, mirrored H/W pad, mirrored depth pad
    std::vector<RedSpec> redSpec;
    int Hin = 0;
    // MFMA operand fragments: one packing job per (layer, use); offsets into the workspace's wpack region
    std::vector<PackJob> jobs;
    PackJob* d_jobs = nullptr;
    int64_t wpack_count = 0;
    std::vector<long> pkFwd, pkBwd;          // per layer: conv fragments for forward / backward-data (-1 = none)
    std::vector<long> pkFwd6, pkBwd6;        // per layer: x6 (pre-split bf16) conv fragments, impl 3
    std::vector<long> pkFwdH, pkBwdH;        // per layer: H3 (scaled fp16 pieces) conv fragments, impl 4
    std::vector<long> pkFwdHt, pkBwdHt;      // per layer, 25 input channels only: the per-tap (not K-concatenated) H3 fragments of the piece-ring strip kernel
    std::vector<long> pkW1h, pkW2h, pkW2Kh, pkW1Ch;   // per block: H3 fragments of the fused expand/decay forward / backward
    std::vector<long> pkW1x6, pkW2x6;        // per block: x6 fragments of the fused expand/decay forward
    std::vector<long> pkW2Kx6, pkW1Cx6;      // per block: extra x6 fragments of the fused backward
    std::vector<long> pkW1, pkW2;            // per block: fused expand/decay forward fragments
    std::vector<long> pkW2B, pkW1C;          // per block: extra fragments of the fused backward
    bool pw_mfma = false;
    bool fwd_amax = false;    // the last training forward filled the amax slots of the saved activations (H3 kernels, impl 4)
    bool fwd_unfused = false; // ... and laid its workspace out for the unfused pointwise pair (impl 0): the backward pass must agree
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg)
    bool prof_on = false;
    unsigned prof_mask = ~0u;   // kernel classes whose launches are bracketed (bit = class index)
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_cls;
    std::vector<double> prof_macs;
    size_t prof_used = 0;
};

enum { CLS_WN = 0, CLS_SMALL, CLS_CONV3_FWD, CLS_CONV3_BWD_DATA, CLS_CONV3_WGRAD, CLS_PW_FWD, CLS_PW_BWD_DATA, CLS_PW_WGRAD,
       // launches served by an x6 kernel (bf16 MFMA pipe) are timed apart from the fp32-MFMA / VALU ones: they price against another peak
       CLS_CONV3_FWD_X6, CLS_CONV3_BWD_DATA_X6, CLS_CONV3_WGRAD_X6, CLS_PW_FWD_X6, CLS_PW_BWD_DATA_X6, CLS_COUNT };

struct ProfScope {
    probav_engine* e; hipStream_t s; bool live;
    ProfScope(const probav_engine* ce, int cls, double macs, hipStream_t st) : e(const_cast<probav_engine*>(ce)), s(st), live(false)
    {
        if (!e->prof_on || !((e->prof_mask >> cls) & 1u) || e->prof_used + 2 > e->prof_ev.size()) return;
        if (e->side.side && st == e->side.side) return;    // launches on the side stream run in the gaps of the caller's chain: events around them would time the chain, not them
        live = true;
        e->prof_cls.push_back(cls);
        e->prof_macs.push_back(macs);
        (void)hipEventRecord(e->prof_ev[e->prof_used], s);
    }
    ~ProfScope()
    {
        if (!live) return;
        (void)hipEventRecord(e->prof_ev[e->prof_used + 1], s);
        e->prof_used += 2;
    }
};
static double geom_macs(const ConvGeom& g)
{
    return (double)g.N * g.Ho * g.Wo * g.To * g.kh * g.kw * g.kt * g.Cin * g.Cout;
}

static int add_layer(probav_engine* e, const std::string& name, int kh, int kw, int kt, int cin, int cout)
{
    LayerRec r;
    memset(&r, 0, sizeof(r));
    snprintf(r.name, sizeof(r.name), "%s", name.c_str());
    r.kh = kh; r.kw = kw; r.kt = kt;
    r.wn.taps = kh * kw * kt; r.wn.Cin = cin; r.wn.Cout = cout; r.wn.K = r.wn.taps * cin;
    r.wn.g_off = (int)e->nparams;
    r.wn.v_off = r.wn.g_off + cout;
    r.wn.b_off = r.wn.v_off + r.wn.K * cout;
    r.wn.w_off = (int)e->weff_count;
    r.wn.n_off = (int)e->cout_total;
    r.wn.r_off = (int)e->cin_total;
    e->nparams += 2 * cout + (int64_t)r.wn.K * cout;
    e->weff_count += (int64_t)r.wn.K * cout;
    e->cout_total += cout;
    e->cin_total += cin;
    e->layers.push_back(r);
    return (int)e->layers.size() - 1;
}

static size_t align_up(size_t v) { return (v + 63) & ~(size_t)63; }      // 64 floats = 256 B

struct Plan {
    size_t weff, weffT, invn, xn, mn;
    size_t amax; int n_amax, amax_bwd, amax_fwd, B;   // amax slots (one 32-bit word each, x6_device.h): region offset, count, first slot of the backward / forward per-sample arrays
    std::vector<size_t> act, dec, red;
    std::vector<int> redH, redT;              // output extent of each reducer
    size_t up, r1, r2, r3, H, wpack;
    // --- what only the reverse pass writes (offsets relative to the SCRATCH base: the tail of a one-piece workspace, or the caller's second buffer) ---
    size_t bamax, dweff2, Hb, dH, gA, gB, gDec, dtail, dr2, dr1, partial;
    size_t fwd_total, bwd_total, total;       // floats: the saved state of a forward pass | the reverse pass's scratch | both
    std::vector<size_t> gblk, gred;
    std::vector<size_t> part_off;      // slab region of the k-th backward-filter launch of a backward pass (relative to `partial`), in launch order
};

static ConvGeom make_geom(int N, int Hi, int Ti, int Cin, int Ho, int To, int Cout, int kh, int kw, int kt,
                          int ph, int pt, int reflect, int relu)
{
    ConvGeom g;
    g.N = N; g.Hi = Hi; g.Wi = Hi; g.Ti = Ti; g.Cin = Cin; g.Ho = Ho; g.Wo = Ho; g.To = To; g.Cout = Cout;
    g.kh = kh; g.kw = kw; g.kt = kt; g.ph = ph; g.pw = ph; g.pt = pt; g.reflect_hw = reflect; g.relu = relu; g.reflect_t = 0;
    return g;
}

// backward-data geometry of a forward layer: a correlation of dy with the flipped, channel-swapped
// kernel and padding k-1-p; for reflect layers the result is the gradient of the PADDED input.
static ConvGeom bwd_data_geom(const ConvGeom& f)
{
    ConvGeom b = f;
    const int php = f.reflect_hw ? 0 : f.ph, pwp = f.reflect_hw ? 0 : f.pw, ptp = f.reflect_t ? 0 : f.pt;
    b.Hi = f.Ho; b.Wi = f.Wo; b.Ti = f.To; b.Cin = f.Cout;
    b.Ho = f.reflect_hw ? f.Hi + 2 * f.ph : f.Hi;
    b.Wo = f.reflect_hw ? f.Wi + 2 * f.pw : f.Wi;
    b.To = f.reflect_t ? f.Ti + 2 * f.pt : f.Ti;
    b.Cout = f.Cin;
    b.ph = f.kh - 1 - php; b.pw = f.kw - 1 - pwp; b.pt = f.kt - 1 - ptp;
    b.reflect_hw = 0; b.reflect_t = 0; b.relu = 0;
    return b;
}

// geometry of reducer k on an input of extent h x h x t
static ConvGeom red_geom(const probav_engine* e, int B, size_t k, int h, int t, int F)
{
    const probav_engine::RedSpec& r = e->redSpec[k];
    ConvGeom g = make_geom(B, h, t, F, h + 2 * r.p - (r.k - 1), t + 2 * r.pt - (r.k - 1), F, r.k, r.k, r.k, r.p, r.pt, r.refl, 1);
    g.reflect_t = r.refl_t;
    return g;
}

static void reducer_extents(const probav_engine* e, std::vector<int>& hh, std::vector<int>& tt)
{
    int h = e->Hin, t = e->cfg.num_img_lr;
    for (size_t k = 0; k < e->iRed.size(); ++k) {
        const probav_engine::RedSpec& r = e->redSpec[k];
        h += 2 * r.p - (r.k - 1);
        t += 2 * r.pt - (r.k - 1);
        hh.push_back(h); tt.push_back(t);
    }
}

// slab floats of the backward-filter launch conv_wgrad() makes for geometry g under the engine's kernel family (the same decisions)
// the low-frequency residual path runs as one launch each way (kernels_direct.hip: resid_path_*) in every family but 0, which stays on the generic direct kernels
static bool resid_path_fused(const probav_engine* e)
{
    return e->impl >= 1 && resid_path_supported(e->Hin, e->cfg.in_channels, e->cfg.scale * e->cfg.scale);
}
static size_t wgrad_need(const probav_engine* e, const ConvGeom& g)
{
    const bool exotic = g.reflect_t || g.ph > 1 || g.pw > 1 || g.pt > 1 || (g.kh != 3 && g.kh != 1);
    if (exotic || (e->impl >= 1 && conv3d_direct_wgrad_is_tuned(g))) return wgrad_partial_floats(g);
    if (e->impl >= 3 && x6_wgrad_supported(g)) return x6_wgrad_partial_floats(g);
    if (e->impl >= 1 && mfma_wgrad_supported(g)) return mfma_wgrad_partial_floats(g);
    return wgrad_partial_floats(g);
}

static Plan make_plan(const probav_engine* e, int B, int training)
{
    Plan p;
    const probav_net_cfg& c = e->cfg;
    const int F = c.num_filters, E = F * c.exp_rate, D = c.dec_channels, T = c.num_img_lr, R = c.num_res_blocks;
    const int Hin = e->Hin, P = c.patch_size_lr, s2 = c.scale * c.scale;
    const size_t V = (size_t)B * Hin * Hin * T;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n); return o; };
    p.weff = take(e->weff_count); p.weffT = take(e->weff_count); p.invn = take(e->cout_total);
    {   // slots: per layer [weights | biases] (whole tensor), per output channel, per input channel (wn_forward's layout); then ONE SLOT PER
        // SAMPLE for act[0..R], dec[0..R-1], the reducer outputs, and for the backward pass's tensors in launch order
        const int L = (int)e->layers.size(), nred = (int)e->iRed.size();
        p.B = B;
        p.amax_fwd = 2 * L + (int)e->cout_total + (int)e->cin_total;
        p.amax_bwd = p.amax_fwd + B * (2 * R + 1 + nred);
        p.n_amax = p.amax_bwd + (training ? B * (2 * R + 2 * nred + 8) : 0);
        p.amax = take((size_t)p.amax_bwd);
    }
    p.wpack = take(e->wpack_count);
    p.xn = take(V * c.in_channels); p.mn = take((size_t)B * Hin * Hin * c.in_channels);
    if (training) {
        for (int i = 0; i <= R; ++i) p.act.push_back(take(V * F));
        for (int i = 0; i < R; ++i) p.dec.push_back(take(V * D));
    } else {
        const size_t a0 = take(V * F), a1 = take(V * F), d0 = take(V * D);
        for (int i = 0; i <= R; ++i) p.act.push_back((i & 1) ? a1 : a0);
        for (int i = 0; i < R; ++i) p.dec.push_back(d0);
    }
    reducer_extents(e, p.redH, p.redT);
    for (size_t k = 0; k < e->iRed.size(); ++k) p.red.push_back(take((size_t)B * p.redH[k] * p.redH[k] * p.redT[k] * F));
    p.up = take((size_t)B * P * P * s2);
    p.r1 = take((size_t)B * (Hin - 2) * (Hin - 2) * s2);
    p.r2 = take((size_t)B * (Hin - 4) * (Hin - 4) * s2);
    p.r3 = take((size_t)B * P * P * s2);
    // the 256-channel hidden tensor (1 KB per voxel) only exists in memory when the pointwise pair runs UNfused (generic kernels, impl 0)
    const bool unfused = !(e->impl >= 1 && e->pw_mfma);
    p.H = take(unfused ? V * E : 0);
    p.fwd_total = off;
    off = 0;
    p.bamax = p.dweff2 = p.Hb = p.dH = p.gA = p.gB = p.gDec = p.dtail = p.dr2 = p.dr1 = p.partial = 0;
    if (training) {
        p.bamax = take((size_t)(p.n_amax - p.amax_bwd));
        p.dweff2 = take(e->weff_count);
        p.Hb = take(unfused ? V * E : 0);              // the recomputed hidden tensor of the unfused path (the forward pass's own copy is saved state: read-only here)
        size_t gmax = (size_t)B * (Hin + 2) * (Hin + 2) * T * F;
        {   // gradients of the (mirror-padded) reducer inputs
            int h = Hin, t = T;
            for (size_t k = 0; k < e->iRed.size(); ++k) {
                const probav_engine::RedSpec& r = e->redSpec[k];
                const size_t q = (size_t)B * (h + 2 * r.p) * (h + 2 * r.p) * (t + 2 * r.pt) * F;
                if (q > gmax) gmax = q;
                h = p.redH[k]; t = p.redT[k];
            }
        }
        p.gA = take(gmax); p.gB = take(gmax); p.gDec = take(V * D);
        // every block's input gradient in its own buffer: the block's backward-filter (needed only by the weight-norm backward at the very end)
        // runs on the low-priority side stream, filling the tails of the main chain's launches, and must find its dY untouched whenever it runs
        p.gblk.clear();
        for (int i = 0; i < R; ++i) p.gblk.push_back(take(V * F));
        p.gred.clear();                                        // the reducers likewise: backward-data output and (mirrored pads) its folded form
        for (size_t k = 0; k < 2 * e->iRed.size(); ++k) p.gred.push_back(take(gmax));
        p.dtail = take((size_t)B * P * P * s2);
        p.dr2 = take((size_t)B * (Hin - 4) * (Hin - 4) * s2);
        p.dr1 = take((size_t)B * (Hin - 2) * (Hin - 2) * s2);
        p.dH = take(unfused ? V * E : 0);
        // one slab region per backward-filter launch, in the order probav_backward issues them: the slabs of a launch are summed on the
        // side stream while the main chain has moved on, so no two launches may share a region
        {
            std::vector<size_t> need;
            const int nred = (int)e->iRed.size();
            need.push_back(wgrad_need(e, make_geom(B, Hin - 4, 1, s2, P, 1, s2, 3, 3, 1, 0, 0, 0, 0)));           // residConv3, 2, 1
            if (resid_path_fused(e)) need.back() = std::max(need.back(), resid_path_slab_floats(B, c.in_channels));        // (the fused reverse pass: one slab per patch, in the first region)
            need.push_back(wgrad_need(e, make_geom(B, Hin - 2, 1, s2, Hin - 4, 1, s2, 3, 3, 1, 0, 0, 0, 0)));
            need.push_back(wgrad_need(e, make_geom(B, Hin, 1, c.in_channels, Hin - 2, 1, s2, 3, 3, 1, 0, 0, 0, 1)));
            need.push_back(wgrad_need(e, make_geom(B, p.redH[nred - 1], p.redT[nred - 1], F, P, 1, s2, 3, 3, 3, 0, 0, 0, 0)));   // upscaleConv1
            for (int k = nred - 1; k >= 0; --k)
                need.push_back(wgrad_need(e, red_geom(e, B, (size_t)k, k ? p.redH[k - 1] : Hin, k ? p.redT[k - 1] : T, F)));
            const ConvGeom ge = make_geom(B, Hin, T, F, Hin, T, E, 1, 1, 1, 0, 0, 0, 1), gd = make_geom(B, Hin, T, E, Hin, T, D, 1, 1, 1, 0, 0, 0, 0);
            const ConvGeom gn = make_geom(B, Hin, T, D, Hin, T, F, 3, 3, 3, 1, 1, 0, 0);
            for (int i = 0; i < R; ++i) {
                need.push_back(wgrad_need(e, gn));
                if (!unfused) need.push_back(mfma_pw_backward_slab_floats(D));
                else { need.push_back(wgrad_need(e, gd)); need.push_back(wgrad_need(e, ge)); }
            }
            need.push_back(wgrad_need(e, make_geom(B, Hin, T, c.in_channels, Hin, T, F, 3, 3, 3, 1, 1, 0, 1)));               // mainConv1
            size_t acc = 0;
            for (size_t q : need) { p.part_off.push_back(acc); acc += (q + 63) & ~(size_t)63; }
            p.partial = take(acc);
        }
    }
    p.bwd_total = off;
    p.total = p.fwd_total + p.bwd_total;
    return p;
}

// Weight cache (optional, caller-owned): everything a forward / backward pass derives from the parameters alone -- effective weights in
// both layouts, inverse norms, the weights' amax slots, the packed MFMA operand fragments.  probav_optimizer_step_fused fills it for the
// parameters it has just updated; the *_wc entry points then skip the weight-norm and packing launches.
struct WcPlan { size_t weff, weffT, invn, amax, wpack, total; int n_wamax; };
static WcPlan make_wc_plan(const probav_engine* e)
{
    WcPlan c; size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n); return o; };
    c.weff = take(e->weff_count); c.weffT = take(e->weff_count); c.invn = take(e->cout_total);
    c.n_wamax = 2 * (int)e->layers.size() + (int)e->cout_total + (int)e->cin_total;
    c.amax = take((size_t)c.n_wamax); c.wpack = take(e->wpack_count);
    c.total = off;
    return c;
}

// ---------------------------------------------------------------------------------------------------
struct Frags { const float* f32 = nullptr; const float* x6 = nullptr; const float* h3 = nullptr; const float* h3t = nullptr; };   // h3t: per-tap H3 fragments of a 25-channel layer

// amax slot addresses inside the workspace (layout: make_plan)
struct AmaxSlots {
    const probav_engine* e; unsigned* base; unsigned* wbase; unsigned* bbase; int L, R, B, fwd;
    // wslots: where the weights' slots live -- the head of the workspace's amax region, or the weight cache's; S: the reverse pass's scratch (its slots live there)
    AmaxSlots(const probav_engine* e_, const Plan& p, const float* W, int R_, unsigned* wslots = nullptr, float* S = nullptr)
        : e(e_), base(reinterpret_cast<unsigned*>(const_cast<float*>(W) + p.amax)), wbase(wslots ? wslots : reinterpret_cast<unsigned*>(const_cast<float*>(W) + p.amax)),
          bbase(S ? reinterpret_cast<unsigned*>(S + p.bamax) : nullptr), L((int)e_->layers.size()), R(R_), B(p.B), fwd(p.amax_fwd) {}
    unsigned* w(int li) const { return wbase + li; }                                   // whole weight tensor of layer li
    unsigned* b(int li) const { return wbase + L + li; }                               // its bias
    unsigned* wcol(int li) const { return wbase + 2 * L + e->layers[li].wn.n_off; }     // per output channel (Cout slots): columns of the forward matrices
    unsigned* wrow(int li) const { return wbase + 2 * L + (int)e->cout_total + e->layers[li].wn.r_off; }   // per input channel (Cin slots): columns of the backward-data matrices
    unsigned* act(int i) const { return base + fwd + B * i; }                         // per-sample arrays (B slots each)
    unsigned* dec(int i) const { return base + fwd + B * (R + 1 + i); }
    unsigned* red(int k) const { return base + fwd + B * (2 * R + 1 + k); }
    unsigned* back(int j) const { return bbase + B * j; }                             // j-th tensor produced by the backward pass
};

static int conv_fwd_launch(const probav_engine* e, const ConvGeom& g, const float* x, const float* gate, const float* w,
                           const Frags& wf, const float* bias, const float* skip, float* y, const Amax& am, bool& reported, hipStream_t s)
{
    const float* wfrag = wf.f32;
    const bool pw = g.kh * g.kw * g.kt == 1;
    const bool bwd = (bias == nullptr);              // only backward-data launches run without a bias
    reported = false;                                // does the kernel write am.y itself?
    // the experimental 19-frame reducer: 5x5x5 kernels, pads of 2, mirrored depth pads (and their backward-data forms): generic kernels
    const bool exotic = g.reflect_t || g.ph > 2 || g.pw > 2 || g.pt > 2 || (!pw && g.kh != 3) || (g.reflect_hw && g.ph > 1);
    // upscaleConv1 and its backward-data (0.2 % of the work): dedicated small VALU kernels instead of 32x32 matrix tiles around a 32x9 product
    if (e->impl >= 1 && !gate && !skip && bwd && conv3d_up_bwd_data_supported(g)) {
        ProfScope ps(e, CLS_CONV3_BWD_DATA, geom_macs(g), s);
        reported = true;
        return conv3d_up_bwd_data(g, x, w, nullptr, y, am.y, s);
    }
    if (exotic) { ProfScope ps(e, bwd ? CLS_CONV3_BWD_DATA : CLS_CONV3_FWD, geom_macs(g), s); return conv3d_direct_forward(g, x, gate, w, bias, skip, y, s); }
    if (e->impl >= 1 && !gate && !skip && bias && !am.y && conv3d_up_forward_supported(g)) {
        ProfScope ps(e, CLS_CONV3_FWD, geom_macs(g), s);
        return conv3d_up_forward(g, x, w, bias, y, s);
    }
    if (e->impl >= 1 && !gate && !skip && bias && conv3d_cin1_forward_supported(g)) {
        ProfScope ps(e, CLS_CONV3_FWD, geom_macs(g), s);
        reported = true;
        return conv3d_cin1_forward(g, x, w, bias, y, am.y, s);
    }
    const bool no_strip = false;
    const bool h3 = e->impl >= 4 && wf.h3 && am.x && am.w;
    const float* wsplit = h3 ? wf.h3 : wf.x6;
    const int arith = h3 ? 2 : 1;
    const bool pring = h3 && x6_strip_wants_tap_fragments(g, 2);                  // the H3 piece-ring strip kernel serves this geometry
    const bool x6s = e->impl >= 3 && wsplit && !no_strip && (mfma_conv_strip_supported(g) || pring);
    const bool x6r = e->impl >= 3 && wsplit && !x6s && x6_conv_rowtile_supported(g);
    const bool x6 = x6s || x6r;
    ProfScope ps(e, pw ? (bwd ? CLS_PW_BWD_DATA : CLS_PW_FWD) : (bwd ? (x6 ? CLS_CONV3_BWD_DATA_X6 : CLS_CONV3_BWD_DATA) : (x6 ? CLS_CONV3_FWD_X6 : CLS_CONV3_FWD)), geom_macs(g), s);
    if (x6s) {
        reported = true;
        const float* wq = (pring && g.Cin == 25 && wf.h3t) ? wf.h3t : wsplit;
        return x6_conv_strip_forward(g, x, gate, wq, bias, skip, y, arith, am, s);
    }
    if (x6r) { reported = true; return x6_conv_rowtile_forward(g, x, gate, wsplit, bias, skip, y, arith, am, s); }
    if (e->impl >= 2 && wfrag && mfma_conv_strip_supported(g)) { reported = true; return mfma_conv_strip_forward(g, x, gate, wfrag, bias, skip, y, am, s); }
    if (e->impl >= 1 && wfrag && mfma_conv_supported(g)) { reported = true; return mfma_conv_forward(g, x, gate, wfrag, bias, skip, y, am, s); }
    return conv3d_direct_forward(g, x, gate, w, bias, skip, y, s);
}
// am.x / am.w: amax slots of x (one per sample) and of the layer's filter columns (H3 kernels); am.y: per-sample slots that must hold the output's amax afterwards
static int conv_fwd(const probav_engine* e, const ConvGeom& g, const float* x, const float* gate, const float* w,
                    const Frags& wf, const float* bias, const float* skip, float* y, const Amax& am, hipStream_t s)
{
    bool reported = false;
    int rc = conv_fwd_launch(e, g, x, gate, w, wf, bias, skip, y, am, reported, s);
    if (rc == PROBAV_OK && am.y && !reported) rc = amax_tensor(y, (size_t)g.Ho * g.Wo * g.To * g.Cout, g.N, am.y, s);
    return rc;
}
// am.x / am.w: per-sample amax slots of x and of dy (H3 backward-filter kernel)
static int conv_wgrad(const probav_engine* e, const ConvGeom& g, const float* x, const float* dy, const float* gate,
                      float* dw, float* db, float* partial, const Amax& am, hipStream_t s)
{
    const bool exotic = g.reflect_t || g.ph > 1 || g.pw > 1 || g.pt > 1 || (g.kh != 3 && g.kh != 1);
    if (exotic || (e->impl >= 1 && conv3d_direct_wgrad_is_tuned(g))) { ProfScope ps(e, CLS_CONV3_WGRAD, geom_macs(g), s); return conv3d_direct_wgrad(g, x, dy, gate, dw, db, partial, s); }
    const bool x6 = e->impl >= 3 && x6_wgrad_supported(g);
    ProfScope ps(e, g.kh * g.kw * g.kt == 1 ? CLS_PW_WGRAD : (x6 ? CLS_CONV3_WGRAD_X6 : CLS_CONV3_WGRAD), geom_macs(g), s);
    if (x6) { const bool h3 = e->impl >= 4 && am.x && am.w; return x6_conv_wgrad(g, x, dy, gate, dw, db, partial, h3 ? 2 : 1, am, s); }
    if (e->impl >= 1 && mfma_wgrad_supported(g)) return mfma_conv_wgrad(g, x, dy, gate, dw, db, partial, s);
    return conv3d_direct_wgrad(g, x, dy, gate, dw, db, partial, s);
}

#define CK(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" {

int probav_abi_version(void) { return PROBAV_ABI_VERSION; }
int probav_engine_side_stream(probav_engine* e, int mode)
{
    if (!e || mode < 0 || mode > 2) { set_error("probav_engine_side_stream: bad argument", hipSuccess); return PROBAV_EINVAL; }
    e->side_mode = mode;
    return PROBAV_OK;
}
int probav_mfma_probe(const void* seed, float* sink, int iters, int launches, void* stream)
{
    if (!seed || !sink || iters < 1 || launches < 1) { set_error("probav_mfma_probe: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    return mfma_probe(seed, sink, iters, launches, (hipStream_t)stream);
}
int probav_mfma_probe_shape(const void* seed, float* sink, int iters, int launches, int shape, void* stream)
{
    if (!seed || !sink || iters < 1 || launches < 1 || shape < 0 || shape > 1) { set_error("probav_mfma_probe_shape: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    return mfma_probe(seed, sink, iters, launches, (hipStream_t)stream, shape);
}
const char* probav_last_error(void) { return probav::last_error(); }

int probav_engine_create(const probav_net_cfg* cfg, probav_engine** out)
{
    if (!cfg || !out) { set_error("probav_engine_create: null argument", hipSuccess); return PROBAV_EINVAL; }
    const int T = cfg->num_img_lr;
    if (cfg->scale != 3 || cfg->max_shift != 2 * cfg->scale) {
        set_error("probav_engine_create: the reference graph only closes for scale=3, maxShift=6 (models/modelsTF.py:45-53)", hipSuccess);
        return PROBAV_EINVAL;
    }
    if (T != 7 && T != 9 && T != 13 && T != 19) {
        set_error("probav_engine_create: numImgLR must be 7, 9, 13 or 19 (models/modelsTF.py:62-69)", hipSuccess);
        return PROBAV_EINVAL;
    }
    if (cfg->num_filters < 1 || cfg->num_res_blocks < 0 || cfg->exp_rate < 1 || cfg->dec_channels < 1 ||
        cfg->patch_size_lr < 1 || !(cfg->std > 0.f) || (cfg->in_channels != 1 && cfg->in_channels != 3)) {
        set_error("probav_engine_create: bad hyper-parameter", hipSuccess);
        return PROBAV_EINVAL;
    }
    probav_engine* e = new probav_engine();
    e->cfg = *cfg;
    e->Hin = cfg->patch_size_lr + cfg->max_shift;
    const int F = cfg->num_filters, E = F * cfg->exp_rate, D = cfg->dec_channels, s2 = cfg->scale * cfg->scale;
    const int Cx = cfg->in_channels;                                   // 1, or 3 for isGrayScale=False (models/modelsTF.py:19-20)
    e->iMain = add_layer(e, "mainConv1", 3, 3, 3, Cx, F);
    for (int i = 0; i < cfg->num_res_blocks; ++i) {
        e->iExp.push_back(add_layer(e, "expConv_" + std::to_string(i), 1, 1, 1, F, E));
        e->iDec.push_back(add_layer(e, "decConv_" + std::to_string(i), 1, 1, 1, E, D));
        e->iNorm.push_back(add_layer(e, "normConv_" + std::to_string(i), 3, 3, 3, D, F));
    }
    typedef probav_engine::RedSpec RS;
    if (T == 9) e->redSpec = {RS{3, 1, 0, 1, 0}, RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}};
    else if (T == 13) e->redSpec = {RS{3, 1, 0, 1, 0}, RS{3, 1, 0, 1, 0}, RS{3, 1, 0, 1, 0}, RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}};
    else if (T == 7) e->redSpec = {RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}};
    else   // T == 19, ConvReduceAndUpscaleEx (models/modelsTF.py:76-121, marked EXPERIMENTAL there): a 5x5x5 layer on a fully mirrored pad of 2,
           // four 3x3x3 layers on mirrored H/W pads of 2, 2, 2, 1 (the first also mirrors one frame of depth), five plain valid ones
        e->redSpec = {RS{5, 2, 2, 1, 1}, RS{3, 2, 1, 1, 1}, RS{3, 2, 0, 1, 0}, RS{3, 2, 0, 1, 0}, RS{3, 1, 0, 1, 0},
                      RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}, RS{3, 0, 0, 0, 0}};
    for (size_t k = 0; k < e->redSpec.size(); ++k)
        e->iRed.push_back(add_layer(e, "convReducer_" + std::to_string(k + 1), e->redSpec[k].k, e->redSpec[k].k, e->redSpec[k].k, F, F));
    e->iResid1 = add_layer(e, "residConv1", 3, 3, 1, Cx, s2);
    e->iUp = add_layer(e, "upscaleConv1", 3, 3, 3, F, s2);
    e->iResid2 = add_layer(e, "residConv2", 3, 3, 1, s2, s2);
    e->iResid3 = add_layer(e, "residConv3", 3, 3, 1, s2, s2);
    // the graph must close: after the reducers one valid 3x3x3 conv lands on [P, P, 1]
    std::vector<int> hh, tt;
    reducer_extents(e, hh, tt);
    if (hh.back() - 2 != cfg->patch_size_lr || tt.back() - 2 != 1) {
        delete e;
        set_error("probav_engine_create: reducer geometry does not collapse to [P,P,1]", hipSuccess);
        return PROBAV_EINVAL;
    }
    // MFMA fragment-packing jobs
    e->pkFwd.assign(e->layers.size(), -1); e->pkBwd.assign(e->layers.size(), -1);
    e->pkFwd6.assign(e->layers.size(), -1); e->pkBwd6.assign(e->layers.size(), -1);
    e->pkFwdH.assign(e->layers.size(), -1); e->pkBwdH.assign(e->layers.size(), -1);
    e->pkFwdHt.assign(e->layers.size(), -1); e->pkBwdHt.assign(e->layers.size(), -1);
    for (size_t li = 0; li < e->layers.size(); ++li) {
        const LayerRec& r = e->layers[li];
        if (r.kh != 3 || r.kw != 3 || r.kt != 3) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int cin = dir ? r.wn.Cout : r.wn.Cin, cout = dir ? r.wn.Cin : r.wn.Cout;
            if ((int)li == e->iMain && dir) continue;               // input-facing: no backward-data
            if (mfma_conv_wfrag_floats(cin, cout) == 0) continue;
            PackJob J; memset(&J, 0, sizeof(J));
            mfma_conv_pack_job(J, cin, cout);
            J.src_is_T = dir; J.src_off = r.wn.w_off; J.dst_off = e->wpack_count;
            (dir ? e->pkBwd : e->pkFwd)[li] = J.dst_off;
            e->wpack_count += J.count;
            e->jobs.push_back(J);
            if ((cin == 25 || cin == 32) && cout <= 32) {           // strip-kernel shapes: also the x6 fragments
                PackJob X; memset(&X, 0, sizeof(X));
                X.type = cin == 25 ? PACK_X6_CONVK : PACK_X6_CONV; X.src_is_T = dir; X.src_off = r.wn.w_off; X.dst_off = e->wpack_count;
                X.count = cin == 25 ? X6_CONVK_FRAG_WORDS : X6_CONV_FRAG_WORDS; X.Cin = cin; X.Cout = cout; X.taps = 27;
                (dir ? e->pkBwd6 : e->pkFwd6)[li] = X.dst_off;
                e->wpack_count += X.count;
                e->jobs.push_back(X);
                X.type += 10; X.dst_off = e->wpack_count;                              // PACK_H3_*: cut per output column of the packed matrix
                X.amax_percol = 1; X.ncol = cout;                                      // columns of weff = output channels, of weffT = input channels
                X.amax_slot = 2 * (int)e->layers.size() + (dir ? (int)e->cout_total + r.wn.r_off : r.wn.n_off);
                X.count = cin == 25 ? H3_CONVK_FRAG_WORDS : H3_CONV_FRAG_WORDS;
                (dir ? e->pkBwdH : e->pkFwdH)[li] = X.dst_off;
                e->wpack_count += X.count;
                e->jobs.push_back(X);
                if (cin == 25) {
                    X.type = PACK_H3_CONVP; X.count = H3_CONVK_FRAG_WORDS; X.dst_off = e->wpack_count;
                    (dir ? e->pkBwdHt : e->pkFwdHt)[li] = X.dst_off;
                    e->wpack_count += X.count;
                    e->jobs.push_back(X);
                }
            }
        }
    }
    e->pw_mfma = mfma_pw_supported(F, E, D);
    if (e->pw_mfma) {
        for (int i = 0; i < cfg->num_res_blocks; ++i) {
            PackJob J; memset(&J, 0, sizeof(J));
            J.type = PACK_PW_A_KCIN; J.src_is_T = 0; J.src_off = e->layers[e->iExp[i]].wn.w_off; J.dst_off = e->wpack_count;
            J.count = 8 * 4 * 64 * 4; J.Cin = F; J.Cout = E;
            e->pkW1.push_back(J.dst_off); e->wpack_count += J.count; e->jobs.push_back(J);
            J.type = PACK_PW_A_KHCH; J.src_off = e->layers[e->iDec[i]].wn.w_off; J.dst_off = e->wpack_count;
            J.Cin = E; J.Cout = D;
            e->pkW2.push_back(J.dst_off); e->wpack_count += J.count; e->jobs.push_back(J);
            J.type = PACK_PW_A_KOUT; J.dst_off = e->wpack_count;                      // backward (b): dH^T = W2 dT^T
            e->pkW2B.push_back(J.dst_off); e->wpack_count += J.count; e->jobs.push_back(J);
            J.type = PACK_PW_A_CIN_KHCH; J.src_off = e->layers[e->iExp[i]].wn.w_off; J.dst_off = e->wpack_count;
            J.Cin = F; J.Cout = E;                                                    // backward (c): dX^T += W1 dH'^T
            e->pkW1C.push_back(J.dst_off); e->wpack_count += J.count; e->jobs.push_back(J);
            PackJob X; memset(&X, 0, sizeof(X));
            X.type = PACK_X6_PW_W1; X.src_off = e->layers[e->iExp[i]].wn.w_off; X.dst_off = e->wpack_count;
            X.count = X6_PW_FRAG_WORDS; X.Cin = F; X.Cout = E;
            e->pkW1x6.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_X6_PW_W2; X.src_off = e->layers[e->iDec[i]].wn.w_off; X.dst_off = e->wpack_count;
            X.Cin = E; X.Cout = D;
            e->pkW2x6.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_X6_PW_W2K; X.dst_off = e->wpack_count;
            e->pkW2Kx6.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_X6_PW_W1C; X.src_off = e->layers[e->iExp[i]].wn.w_off; X.dst_off = e->wpack_count;
            X.Cin = F; X.Cout = E;
            e->pkW1Cx6.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.count = H3_PW_FRAG_WORDS;
            X.type = PACK_H3_PW_W1; X.src_off = e->layers[e->iExp[i]].wn.w_off; X.dst_off = e->wpack_count;
            const int L2 = 2 * (int)e->layers.size();
            X.Cin = F; X.Cout = E; X.amax_slot = e->iExp[i]; X.amax_percol = 0; X.ncol = 0;         // (a): one scale for the tensor
            e->pkW1h.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_H3_PW_W1C; X.dst_off = e->wpack_count;                                        // (c): cut per cin row = per dX column
            X.amax_percol = 1; X.ncol = F; X.amax_slot = L2 + (int)e->cout_total + e->layers[e->iExp[i]].wn.r_off;
            e->pkW1Ch.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_H3_PW_W2; X.src_off = e->layers[e->iDec[i]].wn.w_off; X.dst_off = e->wpack_count;
            X.Cin = E; X.Cout = D; X.amax_percol = 1; X.ncol = D; X.amax_slot = L2 + e->layers[e->iDec[i]].wn.n_off;   // forward: cut per output column d
            e->pkW2h.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
            X.type = PACK_H3_PW_W2K; X.dst_off = e->wpack_count; X.amax_percol = 0; X.ncol = 0; X.amax_slot = e->iDec[i];       // (b): one scale for the tensor
            e->pkW2Kh.push_back(X.dst_off); e->wpack_count += X.count; e->jobs.push_back(X);
        }
    }
    if (!e->jobs.empty()) {
        hipError_t perr = hipMalloc((void**)&e->d_jobs, e->jobs.size() * sizeof(PackJob));
        if (perr == hipSuccess) perr = hipMemcpy(e->d_jobs, e->jobs.data(), e->jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice);
        if (perr != hipSuccess) { set_error("probav_engine_create: pack table upload", perr); delete e; return PROBAV_EHIP; }
    }
    std::vector<WnLayer> h;
    for (auto& r : e->layers) h.push_back(r.wn);
    hipError_t err = hipMalloc((void**)&e->d_layers, h.size() * sizeof(WnLayer));
    if (err == hipSuccess) err = hipMemcpy(e->d_layers, h.data(), h.size() * sizeof(WnLayer), hipMemcpyHostToDevice);
    if (err != hipSuccess) { set_error("probav_engine_create: layer table upload", err); delete e; return PROBAV_EHIP; }
    if (const char* env = getenv("PROBAV_IMPL")) e->impl = atoi(env);
    *out = e;
    return PROBAV_OK;
}

int probav_engine_profile(probav_engine* e, int enable, int max_launches)
{
    if (!e || max_launches < 0) { set_error("probav_engine_profile: bad argument", hipSuccess); return PROBAV_EINVAL; }
    while ((int)e->prof_ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        hipError_t err = hipEventCreate(&ev);
        if (err != hipSuccess) { set_error("probav_engine_profile: hipEventCreate", err); return PROBAV_EHIP; }
        e->prof_ev.push_back(ev);
    }
    e->prof_on = enable != 0;
    e->prof_used = 0; e->prof_cls.clear(); e->prof_macs.clear();
    return PROBAV_OK;
}

int probav_engine_profile_classes(probav_engine* e, uint32_t mask)
{
    if (!e) { set_error("probav_engine_profile_classes: null engine", hipSuccess); return PROBAV_EINVAL; }
    e->prof_mask = mask;
    return PROBAV_OK;
}

int probav_engine_profile_read(probav_engine* e, int nclass, double* ms, double* macs, int64_t* launches)
{
    if (!e || !ms || !macs || !launches || nclass < CLS_COUNT) { set_error("probav_engine_profile_read: bad argument", hipSuccess); return PROBAV_EINVAL; }
    for (int c = 0; c < nclass; ++c) { ms[c] = 0; macs[c] = 0; launches[c] = 0; }
    for (size_t i = 0; i < e->prof_cls.size(); ++i) {
        float t = 0.f;
        hipError_t err = hipEventElapsedTime(&t, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]);     // caller has synchronised
        if (err != hipSuccess) { set_error("probav_engine_profile_read: hipEventElapsedTime", err); return PROBAV_EHIP; }
        const int c = e->prof_cls[i];
        ms[c] += t; macs[c] += e->prof_macs[i]; launches[c] += 1;
    }
    e->prof_used = 0; e->prof_cls.clear(); e->prof_macs.clear();
    return PROBAV_OK;
}

void probav_engine_destroy(probav_engine* e)
{
    if (!e) return;
    for (auto ev : e->prof_ev) (void)hipEventDestroy(ev);
    reduce_free_pending(&e->side);
    if (e->side.side) {
        (void)hipStreamSynchronize(e->side.side);
        for (auto ev : e->side.ev) (void)hipEventDestroy(ev);
        (void)hipEventDestroy(e->side.joined);
        (void)hipStreamDestroy(e->side.side);
    }
    if (e->d_layers) (void)hipFree(e->d_layers);
    if (e->d_jobs) (void)hipFree(e->d_jobs);
    delete e;
}

int64_t probav_param_count(const probav_engine* e) { return e ? e->nparams : -1; }
int probav_num_layers(const probav_engine* e) { return e ? (int)e->layers.size() : -1; }
int64_t probav_weff_count(const probav_engine* e) { return e ? e->weff_count : -1; }
int64_t probav_cout_total(const probav_engine* e) { return e ? e->cout_total : -1; }

int probav_layer_info(const probav_engine* e, int i, char name[32], int64_t* g_off, int64_t* v_off, int64_t* b_off, int32_t shape[5])
{
    if (!e || i < 0 || i >= (int)e->layers.size()) { set_error("probav_layer_info: bad index", hipSuccess); return PROBAV_EINVAL; }
    const LayerRec& r = e->layers[i];
    if (name) memcpy(name, r.name, 32);
    if (g_off) *g_off = r.wn.g_off;
    if (v_off) *v_off = r.wn.v_off;
    if (b_off) *b_off = r.wn.b_off;
    if (shape) { shape[0] = r.kh; shape[1] = r.kw; shape[2] = r.kt; shape[3] = r.wn.Cin; shape[4] = r.wn.Cout; }
    return PROBAV_OK;
}

int probav_engine_set_impl(probav_engine* e, int impl)
{
    if (!e || impl < 0 || impl > 4) { set_error("probav_engine_set_impl: bad argument", hipSuccess); return PROBAV_EINVAL; }
    e->impl = impl;
    return PROBAV_OK;
}

size_t probav_workspace_bytes(const probav_engine* e, int batch, int training)
{
    if (!e || batch < 1) return 0;
    return make_plan(e, batch, training).total * sizeof(float);
}

static ReduceSide* engine_side(probav_engine* e)
{
    if (!e->side_tried) {
        e->side_tried = true;
        ReduceSide c = {};
        // lowest priority: its kernels are fillers for the gaps of the caller's chain, never competitors
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = 0; }
        bool ok = hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, prio_least) == hipSuccess;
        for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&c.ev[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c.joined, hipEventDisableTiming) == hipSuccess;
        if (ok) e->side = c; else (void)hipGetLastError();          // (without it the sums simply stay on the caller's stream)
    }
    return e->side.side ? &e->side : nullptr;
}

struct SideGuard {          // activates the engine's side stream (probav_common.h: ReduceSide) for the calling thread while a pass is being enqueued
    ReduceSide* c;
    hipStream_t s;
    SideGuard(ReduceSide* c_, hipStream_t s_, int defer = 0) : c(c_), s(s_) { if (c) { c->k = 0; c->last = nullptr; c->defer = defer; reduce_side_activate(c); reduce_drop_pending(); } }
    ~SideGuard()
    {
        if (!c) return;
        reduce_drop_pending();                      // (a pass that returned early: queued launches may point into its frame -- never run them here)
        if (c->k != 0) (void)reduce_join(s);        // a pass that returned early (an error): whatever was forked still rejoins the caller's stream
        c->defer = 0;
        reduce_side_activate(nullptr);
    }
};
static bool side_stream_disabled() { static const bool v = getenv("PROBAV_NO_SIDE_STREAM") != nullptr; return v; }   // diagnostic: everything on the caller's stream

static int forward_impl(probav_engine* e, const float* params, const float* x, float* y, void* ws, size_t ws_bytes,
                        int B, int training, const float* WC, void* stream)
{
    if (!e || !params || !x || !y || !ws || B < 1) { set_error("probav_forward: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Plan p = make_plan(e, B, training);
    if (ws_bytes < p.fwd_total * sizeof(float)) { set_error("probav_forward: workspace too small", hipSuccess); return PROBAV_ENOSPACE; }
    float* W = (float*)ws;
    const WcPlan wc = make_wc_plan(e);
    // where the parameter-derived tensors live: inside the workspace (recomputed by this call) or in the caller's weight cache
    const float* Wweff = WC ? WC + wc.weff : W + p.weff;
    const float* Wpack = WC ? WC + wc.wpack : W + p.wpack;
    const probav_net_cfg& c = e->cfg;
    const int F = c.num_filters, E = F * c.exp_rate, D = c.dec_channels, T = c.num_img_lr, R = c.num_res_blocks;
    const int Hin = e->Hin, P = c.patch_size_lr, s2 = c.scale * c.scale;
    auto weff = [&](int li) { return Wweff + e->layers[li].wn.w_off; };
    auto bias = [&](int li) { return params + e->layers[li].wn.b_off; };
    auto frag = [&](int li) -> Frags {
        Frags f;
        if (e->pkFwd[li] >= 0) f.f32 = Wpack + e->pkFwd[li];
        if (e->pkFwd6[li] >= 0) f.x6 = Wpack + e->pkFwd6[li];
        if (e->pkFwdH[li] >= 0) f.h3 = Wpack + e->pkFwdH[li];
        if (e->pkFwdHt[li] >= 0) f.h3t = Wpack + e->pkFwdHt[li];
        return f;
    };
    // amax slots (H3 arithmetic, impl 4): every tensor an H3 kernel reads has its largest magnitude in a slot by then
    const bool h3 = e->impl >= 4;
    const AmaxSlots A(e, p, W, R, WC ? reinterpret_cast<unsigned*>(const_cast<float*>(WC + wc.amax)) : nullptr);
    auto amx = [&](const unsigned* ax, int li, unsigned* ay) { Amax m; if (h3) { m.x = ax; m.w = A.wcol(li); m.y = ay; } return m; };
    // (the per-sample amax slots -- the atomicMax targets -- are cleared by head_kernel below; the weight slots in front of them are plain stores of wn_forward_kernel /
    // wn_rowmax_kernel, every one of them written before anything reads it)
    if (training) { e->fwd_amax = h3; e->fwd_unfused = !(e->impl >= 1 && e->pw_mfma); }

    if (!WC) {
        { ProfScope ps(e, CLS_WN, 0.0, s); CK(wn_forward(e->d_layers, (int)e->layers.size(), (int)e->cout_total, (int)e->cin_total, params, W + p.weff, W + p.weffT, W + p.invn, h3 ? A.base : nullptr, s)); }
        if (e->impl >= 1) { ProfScope ps(e, CLS_WN, 0.0, s); CK(mfma_pack(e->d_jobs, (int)e->jobs.size(), W + p.weff, W + p.weffT, W + p.wpack, A.base, s)); }
    }
    CK(head_forward(x, W + p.xn, W + p.mn, B * Hin * Hin, T, c.in_channels, c.mean, c.std, s, h3 ? A.base + p.amax_fwd : nullptr, h3 ? p.amax_bwd - p.amax_fwd : 0));
    // the low-frequency residual path (three small 2-D convolutions on the temporal mean) meets the main path only in tail_forward: it runs
    // on the side stream, in the gaps of the chip-filling launches
    SideGuard side_guard((side_stream_disabled() || e->side_mode == 0) ? nullptr : engine_side(e), s);
    {
        hipStream_t rs = reduce_fork(s);
        if (resid_path_fused(e)) {                         // the three layers as one launch (kernels_direct.hip); family 0 keeps the generic direct kernels
            CK(resid_path_forward(B, Hin, c.in_channels, W + p.mn, weff(e->iResid1), bias(e->iResid1), weff(e->iResid2), bias(e->iResid2), weff(e->iResid3), bias(e->iResid3),
                                  W + p.r1, W + p.r2, W + p.r3, rs));
        } else {
        CK(conv_fwd(e, make_geom(B, Hin, 1, c.in_channels, Hin - 2, 1, s2, 3, 3, 1, 0, 0, 0, 1), W + p.mn, nullptr, weff(e->iResid1), frag(e->iResid1), bias(e->iResid1), nullptr, W + p.r1, Amax(), rs));
        CK(conv_fwd(e, make_geom(B, Hin - 2, 1, s2, Hin - 4, 1, s2, 3, 3, 1, 0, 0, 0, 0), W + p.r1, nullptr, weff(e->iResid2), frag(e->iResid2), bias(e->iResid2), nullptr, W + p.r2, Amax(), rs));
        CK(conv_fwd(e, make_geom(B, Hin - 4, 1, s2, P, 1, s2, 3, 3, 1, 0, 0, 0, 0), W + p.r2, nullptr, weff(e->iResid3), frag(e->iResid3), bias(e->iResid3), nullptr, W + p.r3, Amax(), rs));
        }
    }
    CK(conv_fwd(e, make_geom(B, Hin, T, c.in_channels, Hin, T, F, 3, 3, 3, 1, 1, 0, 1), W + p.xn, nullptr, weff(e->iMain), frag(e->iMain), bias(e->iMain), nullptr, W + p.act[0], amx(nullptr, e->iMain, A.act(0)), s));
    for (int i = 0; i < R; ++i) {
        if (e->impl >= 1 && e->pw_mfma) {
            // fused expConv + ReLU + decConv: the 256-channel tensor never leaves the accumulators
            const long nvox = (long)B * Hin * Hin * T;
            ProfScope ps(e, e->impl >= 3 ? CLS_PW_FWD_X6 : CLS_PW_FWD, (double)nvox * ((double)F * E + (double)E * D), s);
            if (h3) {
                PwAmax m; m.x = A.act(i); m.w1 = A.w(e->iExp[i]); m.w2 = A.w(e->iDec[i]); m.w2c = A.wcol(e->iDec[i]); m.b1 = A.b(e->iExp[i]); m.y = A.dec(i);
                CK(x6_pw_forward(W + p.act[i], Wpack + e->pkW1h[i], Wpack + e->pkW2h[i], bias(e->iExp[i]), bias(e->iDec[i]),
                                 W + p.dec[i], nvox, nvox / B, D, 2, m, s));
            } else if (e->impl >= 3) {
                CK(x6_pw_forward(W + p.act[i], Wpack + e->pkW1x6[i], Wpack + e->pkW2x6[i], bias(e->iExp[i]), bias(e->iDec[i]),
                                 W + p.dec[i], nvox, nvox / B, D, 1, PwAmax(), s));
            } else
                CK(mfma_pw_forward(W + p.act[i], Wpack + e->pkW1[i], Wpack + e->pkW2[i], bias(e->iExp[i]), bias(e->iDec[i]),
                                   W + p.dec[i], nvox, D, s));
        } else {
            CK(conv_fwd(e, make_geom(B, Hin, T, F, Hin, T, E, 1, 1, 1, 0, 0, 0, 1), W + p.act[i], nullptr, weff(e->iExp[i]), frag(e->iExp[i]), bias(e->iExp[i]), nullptr, W + p.H, Amax(), s));
            CK(conv_fwd(e, make_geom(B, Hin, T, E, Hin, T, D, 1, 1, 1, 0, 0, 0, 0), W + p.H, nullptr, weff(e->iDec[i]), frag(e->iDec[i]), bias(e->iDec[i]), nullptr, W + p.dec[i], amx(nullptr, e->iDec[i], A.dec(i)), s));
        }
        CK(conv_fwd(e, make_geom(B, Hin, T, D, Hin, T, F, 3, 3, 3, 1, 1, 0, 0), W + p.dec[i], nullptr, weff(e->iNorm[i]), frag(e->iNorm[i]), bias(e->iNorm[i]), W + p.act[i], W + p.act[i + 1], amx(A.dec(i), e->iNorm[i], A.act(i + 1)), s));
    }
    const float* cur = W + p.act[R];
    const unsigned* acur = A.act(R);
    int h = Hin, t = T;
    for (size_t k = 0; k < e->iRed.size(); ++k) {
        CK(conv_fwd(e, red_geom(e, B, k, h, t, F), cur, nullptr,
                    weff(e->iRed[k]), frag(e->iRed[k]), bias(e->iRed[k]), nullptr, W + p.red[k], amx(acur, e->iRed[k], A.red((int)k)), s));
        cur = W + p.red[k]; acur = A.red((int)k); h = p.redH[k]; t = p.redT[k];
    }
    CK(conv_fwd(e, make_geom(B, h, t, F, P, 1, s2, 3, 3, 3, 0, 0, 0, 0), cur, nullptr, weff(e->iUp), frag(e->iUp), bias(e->iUp), nullptr, W + p.up, amx(acur, e->iUp, nullptr), s));
    CK(reduce_join(s));                                                       // the residual path has arrived
    CK(tail_forward(W + p.up, W + p.r3, y, B, P, c.scale, c.mean, c.std, s));
    return PROBAV_OK;
}

int probav_forward(probav_engine* e, const float* params, const float* x, float* y, void* ws, size_t ws_bytes,
                   int B, int training, void* stream)
{
    return forward_impl(e, params, x, y, ws, ws_bytes, B, training, nullptr, stream);
}
int probav_forward_wc(probav_engine* e, const float* params, const float* x, float* y, void* ws, size_t ws_bytes,
                      int B, int training, const void* wcache, size_t wcache_bytes, void* stream)
{
    if (!e || !wcache || wcache_bytes < make_wc_plan(e).total * sizeof(float)) { set_error("probav_forward_wc: weight cache missing / too small", hipSuccess); return PROBAV_EINVAL; }
    return forward_impl(e, params, x, y, ws, ws_bytes, B, training, (const float*)wcache, stream);
}

// ws: the saved state of the matching forward pass (READ ONLY here); scratch: what the reverse pass writes on its way (gradient buffers, slabs, its amax
// slots; contents meaningless before and after).  scratch == nullptr: the one-piece form, the scratch is the tail of `ws`.
static int backward_impl(probav_engine* e, const float* params, const float* dy, float* grads, const void* ws, size_t ws_bytes,
                         void* scratch, size_t scratch_bytes, int B, const float* WC, void* stream)
{
    if (!e || !params || !dy || !grads || !ws || B < 1) { set_error("probav_backward: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Plan p = make_plan(e, B, 1);
    if (ws_bytes < (scratch ? p.fwd_total : p.total) * sizeof(float)) { set_error("probav_backward: workspace too small", hipSuccess); return PROBAV_ENOSPACE; }
    if (scratch && scratch_bytes < p.bwd_total * sizeof(float)) { set_error("probav_backward: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
    const float* W = (const float*)ws;
    float* S = scratch ? (float*)scratch : const_cast<float*>(W) + p.fwd_total;
    const WcPlan wc = make_wc_plan(e);
    const float* Wweff = WC ? WC + wc.weff : W + p.weff;
    const float* WweffT = WC ? WC + wc.weffT : W + p.weffT;
    const float* Winvn = WC ? WC + wc.invn : W + p.invn;
    const float* Wpack = WC ? WC + wc.wpack : W + p.wpack;
    const probav_net_cfg& c = e->cfg;
    const int F = c.num_filters, E = F * c.exp_rate, D = c.dec_channels, T = c.num_img_lr, R = c.num_res_blocks;
    const int Hin = e->Hin, P = c.patch_size_lr, s2 = c.scale * c.scale;
    auto weffT = [&](int li) { return WweffT + e->layers[li].wn.w_off; };
    auto fragT = [&](int li) -> Frags {
        Frags f;
        if (e->pkBwd[li] >= 0) f.f32 = Wpack + e->pkBwd[li];
        if (e->pkBwd6[li] >= 0) f.x6 = Wpack + e->pkBwd6[li];
        if (e->pkBwdH[li] >= 0) f.h3 = Wpack + e->pkBwdH[li];
        if (e->pkBwdHt[li] >= 0) f.h3t = Wpack + e->pkBwdHt[li];
        return f;
    };
    // amax slots of the gradient tensors, in launch order (the forward pass left those of the weights and activations)
    const bool h3 = e->impl >= 4;
    const AmaxSlots A(e, p, W, R, WC ? reinterpret_cast<unsigned*>(const_cast<float*>(WC + wc.amax)) : nullptr, S);
    int nback = 0;
    auto new_slot = [&]() -> unsigned* { return h3 ? A.back(nback++) : nullptr; };
    auto amx = [&](const unsigned* ax, int li, unsigned* ay) { Amax m; if (h3) { m.x = ax; m.w = A.wrow(li); m.y = ay; } return m; };   // backward-data: the matrix' columns are the layer's INPUT channels
    if (e->fwd_unfused != !(e->impl >= 1 && e->pw_mfma)) {
        set_error("probav_backward: the kernel family changed between forward and backward in a way that changes the workspace layout (impl 0 <-> >= 1)", hipSuccess);
        return PROBAV_EINVAL;
    }
    if (h3 && !e->fwd_amax) {
        set_error("probav_backward: the H3 kernels (impl 4) need the amax slots of a forward pass run with the same kernel family", hipSuccess);
        return PROBAV_EINVAL;
    }
    // (the reverse pass's per-sample amax slots are cleared by tail_bwd_kernel, its first launch)
    auto dweff = [&](int li) { return S + p.dweff2 + e->layers[li].wn.w_off; };
    auto dbias = [&](int li) { return grads + e->layers[li].wn.b_off; };
    int npart = 0;
    auto next_part = [&]() -> float* { const size_t k = (size_t)npart < p.part_off.size() ? (size_t)npart : p.part_off.size() - 1; ++npart; return S + p.partial + p.part_off[k]; };
    // (defer: the slab sums and the small launches that only the weight-norm backward waits for are queued and leave in a few flushes -- one event record
    // on the launch stream per flush instead of one per launch, probav_common.h)
    SideGuard side_guard((side_stream_disabled() || e->side_mode == 0) ? nullptr : engine_side(e), s, 1);

    CK(tail_backward(dy, S + p.dtail, B, P, c.scale, c.std, s, h3 ? A.back(0) : nullptr, h3 ? p.n_amax - p.amax_bwd : 0));
    // low-frequency residual path (models/modelsTF.py:45-53), last layer first: beside the main chain, nothing below depends on it until the weight-norm
    // backward -- queued (its slab regions are taken now, in the plan's launch order) and launched at the first flush, when the launch stream has its next kernels
    {
        float* const rp0 = next_part(); float* const rp1 = next_part(); float* const rp2 = next_part();
        // (the queued launch runs later, from reduce_flush: everything it needs is captured BY VALUE -- pointers and extents, nothing of this frame)
        const float *r2 = W + p.r2, *r1 = W + p.r1, *mn = W + p.mn, *wT3 = weffT(e->iResid3), *wT2 = weffT(e->iResid2);
        const Frags fT3 = fragT(e->iResid3), fT2 = fragT(e->iResid2);
        float *dtail = S + p.dtail, *dr2 = S + p.dr2, *dr1 = S + p.dr1;
        float *dw3 = dweff(e->iResid3), *db3 = dbias(e->iResid3), *dw2 = dweff(e->iResid2), *db2 = dbias(e->iResid2), *dw1 = dweff(e->iResid1), *db1 = dbias(e->iResid1);
        const int inch = c.in_channels;
        const float *w2r = Wweff + e->layers[e->iResid2].wn.w_off, *w3r = Wweff + e->layers[e->iResid3].wn.w_off;
        if (resid_path_fused(e))
            CK(reduce_later(s, [=](hipStream_t rs) -> int {    // one launch + one slab sum (rp0 holds the patches' slabs: make_plan sized it)
                return resid_path_backward(B, Hin, inch, mn, r1, r2, dtail, w2r, w3r, dw1, db1, dw2, db2, dw3, db3, rp0, rs); }));
        else
        CK(reduce_later(s, [=](hipStream_t rs) -> int {
            const ConvGeom g3 = make_geom(B, Hin - 4, 1, s2, P, 1, s2, 3, 3, 1, 0, 0, 0, 0);
            CK(conv_wgrad(e, g3, r2, dtail, nullptr, dw3, db3, rp0, Amax(), rs));
            CK(conv_fwd(e, bwd_data_geom(g3), dtail, nullptr, wT3, fT3, nullptr, nullptr, dr2, Amax(), rs));
            const ConvGeom g2 = make_geom(B, Hin - 2, 1, s2, Hin - 4, 1, s2, 3, 3, 1, 0, 0, 0, 0);
            CK(conv_wgrad(e, g2, r1, dr2, nullptr, dw2, db2, rp1, Amax(), rs));
            CK(conv_fwd(e, bwd_data_geom(g2), dr2, nullptr, wT2, fT2, nullptr, nullptr, dr1, Amax(), rs));
            const ConvGeom g1 = make_geom(B, Hin, 1, inch, Hin - 2, 1, s2, 3, 3, 1, 0, 0, 0, 1);
            CK(conv_wgrad(e, g1, mn, dr1, r1, dw1, db1, rp2, Amax(), rs));
            return PROBAV_OK;
        }));
    }
    // upscale + reducers (models/modelsTF.py:152-164)
    const int nred = (int)e->iRed.size();
    float* cur = S + p.gA;
    float* oth = S + p.gB;
    unsigned* acur = new_slot();                    // amax slot of the tensor `cur` holds
    {
        const int h = p.redH[nred - 1], t = p.redT[nred - 1];
        const ConvGeom gu = make_geom(B, h, t, F, P, 1, s2, 3, 3, 3, 0, 0, 0, 0);
        float* const up_part = next_part();
        const float* const upx = W + p.red[nred - 1];
        float *const updy = S + p.dtail, *const updw = dweff(e->iUp), *const updb = dbias(e->iUp);
        CK(reduce_later(s, [=](hipStream_t rs) -> int {       // (only the weight-norm backward reads it; captured by value: it runs from a later flush)
            return conv_wgrad(e, gu, upx, updy, nullptr, updw, updb, up_part, Amax(), rs); }));
        CK(conv_fwd(e, bwd_data_geom(gu), S + p.dtail, nullptr, weffT(e->iUp), fragT(e->iUp), nullptr, nullptr, cur, amx(nullptr, e->iUp, acur), s));
    }
    for (int k = nred - 1; k >= 0; --k) {
        const probav_engine::RedSpec& rs = e->redSpec[k];
        const int refl = rs.refl;
        const int hi = k ? p.redH[k - 1] : Hin, ti = k ? p.redT[k - 1] : T;
        const float* xin = k ? W + p.red[k - 1] : W + p.act[R];
        const ConvGeom gr = red_geom(e, B, (size_t)k, hi, ti, F);
        // the backward-filter only feeds the weight-norm backward at the very end: side stream; every tensor it reads stays untouched
        // (each stage of the chain writes a buffer of its own)
        { Amax m; if (h3) { m.x = k ? A.red(k - 1) : A.act(R); m.w = acur; }
          CK(conv_wgrad(e, gr, xin, cur, W + p.red[k], dweff(e->iRed[k]), dbias(e->iRed[k]), next_part(), m, e->side_mode >= 2 ? reduce_fork(s) : s)); }
        float* outA = S + p.gred[2 * k];
        float* outB = S + p.gred[2 * k + 1];
        unsigned* aoth = new_slot();
        CK(conv_fwd(e, bwd_data_geom(gr), cur, W + p.red[k], weffT(e->iRed[k]), fragT(e->iRed[k]), nullptr, nullptr, outA, amx(acur, e->iRed[k], aoth), s));
        if (refl) {
            acur = new_slot();                      // the folded gradient is a new tensor
            if (rs.p == 1 && !rs.refl_t) CK(reflect_fold(outA, outB, B, hi, hi, ti * F, acur, s));
            else {
                CK(reflect_fold3(outA, outB, B, hi, hi, ti, F, rs.p, rs.p, rs.refl_t ? rs.pt : 0, s));
                if (h3) CK(amax_tensor(outB, (size_t)hi * hi * ti * F, B, acur, s));
            }
            cur = outB;
        } else {
            cur = outA;
            acur = aoth;
        }
    }
    oth = S + p.gB;                                 // (the unfused block path below ping-pongs between `cur` and this)
    // residual blocks (models/modelsTF.py:177-189), last first.  cur = d loss / d act[i+1]
    const ConvGeom ge = make_geom(B, Hin, T, F, Hin, T, E, 1, 1, 1, 0, 0, 0, 1);
    const ConvGeom gd = make_geom(B, Hin, T, E, Hin, T, D, 1, 1, 1, 0, 0, 0, 0);
    const ConvGeom gn = make_geom(B, Hin, T, D, Hin, T, F, 3, 3, 3, 1, 1, 0, 0);
    CK(reduce_flush(s));                                   // the residual path, the upscale layer's backward-filter, the reducers' slab sums: one fork
    for (int i = R - 1; i >= 0; --i) {
        float* gDec = S + p.gDec;
        float* Hbuf = S + p.Hb;
        float* dH = S + p.dH;
        const int le = e->iExp[i], ld = e->iDec[i], ln = e->iNorm[i];
        // normConv_i: d loss/d w, then d loss/d dec_i
        const bool fusedp = e->impl >= 1 && e->pw_mfma;
        { Amax m; if (h3) { m.x = A.dec(i); m.w = acur; }
          // (from the second block on, the last thing enqueued on s was the previous block's pointwise backward, whose slab sums forked right behind it)
          CK(conv_wgrad(e, gn, W + p.dec[i], cur, nullptr, dweff(ln), dbias(ln), next_part(), m,
                        (fusedp && e->side_mode >= 2) ? (i < R - 1 ? reduce_fork_adjacent(s) : reduce_fork(s)) : s)); }
        if (fusedp) oth = S + p.gblk[i];                   // this block's dX goes to its own buffer: `cur` stays intact for the late backward-filter
        unsigned* agdec = new_slot();
        CK(conv_fwd(e, bwd_data_geom(gn), cur, nullptr, weffT(ln), fragT(ln), nullptr, nullptr, gDec, amx(acur, ln, agdec), s));
        if (e->impl >= 1 && e->pw_mfma) {
            // fused: H recompute, dH, ReLU gate, dX (+ skip), dW1, dW2, db1, db2 -- nothing 256-wide touches HBM
            const long nvox = (long)B * Hin * Hin * T;
            unsigned* anew = new_slot();                     // amax slot of dX
            {   // (the class's bracket ends HERE: the flush below launches the batched slab sums on this stream, and they are no part of this class)
            ProfScope ps(e, e->impl >= 3 ? CLS_PW_BWD_DATA_X6 : CLS_PW_BWD_DATA, (double)nvox * (2.0 * F * E + 2.0 * E * D), s);   // bwd-data + bwd-filter of expConv and decConv; the recompute of H (F*E more) is not algorithmic work
            if (h3) {
                PwAmax m; m.x = A.act(i); m.w1 = A.w(le); m.w2 = A.w(ld); m.w1r = A.wrow(le); m.b1 = A.b(le); m.dt = agdec; m.y = anew;
                CK(x6_pw_backward(W + p.act[i], gDec, cur, Wpack + e->pkW1h[i], Wpack + e->pkW2Kh[i], Wpack + e->pkW1Ch[i],
                                  params + e->layers[le].wn.b_off, oth, dweff(le), dweff(ld), dbias(le), dbias(ld), next_part(), nvox, nvox / B, D, 2, m, s));
            } else if (e->impl >= 3)
                CK(x6_pw_backward(W + p.act[i], gDec, cur, Wpack + e->pkW1x6[i], Wpack + e->pkW2Kx6[i], Wpack + e->pkW1Cx6[i],
                                  params + e->layers[le].wn.b_off, oth, dweff(le), dweff(ld), dbias(le), dbias(ld), next_part(), nvox, nvox / B, D, 1, PwAmax(), s));
            else
                CK(mfma_pw_backward(W + p.act[i], gDec, cur, Wpack + e->pkW1[i], Wpack + e->pkW2B[i], Wpack + e->pkW1C[i],
                                    params + e->layers[le].wn.b_off, oth, dweff(le), dweff(ld), dbias(le), dbias(ld), next_part(), nvox, D, s));
            }
            float* tmp2 = cur; cur = oth; oth = tmp2;
            acur = anew;
            // what has been queued leaves every fourth block, and before the last one: the small launches to the side stream, the slab sums as ONE kernel on this stream
            // (round 5: slab_sum_later; one flush at the very end instead measured the same, +0.1 %)
            if (((R - 1 - i) & 3) == 3 || i == 1) CK(reduce_flush(s));
            continue;
        }
        // recompute H = relu(expConv_i(act[i])): the 256-channel tensor is never kept (1 KB/voxel/block)
        CK(conv_fwd(e, ge, W + p.act[i], nullptr, Wweff + e->layers[le].wn.w_off, Frags(), params + e->layers[le].wn.b_off, nullptr, Hbuf, Amax(), s));
        // decConv_i
        CK(conv_wgrad(e, gd, Hbuf, gDec, nullptr, dweff(ld), dbias(ld), next_part(), Amax(), s));
        CK(conv_fwd(e, bwd_data_geom(gd), gDec, nullptr, weffT(ld), fragT(ld), nullptr, nullptr, dH, Amax(), s));
        // expConv_i: ReLU gate (H > 0) applied where dH is consumed; skip path adds d loss/d act[i+1]
        CK(conv_wgrad(e, ge, W + p.act[i], dH, Hbuf, dweff(le), dbias(le), next_part(), Amax(), s));
        unsigned* aoth = new_slot();
        CK(conv_fwd(e, bwd_data_geom(ge), dH, Hbuf, weffT(le), fragT(le), nullptr, cur, oth, amx(nullptr, le, aoth), s));
        float* tmp = cur; cur = oth; oth = tmp;
        acur = aoth;
    }
    // mainConv1 (input-facing: no backward-data).  What is still queued leaves first, and this layer's own slab sum stays on the launch stream: forked, the
    // weight-norm backward would wait for an event record, a 10-us kernel on the side stream and the join's record -- 45 us of hole at the end of the pass.
    CK(reduce_flush(s));
    {
        ReduceSide* const ctx = side_guard.c;
        reduce_side_activate(nullptr);
        const int rc = conv_wgrad(e, make_geom(B, Hin, T, c.in_channels, Hin, T, F, 3, 3, 3, 1, 1, 0, 1), W + p.xn, cur, W + p.act[0],
                                  dweff(e->iMain), dbias(e->iMain), next_part(), Amax(), s);
        reduce_side_activate(ctx);
        if (rc) return rc;
    }
    CK(reduce_join(s));                                                       // every slab sum has landed in dweff / the bias gradients
    { ProfScope ps(e, CLS_WN, 0.0, s); CK(wn_backward(e->d_layers, (int)e->layers.size(), (int)e->cout_total, params, S + p.dweff2, Winvn, grads, s)); }
    return PROBAV_OK;
}

int probav_backward(probav_engine* e, const float* params, const float* dy, float* grads, void* ws, size_t ws_bytes, int B, void* stream)
{
    return backward_impl(e, params, dy, grads, ws, ws_bytes, nullptr, 0, B, nullptr, stream);
}
int probav_backward_split(probav_engine* e, const float* params, const float* dy, float* grads, const void* saved, size_t saved_bytes,
                          void* scratch, size_t scratch_bytes, int B, const void* wcache, size_t wcache_bytes, void* stream)
{
    if (!scratch) { set_error("probav_backward_split: null scratch", hipSuccess); return PROBAV_EINVAL; }
    if (wcache && (!e || wcache_bytes < make_wc_plan(e).total * sizeof(float))) { set_error("probav_backward_split: weight cache too small", hipSuccess); return PROBAV_EINVAL; }
    return backward_impl(e, params, dy, grads, saved, saved_bytes, scratch, scratch_bytes, B, (const float*)wcache, stream);
}
int probav_workspace_split(const probav_engine* e, int batch, size_t* saved_bytes, size_t* scratch_bytes)
{
    if (!e || batch < 1 || !saved_bytes || !scratch_bytes) { set_error("probav_workspace_split: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    const Plan p = make_plan(e, batch, 1);
    *saved_bytes = p.fwd_total * sizeof(float);
    *scratch_bytes = p.bwd_total * sizeof(float);
    return PROBAV_OK;
}
int probav_backward_wc(probav_engine* e, const float* params, const float* dy, float* grads, void* ws, size_t ws_bytes, int B,
                       const void* wcache, size_t wcache_bytes, void* stream)
{
    if (!e || !wcache || wcache_bytes < make_wc_plan(e).total * sizeof(float)) { set_error("probav_backward_wc: weight cache missing / too small", hipSuccess); return PROBAV_EINVAL; }
    return backward_impl(e, params, dy, grads, ws, ws_bytes, nullptr, 0, B, (const float*)wcache, stream);
}

size_t probav_weight_cache_bytes(const probav_engine* e) { return e ? make_wc_plan(e).total * sizeof(float) : 0; }

int probav_optimizer_step_fused(probav_engine* e, float* params, const float* grads, float* m, float* v, float lr, float beta1, float beta2,
                                float eps, float c_g, float c_m, float c_v, void* wcache, size_t wcache_bytes, void* stream)
{
    if (!e || !params || !grads || !m || !v || !wcache) { set_error("probav_optimizer_step_fused: null argument", hipSuccess); return PROBAV_EINVAL; }
    const WcPlan wc = make_wc_plan(e);
    if (wcache_bytes < wc.total * sizeof(float)) { set_error("probav_optimizer_step_fused: weight cache too small", hipSuccess); return PROBAV_ENOSPACE; }
    hipStream_t s = (hipStream_t)stream;
    float* C = (float*)wcache;
    unsigned* wam = reinterpret_cast<unsigned*>(C + wc.amax);
    if (hipMemsetAsync(wam, 0, (size_t)wc.n_wamax * sizeof(unsigned), s) != hipSuccess) { set_error("probav_optimizer_step_fused: amax reset", hipGetLastError()); return PROBAV_EHIP; }
    // one launch: the update of all 132 tensors + the weight normalisation of the updated parameters (+ the per-row maxima and the operand
    // packing of the next pass behind it): the next probav_forward_wc starts at the head kernel
    { ProfScope ps(e, CLS_WN, 0.0, s);
      CK(optimizer_wn_step(e->d_layers, (int)e->layers.size(), (int)e->cout_total, (int)e->cin_total, params, grads, m, v, lr, beta1, beta2, eps, c_g, c_m, c_v,
                           C + wc.weff, C + wc.weffT, C + wc.invn, wam, s)); }
    if (!e->jobs.empty()) { ProfScope ps(e, CLS_WN, 0.0, s); CK(mfma_pack(e->d_jobs, (int)e->jobs.size(), C + wc.weff, C + wc.weffT, C + wc.wpack, wam, s)); }
    return PROBAV_OK;
}


int probav_weight_cache_build(probav_engine* e, const float* params, void* wcache, size_t wcache_bytes, void* stream)
{
    if (!e || !params || !wcache) { set_error("probav_weight_cache_build: null argument", hipSuccess); return PROBAV_EINVAL; }
    const WcPlan wc = make_wc_plan(e);
    if (wcache_bytes < wc.total * sizeof(float)) { set_error("probav_weight_cache_build: weight cache too small", hipSuccess); return PROBAV_ENOSPACE; }
    hipStream_t s = (hipStream_t)stream;
    float* C = (float*)wcache;
    unsigned* wam = reinterpret_cast<unsigned*>(C + wc.amax);
    if (hipMemsetAsync(wam, 0, (size_t)wc.n_wamax * sizeof(unsigned), s) != hipSuccess) { set_error("probav_weight_cache_build: amax reset", hipGetLastError()); return PROBAV_EHIP; }
    // exactly what a forward pass without a cache does first: weight normalisation (+ per-column / per-row maxima) and operand packing
    { ProfScope ps(e, CLS_WN, 0.0, s);
      CK(wn_forward(e->d_layers, (int)e->layers.size(), (int)e->cout_total, (int)e->cin_total, params, C + wc.weff, C + wc.weffT, C + wc.invn, wam, s)); }
    if (!e->jobs.empty()) { ProfScope ps(e, CLS_WN, 0.0, s); CK(mfma_pack(e->d_jobs, (int)e->jobs.size(), C + wc.weff, C + wc.weffT, C + wc.wpack, wam, s)); }
    return PROBAV_OK;
}

// ---- introspection (parity tests) -----------------------------------------------------------------
int probav_workspace_view(const probav_engine* e, int batch, int training, int kind, int index, int64_t* offset_floats, int64_t* count)
{
    if (!e || batch < 1 || !offset_floats || !count) { set_error("probav_workspace_view: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    const Plan p = make_plan(e, batch, training);
    const probav_net_cfg& c = e->cfg;
    const int64_t V = (int64_t)batch * e->Hin * e->Hin * c.num_img_lr;
    const int R = c.num_res_blocks, nred = (int)e->iRed.size(), s2 = c.scale * c.scale;
    switch (kind) {
    case PROBAV_VIEW_ACT: if (index < 0 || index > R) break; *offset_floats = (int64_t)p.act[index]; *count = V * c.num_filters; return PROBAV_OK;
    case PROBAV_VIEW_DEC: if (index < 0 || index >= R) break; *offset_floats = (int64_t)p.dec[index]; *count = V * c.dec_channels; return PROBAV_OK;
    case PROBAV_VIEW_RED: if (index < 0 || index >= nred) break;
        *offset_floats = (int64_t)p.red[index]; *count = (int64_t)batch * p.redH[index] * p.redH[index] * p.redT[index] * c.num_filters; return PROBAV_OK;
    case PROBAV_VIEW_RESID1: if (index != 0) break; *offset_floats = (int64_t)p.r1; *count = (int64_t)batch * (e->Hin - 2) * (e->Hin - 2) * s2; return PROBAV_OK;
    default: break;
    }
    set_error("probav_workspace_view: no such tensor", hipSuccess);
    return PROBAV_EINVAL;
}

int probav_debug_hidden(probav_engine* e, const float* params, const void* ws, size_t ws_bytes, int B, int block, float* hidden, float* dec_scratch, const void* wcache, void* stream)
{
    if (!e || !params || !ws || !hidden || !dec_scratch || B < 1 || block < 0 || block >= e->cfg.num_res_blocks) { set_error("probav_debug_hidden: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    if (e->impl < 3 || !e->pw_mfma) { set_error("probav_debug_hidden: only the split-operand kernel families (impl 3, 4) expose their hidden tile", hipSuccess); return PROBAV_EINVAL; }
    const Plan p = make_plan(e, B, 1);
    if (ws_bytes < p.fwd_total * sizeof(float)) { set_error("probav_debug_hidden: workspace too small", hipSuccess); return PROBAV_ENOSPACE; }
    const float* W = (const float*)ws;
    const probav_net_cfg& c = e->cfg;
    const int D = c.dec_channels, R = c.num_res_blocks, i = block;
    const long nvox = (long)B * e->Hin * e->Hin * c.num_img_lr;
    const WcPlan wc = make_wc_plan(e);
    const float* WC = (const float*)wcache;                           // the forward pass ran from the weight cache: its fragments and weight slots live there
    const float* Wpack = WC ? WC + wc.wpack : W + p.wpack;
    const AmaxSlots A(e, p, W, R, WC ? reinterpret_cast<unsigned*>(const_cast<float*>(WC + wc.amax)) : nullptr);
    // the forward launch of block i again, into the caller's scratch output (the saved state is only read), with the hidden tile written out;
    // no amax report (the slots of the saved tensors stay as the forward pass left them)
    PwAmax m;
    const bool h3 = e->impl >= 4;
    if (h3) { m.x = A.act(i); m.w1 = A.w(e->iExp[i]); m.w2 = A.w(e->iDec[i]); m.w2c = A.wcol(e->iDec[i]); m.b1 = A.b(e->iExp[i]); }
    return x6_pw_forward(W + p.act[i], Wpack + (h3 ? e->pkW1h[i] : e->pkW1x6[i]), Wpack + (h3 ? e->pkW2h[i] : e->pkW2x6[i]),
                         params + e->layers[e->iExp[i]].wn.b_off, params + e->layers[e->iDec[i]].wn.b_off, dec_scratch, nvox, nvox / B, D, h3 ? 2 : 1, m,
                         (hipStream_t)stream, hidden);
}

int probav_debug_hidden_from_forward_kernel(int on) { x6_pw_dump_from_forward_kernel(on); return PROBAV_OK; }

// ---- single operators ---------------------------------------------------------------------------
static ConvGeom geom_from(const int32_t a[17])
{
    ConvGeom g;
    g.N = a[0]; g.Hi = a[1]; g.Wi = a[2]; g.Ti = a[3]; g.Cin = a[4]; g.Ho = a[5]; g.Wo = a[6]; g.To = a[7]; g.Cout = a[8];
    g.kh = a[9]; g.kw = a[10]; g.kt = a[11]; g.ph = a[12]; g.pw = a[13]; g.pt = a[14]; g.reflect_hw = a[15]; g.relu = a[16]; g.reflect_t = 0;
    return g;
}
static bool geom_ok(const ConvGeom& g)
{
    if (g.N < 1 || g.Cin < 1 || g.Cout < 1 || g.kh < 1 || g.kw < 1 || g.kt < 1) return false;
    if (g.Ho < 1 || g.Wo < 1 || g.To < 1 || g.Hi < 1 || g.Wi < 1 || g.Ti < 1) return false;
    if (g.ph < 0 || g.pw < 0 || g.pt < 0) return false;
    if (g.reflect_hw && (g.ph >= g.Hi || g.pw >= g.Wi || g.kh - 1 - g.ph >= g.Hi || g.kw - 1 - g.pw >= g.Wi)) return false;
    return true;
}

// scratch owned by the library for the single-operator entry points (parity tests): packing raw Keras-layout
// weights into MFMA fragments needs a device buffer, allocated on first use -- the engine path never does this.
static float* g_op_frag = nullptr;
static PackJob* g_op_job = nullptr;
static unsigned* g_op_amax = nullptr;    // amax slots of a single-operator call with H3 arithmetic (the engine gets them from the producing kernels)
static size_t g_op_amax_cap = 0;
static int op_scratch()
{
    if (g_op_frag) return PROBAV_OK;
    hipError_t err = hipMalloc((void**)&g_op_frag, (size_t)4 << 20);
    if (err == hipSuccess) err = hipMalloc((void**)&g_op_job, 4 * sizeof(PackJob));
    if (err != hipSuccess) { set_error("single-operator scratch allocation", err); return PROBAV_EHIP; }
    return PROBAV_OK;
}
// `count` zeroed slots in g_op_amax
static int op_amax_reserve(size_t count, hipStream_t s)
{
    if (count > g_op_amax_cap) {
        hipError_t err = hipStreamSynchronize(s);
        if (err == hipSuccess && g_op_amax) err = hipFree(g_op_amax);
        g_op_amax = nullptr; g_op_amax_cap = 0;
        if (err == hipSuccess) err = hipMalloc((void**)&g_op_amax, (count + 1024) * sizeof(unsigned));
        if (err != hipSuccess) { set_error("single-operator amax allocation", err); return PROBAV_EHIP; }
        g_op_amax_cap = count + 1024;
    }
    if (hipMemsetAsync(g_op_amax, 0, count * sizeof(unsigned), s) != hipSuccess) { set_error("single-operator amax reset", hipGetLastError()); return PROBAV_EHIP; }
    return PROBAV_OK;
}
// convolution operands: slots [0, N) = x per sample, [N, 2N) = output / dY per sample, [2N, 2N + cols) = filter per output column
static int op_amax_conv(const ConvGeom& g, const float* x, const float* w, const float* dy, hipStream_t s)
{
    int rc = op_amax_reserve((size_t)2 * g.N + 256, s);
    if (!rc) rc = amax_tensor(x, (size_t)g.Hi * g.Wi * g.Ti * g.Cin, g.N, g_op_amax, s);
    if (!rc && dy) rc = amax_tensor(dy, (size_t)g.Ho * g.Wo * g.To * g.Cout, g.N, g_op_amax + g.N, s);
    if (!rc && w) rc = amax_columns(w, (long)g.kh * g.kw * g.kt * g.Cin, g.Cout, g_op_amax + 2 * g.N, s);
    return rc;
}
static int op_pack(const ConvGeom& g, const float* w, hipStream_t s, int split = 0, bool per_tap = false)   // split: 0 fp32 fragments, 1 X6, 2 H3
{
    const size_t n = split ? (size_t)X6_CONV_FRAG_WORDS : mfma_conv_wfrag_floats(g.Cin, g.Cout);
    if (n == 0) { set_error("probav_conv3d_forward: channel configuration not supported by the MFMA kernel", hipSuccess); return PROBAV_EINVAL; }
    int rc = op_scratch();
    if (rc) return rc;
    if (n * sizeof(float) > ((size_t)4 << 20)) { set_error("probav_conv3d_forward: fragment scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
    PackJob J; memset(&J, 0, sizeof(J));
    if (split) {
        const bool kc = g.Cin == 25 && !per_tap;                            // K-concatenated form (per_tap: the H3 piece-ring strip kernel)
        J.type = kc ? PACK_X6_CONVK : PACK_X6_CONV; J.count = kc ? X6_CONVK_FRAG_WORDS : X6_CONV_FRAG_WORDS; J.Cin = g.Cin; J.Cout = g.Cout; J.taps = 27;
        if (split == 2) { J.type += 10; J.count = kc ? H3_CONVK_FRAG_WORDS : H3_CONV_FRAG_WORDS; J.amax_percol = 1; J.ncol = g.Cout; J.amax_slot = 2 * g.N; }
        if (split == 2 && g.Cin == 25 && per_tap) { J.type = PACK_H3_CONVP; J.count = H3_CONVK_FRAG_WORDS; }
    } else mfma_conv_pack_job(J, g.Cin, g.Cout);
    hipError_t err = hipStreamSynchronize(s);
    if (err == hipSuccess) err = hipMemcpy(g_op_job, &J, sizeof(J), hipMemcpyHostToDevice);
    if (err != hipSuccess) { set_error("probav_conv3d_forward: job upload", err); return PROBAV_EHIP; }
    return mfma_pack(g_op_job, 1, w, w, g_op_frag, g_op_amax, s);
}

int probav_conv3d_forward(const int32_t geom[17], const float* x, const float* gate, const float* w, const float* bias,
                          const float* skip, float* y, int impl, void* stream)
{
    if (!geom || !x || !w || !y) { set_error("probav_conv3d_forward: null argument", hipSuccess); return PROBAV_EINVAL; }
    const ConvGeom g = geom_from(geom);
    if (!geom_ok(g)) { set_error("probav_conv3d_forward: bad geometry", hipSuccess); return PROBAV_EINVAL; }
    if (impl < 0 || impl > 4) { set_error("probav_conv3d_forward: impl must be 0..4", hipSuccess); return PROBAV_EINVAL; }
    // (what the engine does for these two geometries in every kernel family but 0)
    if (impl >= 1 && !gate && !skip && bias && conv3d_up_forward_supported(g)) return conv3d_up_forward(g, x, w, bias, y, (hipStream_t)stream);
    if (impl >= 1 && !gate && !skip && !g.relu && conv3d_up_bwd_data_supported(g)) return conv3d_up_bwd_data(g, x, w, bias, y, nullptr, (hipStream_t)stream);
    if (impl >= 1) {
        const bool pstrip = impl == 4 && x6_strip_wants_tap_fragments(g, 2);
        const bool x6row = impl >= 3 && !mfma_conv_strip_supported(g) && !pstrip && x6_conv_rowtile_supported(g);
        const bool okk = x6row || pstrip || (impl >= 2 ? mfma_conv_strip_supported(g) : mfma_conv_supported(g));
        if (!okk) { set_error("probav_conv3d_forward: geometry not supported by this MFMA kernel", hipSuccess); return PROBAV_EINVAL; }
        int rc = op_scratch();
        if (rc) return rc;
        Amax am;
        if (impl == 4) {
            if (g.Cout > 256) { set_error("probav_conv3d_forward: Cout > 256", hipSuccess); return PROBAV_EINVAL; }
            rc = op_amax_conv(g, x, w, nullptr, (hipStream_t)stream);
            if (rc) return rc;
            am.x = g_op_amax; am.w = g_op_amax + 2 * g.N; am.y = g_op_amax + g.N;
        }
        rc = op_pack(g, w, (hipStream_t)stream, impl >= 3 ? impl - 2 : 0, pstrip);
        if (rc) return rc;
        if (x6row) return x6_conv_rowtile_forward(g, x, gate, g_op_frag, bias, skip, y, impl - 2, am, (hipStream_t)stream);
        if (impl >= 3) return x6_conv_strip_forward(g, x, gate, g_op_frag, bias, skip, y, impl - 2, am, (hipStream_t)stream);
        if (impl == 2) return mfma_conv_strip_forward(g, x, gate, g_op_frag, bias, skip, y, am, (hipStream_t)stream);
        return mfma_conv_forward(g, x, gate, g_op_frag, bias, skip, y, am, (hipStream_t)stream);
    }
    return conv3d_direct_forward(g, x, gate, w, bias, skip, y, (hipStream_t)stream);
}

size_t probav_conv3d_wgrad_scratch_bytes(const int32_t geom[17], int impl)
{
    if (!geom) return 0;
    const ConvGeom g = geom_from(geom);
    if (impl == 3 || impl == 4) return x6_wgrad_supported(g) ? x6_wgrad_partial_floats(g) * sizeof(float) : 0;
    if (impl == 1) return mfma_wgrad_supported(g) ? mfma_wgrad_partial_floats(g) * sizeof(float) : 0;
    return wgrad_partial_floats(g) * sizeof(float);
}

int probav_conv3d_wgrad(const int32_t geom[17], const float* x, const float* dy, const float* gate, float* dw, float* db,
                        void* scratch, size_t scratch_bytes, int impl, void* stream)
{
    if (!geom || !x || !dy || !dw || !scratch) { set_error("probav_conv3d_wgrad: null argument", hipSuccess); return PROBAV_EINVAL; }
    const ConvGeom g = geom_from(geom);
    if (!geom_ok(g)) { set_error("probav_conv3d_wgrad: bad geometry", hipSuccess); return PROBAV_EINVAL; }
    if (impl == 3 || impl == 4) {
        if (!x6_wgrad_supported(g)) { set_error("probav_conv3d_wgrad: geometry not supported by the x6 kernel", hipSuccess); return PROBAV_EINVAL; }
        if (scratch_bytes < x6_wgrad_partial_floats(g) * sizeof(float)) { set_error("probav_conv3d_wgrad: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
        Amax am;
        if (impl == 4) {
            int rc = op_scratch();
            if (!rc) rc = op_amax_conv(g, x, nullptr, dy, (hipStream_t)stream);
            if (rc) return rc;
            am.x = g_op_amax; am.w = g_op_amax + g.N;
        }
        return x6_conv_wgrad(g, x, dy, gate, dw, db, (float*)scratch, impl == 4 ? 2 : 1, am, (hipStream_t)stream);
    }
    if (impl == 1) {
        if (!mfma_wgrad_supported(g)) { set_error("probav_conv3d_wgrad: geometry not supported by the MFMA kernel", hipSuccess); return PROBAV_EINVAL; }
        if (scratch_bytes < mfma_wgrad_partial_floats(g) * sizeof(float)) { set_error("probav_conv3d_wgrad: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
        return mfma_conv_wgrad(g, x, dy, gate, dw, db, (float*)scratch, (hipStream_t)stream);
    }
    if (scratch_bytes < wgrad_partial_floats(g) * sizeof(float)) { set_error("probav_conv3d_wgrad: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
    return conv3d_direct_wgrad(g, x, dy, gate, dw, db, (float*)scratch, (hipStream_t)stream);
}

// fused pointwise pair, single-operator form (packs the Keras-layout weights into MFMA fragments in library scratch)
static int op_pack_pw(const float* w1, const float* w2, int D, hipStream_t s, const float** f1, const float** f2, const float** f2b, const float** f1c)
{
    { int rc0 = op_scratch(); if (rc0) return rc0; }
    static PackJob* d_jobs4 = nullptr;
    if (!d_jobs4) { hipError_t err = hipMalloc((void**)&d_jobs4, 4 * sizeof(PackJob)); if (err != hipSuccess) { set_error("probav_pw: job allocation", err); return PROBAV_EHIP; } }
    PackJob J[4]; memset(J, 0, sizeof(J));
    const int types[4] = {PACK_PW_A_KCIN, PACK_PW_A_KHCH, PACK_PW_A_KOUT, PACK_PW_A_CIN_KHCH};
    for (int k = 0; k < 4; ++k) {
        J[k].type = types[k]; J[k].count = 8192; J[k].dst_off = 262144 + 8192 * k;       // behind the conv fragment area
        const bool from_w1 = (k == 0 || k == 3);
        J[k].src_is_T = from_w1 ? 0 : 1;                                                  // weff := w1, weffT := w2
        J[k].Cin = from_w1 ? 32 : 256; J[k].Cout = from_w1 ? 256 : D;
    }
    hipError_t err = hipStreamSynchronize(s);
    if (err == hipSuccess) err = hipMemcpy(d_jobs4, J, sizeof(J), hipMemcpyHostToDevice);
    if (err != hipSuccess) { set_error("probav_pw: job upload", err); return PROBAV_EHIP; }
    *f1 = g_op_frag + J[0].dst_off; *f2 = g_op_frag + J[1].dst_off; *f2b = g_op_frag + J[2].dst_off; *f1c = g_op_frag + J[3].dst_off;
    return mfma_pack(d_jobs4, 4, w1, w2, g_op_frag, nullptr, s);
}

// h3: PACK_H3_* fragments; the weights' amax must already be in the slots op_amax_pw lays out (wbase = index of the first weight slot)
static int op_pack_pw_x6(const float* w1, const float* w2, int D, hipStream_t s, const float** f1, const float** f2,
                         const float** f2k = nullptr, const float** f1c = nullptr, bool h3 = false, int wbase = 0)
{
    static float* frag = nullptr;
    static PackJob* d_jobs = nullptr;
    if (!frag) {
        hipError_t err = hipMalloc((void**)&frag, (size_t)4 * X6_PW_FRAG_WORDS * 4);
        if (err == hipSuccess) err = hipMalloc((void**)&d_jobs, 4 * sizeof(PackJob));
        if (err != hipSuccess) { set_error("probav_pw (x6): scratch allocation", err); return PROBAV_EHIP; }
    }
    PackJob J[4]; memset(J, 0, sizeof(J));
    J[2].type = PACK_X6_PW_W2K; J[2].src_is_T = 1; J[2].dst_off = 2 * X6_PW_FRAG_WORDS; J[2].count = X6_PW_FRAG_WORDS; J[2].Cin = 256; J[2].Cout = D;
    J[3].type = PACK_X6_PW_W1C; J[3].src_is_T = 0; J[3].dst_off = 3 * X6_PW_FRAG_WORDS; J[3].count = X6_PW_FRAG_WORDS; J[3].Cin = 32; J[3].Cout = 256;
    J[0].type = PACK_X6_PW_W1; J[0].src_is_T = 0; J[0].dst_off = 0; J[0].count = X6_PW_FRAG_WORDS; J[0].Cin = 32; J[0].Cout = 256;
    J[1].type = PACK_X6_PW_W2; J[1].src_is_T = 1; J[1].dst_off = X6_PW_FRAG_WORDS; J[1].count = X6_PW_FRAG_WORDS; J[1].Cin = 256; J[1].Cout = D;
    if (h3) {
        for (int k = 0; k < 4; ++k) { J[k].type += 10; J[k].count = H3_PW_FRAG_WORDS; }
        J[0].amax_slot = wbase + 0;                                              // W1 as the operand of (a): one scale
        J[2].amax_slot = wbase + 1;                                              // W2 as the operand of (b): one scale
        J[1].amax_percol = 1; J[1].ncol = D; J[1].amax_slot = wbase + 8;         // W2 forward: per output column d
        J[3].amax_percol = 1; J[3].ncol = 32; J[3].amax_slot = wbase + 40;       // W1 as the operand of (c): per cin row
    }
    hipError_t err = hipStreamSynchronize(s);
    if (err == hipSuccess) err = hipMemcpy(d_jobs, J, sizeof(J), hipMemcpyHostToDevice);
    if (err != hipSuccess) { set_error("probav_pw (x6): job upload", err); return PROBAV_EHIP; }
    *f1 = frag; *f2 = frag + X6_PW_FRAG_WORDS;
    if (f2k) *f2k = frag + 2 * X6_PW_FRAG_WORDS;
    if (f1c) *f1c = frag + 3 * X6_PW_FRAG_WORDS;
    return mfma_pack(d_jobs, 4, w1, w2, frag, g_op_amax, s);
}
// amax of the operands of a single-operator call of the fused pointwise pair, ns samples: [0, ns) x, [ns, 2ns) d_dec, [2ns, 3ns) output;
// wbase = 3 ns: +0 w1, +1 w2, +2 b1 (whole tensors), +8 .. w2 per output column, +40 .. w1 per input row
static int op_amax_pw(const float* x, const float* w1, const float* w2, const float* b1, const float* d_dec, long nvox, long vps, int D, hipStream_t s, PwAmax& m)
{
    int rc = op_scratch();
    if (rc) return rc;
    const int ns = (int)(nvox / vps), wb = 3 * ns;
    rc = op_amax_reserve((size_t)wb + 80, s);
    if (!rc) rc = amax_tensor(x, (size_t)vps * 32, ns, g_op_amax, s);
    if (!rc && d_dec) rc = amax_tensor(d_dec, (size_t)vps * D, ns, g_op_amax + ns, s);
    if (!rc) rc = amax_tensor(w1, 32 * 256, 1, g_op_amax + wb + 0, s);
    if (!rc) rc = amax_tensor(w2, (size_t)256 * D, 1, g_op_amax + wb + 1, s);
    if (!rc) rc = amax_tensor(b1, 256, 1, g_op_amax + wb + 2, s);
    if (!rc) rc = amax_columns(w2, 256, D, g_op_amax + wb + 8, s);
    if (!rc) rc = amax_tensor(w1, 256, 32, g_op_amax + wb + 40, s);              // rows of W1 [32][256]
    m.x = g_op_amax; m.dt = g_op_amax + ns; m.y = g_op_amax + 2 * ns;
    m.w1 = g_op_amax + wb; m.w2 = g_op_amax + wb + 1; m.b1 = g_op_amax + wb + 2; m.w2c = g_op_amax + wb + 8; m.w1r = g_op_amax + wb + 40;
    return rc;
}

int probav_pw_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* dec,
                      int64_t nvox, int64_t vox_per_sample, int D, int impl, void* stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !dec || nvox < 1 || impl < 2 || impl > 4) { set_error("probav_pw_forward: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    if (!mfma_pw_supported(32, 256, D)) { set_error("probav_pw_forward: needs F=32, E=256, D<=26", hipSuccess); return PROBAV_EINVAL; }
    long vps = vox_per_sample > 0 ? (long)vox_per_sample : (long)nvox;
    if (nvox % vps) { set_error("probav_pw_forward: nvox is not a multiple of vox_per_sample", hipSuccess); return PROBAV_EINVAL; }
    if (impl >= 3) {
        const float *g1, *g2;
        PwAmax am;
        if (impl == 4) {
            int rc = op_amax_pw(x, w1, w2, b1, nullptr, (long)nvox, vps, D, (hipStream_t)stream, am);
            if (rc) return rc;
        }
        int rc = op_pack_pw_x6(w1, w2, D, (hipStream_t)stream, &g1, &g2, nullptr, nullptr, impl == 4, 3 * (int)(nvox / vps));
        if (rc) return rc;
        return x6_pw_forward(x, g1, g2, b1, b2, dec, (long)nvox, vps, D, impl - 2, am, (hipStream_t)stream);
    }
    const float *f1, *f2, *f2b, *f1c;
    int rc = op_pack_pw(w1, w2, D, (hipStream_t)stream, &f1, &f2, &f2b, &f1c);
    if (rc) return rc;
    return mfma_pw_forward(x, f1, f2, b1, b2, dec, (long)nvox, D, (hipStream_t)stream);
}

size_t probav_pw_backward_scratch_bytes(int D) { return mfma_pw_backward_slab_floats(D) * sizeof(float); }

int probav_pw_backward(const float* x, const float* d_dec, const float* d_skip, const float* w1, const float* b1, const float* w2,
                       float* dx, float* dw1, float* db1, float* dw2, float* db2, void* scratch, size_t scratch_bytes,
                       int64_t nvox, int64_t vox_per_sample, int D, int impl, void* stream)
{
    if (!x || !d_dec || !d_skip || !w1 || !b1 || !w2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !scratch || nvox < 1 || impl < 2 || impl > 4) {
        set_error("probav_pw_backward: null/invalid argument", hipSuccess); return PROBAV_EINVAL;
    }
    if (!mfma_pw_supported(32, 256, D)) { set_error("probav_pw_backward: needs F=32, E=256, D<=26", hipSuccess); return PROBAV_EINVAL; }
    if (scratch_bytes < probav_pw_backward_scratch_bytes(D)) { set_error("probav_pw_backward: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
    long vps = vox_per_sample > 0 ? (long)vox_per_sample : (long)nvox;
    if (nvox % vps) { set_error("probav_pw_backward: nvox is not a multiple of vox_per_sample", hipSuccess); return PROBAV_EINVAL; }
    const float *f1, *f2, *f2b, *f1c;
    if (impl >= 3) {
        PwAmax am;
        if (impl == 4) {
            int rc = op_amax_pw(x, w1, w2, b1, d_dec, (long)nvox, vps, D, (hipStream_t)stream, am);
            if (rc) return rc;
        }
        int rc = op_pack_pw_x6(w1, w2, D, (hipStream_t)stream, &f1, &f2, &f2b, &f1c, impl == 4, 3 * (int)(nvox / vps));
        if (rc) return rc;
        return x6_pw_backward(x, d_dec, d_skip, f1, f2b, f1c, b1, dx, dw1, dw2, db1, db2, (float*)scratch, (long)nvox, vps, D, impl - 2, am, (hipStream_t)stream);
    }
    int rc = op_pack_pw(w1, w2, D, (hipStream_t)stream, &f1, &f2, &f2b, &f1c);
    if (rc) return rc;
    return mfma_pw_backward(x, d_dec, d_skip, f1, f2b, f1c, b1, dx, dw1, dw2, db1, db2, (float*)scratch, (long)nvox, D, (hipStream_t)stream);
}

int probav_wn_forward(probav_engine* e, const float* params, float* weff, float* weffT, float* inv_norm, void* stream)
{
    if (!e || !params || !weff || !weffT || !inv_norm) { set_error("probav_wn_forward: null argument", hipSuccess); return PROBAV_EINVAL; }
    return wn_forward(e->d_layers, (int)e->layers.size(), (int)e->cout_total, (int)e->cin_total, params, weff, weffT, inv_norm, nullptr, (hipStream_t)stream);
}
int probav_wn_backward(probav_engine* e, const float* params, const float* dweff, const float* inv_norm, float* grads, void* stream)
{
    if (!e || !params || !dweff || !inv_norm || !grads) { set_error("probav_wn_backward: null argument", hipSuccess); return PROBAV_EINVAL; }
    return wn_backward(e->d_layers, (int)e->layers.size(), (int)e->cout_total, params, dweff, inv_norm, grads, (hipStream_t)stream);
}

int probav_shift_loss_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size, int border,
                              int bit_depth, float* l1, float* l2, float* cpsnr, int32_t* arg_l1, int32_t* arg_l2,
                              float* mean_l1, float* mean_l2, void* stream)
{
    if (!hr || !mask || !pred || !l1 || !l2 || !cpsnr || !arg_l1 || !arg_l2 || !mean_l1 || !mean_l2) {
        set_error("probav_shift_loss_forward: null argument", hipSuccess); return PROBAV_EINVAL;
    }
    const float maxv = (float)((1u << bit_depth) - 1u);        // Losses.numBytes = 2**bitDepth - 1 (models/loss.py:19)
    return shift_loss_forward(hr, mask, pred, batch, size, border, l1, l2, cpsnr, arg_l1, arg_l2, mean_l1, mean_l2, maxv, (hipStream_t)stream);
}
int probav_shift_loss_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg, int batch, int size,
                               int border, int which, const float* upstream, float* dpred, void* stream)
{
    if (!hr || !mask || !pred || !arg || !dpred || batch < 1) { set_error("probav_shift_loss_backward: null argument", hipSuccess); return PROBAV_EINVAL; }
    return shift_loss_backward(hr, mask, pred, arg, batch, size, border, which, upstream, dpred, (hipStream_t)stream);
}
int probav_shift_l1edge_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size, int border, float pi,
                                float* loss, int32_t* arg, float* mean, void* stream)
{
    if (!hr || !mask || !pred || !loss || !arg || !mean || batch < 1) { set_error("probav_shift_l1edge_forward: null argument", hipSuccess); return PROBAV_EINVAL; }
    return shift_l1edge_forward(hr, mask, pred, batch, size, border, pi, loss, arg, mean, mean + 1, (hipStream_t)stream);
}
int probav_shift_l1edge_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg, int batch, int size,
                                 int border, float pi, const float* upstream, float* dpred, void* stream)
{
    if (!hr || !mask || !pred || !arg || !dpred || batch < 1) { set_error("probav_shift_l1edge_backward: null argument", hipSuccess); return PROBAV_EINVAL; }
    return shift_l1edge_backward(hr, mask, pred, arg, batch, size, border, pi, upstream, dpred, (hipStream_t)stream);
}
size_t probav_revssim_scratch_bytes(int batch, int border) { return batch > 0 && border >= 0 ? revssim_scratch_bytes(batch, border) : 0; }
int probav_revssim_forward(const float* hr, const uint8_t* mask, const float* pred, int batch, int size, int border, int bit_depth,
                           float eta, void* scratch, size_t scratch_bytes, float* loss, int32_t* arg, void* stream)
{
    if (!hr || !mask || !pred || !scratch || !loss || !arg || batch < 1) { set_error("probav_revssim_forward: null argument", hipSuccess); return PROBAV_EINVAL; }
    if (scratch_bytes < revssim_scratch_bytes(batch, border)) { set_error("probav_revssim_forward: scratch too small", hipSuccess); return PROBAV_ENOSPACE; }
    const float maxv = (float)((1u << bit_depth) - 1u);
    return revssim_forward(hr, mask, pred, batch, size, border, maxv, eta, (double*)scratch, loss, arg, (hipStream_t)stream);
}
int probav_revssim_backward(const float* hr, const uint8_t* mask, const float* pred, const int32_t* arg, const void* scratch, int batch,
                            int size, int border, int bit_depth, float eta, const float* upstream, float* dpred, void* stream)
{
    if (!hr || !mask || !pred || !arg || !scratch || !dpred || batch < 1) { set_error("probav_revssim_backward: null argument", hipSuccess); return PROBAV_EINVAL; }
    const float maxv = (float)((1u << bit_depth) - 1u);
    return revssim_backward(hr, mask, pred, arg, (const double*)scratch, batch, size, border, maxv, eta, upstream, dpred, (hipStream_t)stream);
}
int probav_nadam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float c_g, float c_m, float c_v, void* stream)
{
    if (!params || !grads || !m || !v || n < 0) { set_error("probav_nadam_step: null/invalid argument", hipSuccess); return PROBAV_EINVAL; }
    return nadam_step(params, grads, m, v, (long)n, lr, beta1, beta2, eps, c_g, c_m, c_v, (hipStream_t)stream);
}

int probav_clip_round(const float* in, float* out, size_t n, float lo, float hi, void* stream)
{
    if (!in || !out) { set_error("probav_clip_round: null argument", hipSuccess); return PROBAV_EINVAL; }
    if (n == 0) return PROBAV_OK;
    return clip_round(in, out, n, lo, hi, (hipStream_t)stream);
}

}  // extern "C"
