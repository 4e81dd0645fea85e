// Device helpers of the split-operand kernels (kernels_x6.hip, and the split variants inside kernels_mfma.hip).
//
// Two arithmetics evaluate an fp32 product on the 16-bit matrix pipe; a kernel is written once against the `AR` tag:
//   X6  three bf16 truncation pieces per value, the six largest piece products (v_mfma_f32_32x32x16_bf16).  No scaling.
//   H3  two fp16 round-to-nearest pieces per value (a = a0 + a1 to 2^-24 |a|: the sign of a1 is the 23rd bit), three piece
//       products a1 b0 + a0 b1 + a0 b0 (v_mfma_f32_32x32x16_f16; dropped a1 b1 <= 2^-24 |ab|) -- half the MFMAs of X6.  fp16 has
//       5 exponent bits, so every operand is multiplied by a power of two that puts the largest magnitude of its SCALING GROUP (or
//       a bound on it) into [2^14, 2^15); the product is scaled back once, in the epilogue.  A scale may vary along an operand's
//       free index but not along the contracted one, so the groups are: one SAMPLE (patch) of an activation / gradient tensor
//       (samples never meet in a forward or backward-data contraction: a patch's result does not depend on its batch mates, bit
//       for bit) and one OUTPUT COLUMN of a filter matrix.  Elements down to 2^-17 of their group's maximum keep full relative
//       precision, smaller ones an absolute error of 2^-40 of that maximum.
#pragma once
#include <hip/hip_runtime.h>

namespace probav {

// per-wave phase stamps for tools/diag_x6.hip (diagnostic build only: -DPROBAV_STAMP; g_stamps lives in kernels_mfma.hip)
#ifdef PROBAV_STAMP2
#define XS2(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); xs2[k] += t_ - xs2t; xs2t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define XS2(k) do { } while (0)
#endif
#ifdef PROBAV_STAMP
#define XS_ENTRY do { } while (0)
#define XS_DECL unsigned long long xs_t = __builtin_amdgcn_s_memtime(), xs_acc[8] = {xs_t, 0, 0, 0, 0, 0, 0, 0}
#define XS_ACC(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); xs_acc[k] += t_ - xs_t; xs_t = t_; } while (0)
#define XS_OUT do { xs_acc[7] = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) for (int k_ = 0; k_ < 8; ++k_) g_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + k_] = xs_acc[k_]; } while (0)
#elif defined(PROBAV_STAMP_CLOCK)
// clock-only build (tools/kbench.hip -DPROBAV_STAMP_CLOCK): two stamps per wave, none inside the loops; in-kernel clock = cycles / (100 MHz ticks) * 0.1 GHz
// XS_ENTRY (first statement of a kernel): the wave's arrival, so that a launch can be taken apart -- dispatch ramp, prologue (entry -> XS_DECL), loop, tail.
// Slots per wave: [0] cycles and [1] 100-MHz ticks between XS_DECL and XS_OUT, [2] entry, [3] XS_DECL, [4] XS_OUT (absolute 100-MHz ticks), [5] XCC_ID << 32 | HW_ID (where the wave ran)
static constexpr unsigned long long xs_re = 0;            // (kernels without an XS_ENTRY: the local one shadows this)
#define XS_ENTRY const unsigned long long xs_re = __builtin_amdgcn_s_memrealtime()
#define XS_DECL const unsigned long long xs_c0 = __builtin_amdgcn_s_memtime(), xs_r0 = __builtin_amdgcn_s_memrealtime()
#define XS_ACC(k) do { } while (0)
#define XS_OUT do { const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) { unsigned long long* o_ = g_stamps + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; o_[0] = c1_ - xs_c0; o_[1] = r1_ - xs_r0; o_[2] = xs_re; o_[3] = xs_r0; o_[4] = r1_; \
        unsigned hw_, xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xc_)); o_[5] = ((unsigned long long)(xc_ & 15) << 32) | hw_; } } while (0)
#else
#define XS_ENTRY do { } while (0)
#define XS_DECL do { } while (0)
#define XS_ACC(k) do { } while (0)
#define XS_OUT do { } while (0)
#endif


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
union Frag { uint4 u; bf16x8 v; f16x8 h; s16x4 hs[2]; };

struct X6 { static constexpr int NP = 3; static constexpr bool SCALED = false; };
struct H3 { static constexpr int NP = 2; static constexpr bool SCALED = true; };

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).v, (b).v, (c), 0, 0, 0)
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a).h, (b).h, (c), 0, 0, 0)

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- H3 scaling: amax slots hold the bit pattern of a non-negative float (atomicMax on the bits orders them) ----
// exponent e with amax * 2^e in [2^14, 2^15); 0 for an all-zero tensor; clamped so that 2^e stays a normal float
// Upper clamps: a group whose maximum is below 2^-46 (activations, gradients) / 2^-16 (filters) is scaled as if it had that maximum and
// gives up one bit per binade below it.  They bound the sum of two operand exponents by 90, so that a value of ordinary magnitude
// brought to an accumulator's scale (the skip tile of the strip kernels) cannot overflow whatever the operands hold.
constexpr int H3_EMAX_ACT = 60, H3_EMAX_W = 30;
__device__ __forceinline__ int h3_exp_raw(unsigned amax_bits, int emax)
{
    const int E = (int)((amax_bits >> 23) & 0xffu);
    const int e = E == 0 ? 0 : 141 - E;
    return e > emax ? emax : e;
}
__device__ __forceinline__ int h3_exp(unsigned amax_bits) { return h3_exp_raw(amax_bits, H3_EMAX_ACT); }      // activations, gradients, bounds on register-resident tiles
__device__ __forceinline__ int h3_exp_w(unsigned amax_bits) { return h3_exp_raw(amax_bits, H3_EMAX_W); }      // filters
__device__ __forceinline__ int h3_exp(float bound) { return h3_exp(__float_as_uint(bound)); }
// largest slot of a per-sample array (wave-wide; every lane returns it): the scale of a tensor that is contracted over its samples
__device__ __forceinline__ unsigned amax_over_samples(const unsigned* slots, int n)
{
    unsigned m = 0u;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) { const unsigned v = slots[i]; m = v > m ? v : m; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned v = (unsigned)__shfl_xor((int)m, o, 64); m = v > m ? v : m; }
    return m;
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }      // -126 <= e <= 127
// the largest |value| a wave has produced -> its tensor's slot (one atomic per wave)
__device__ __forceinline__ void amax_commit(float m, unsigned* slot)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // NO read of the slot first ("skip the atomic if the slot already holds more"): the per-sample slots of a tensor share a few cache lines, and a load of a
    // line that thousands of atomics are queued on waits behind them -- measured on reflect_fold: 91 us with the guard, 27 without (tools/copybw.hip)
    if ((threadIdx.x & 63) == 0) atomicMax(slot, __float_as_uint(m));
}

// ---- X6: the three truncation pieces of one value, as fp32 bit patterns whose low 16 bits are zero
// (the mask lives in a scalar register: as a literal every v_and_b32 would be an 8-byte instruction)
__device__ __forceinline__ unsigned hi_mask()
{
    unsigned m;
    asm("s_mov_b32 %0, 0xffff0000" : "=s"(m));              // not volatile: one per kernel after CSE / hoisting
    return m;
}
__device__ __forceinline__ void pieces(float x, unsigned& p0, unsigned& p1, unsigned& p2)
{
    const unsigned M = hi_mask();
    p0 = __float_as_uint(x) & M;
    const float r = x - __uint_as_float(p0);
    p1 = __float_as_uint(r) & M;
    p2 = __float_as_uint(r - __uint_as_float(p1));          // <= 8 significant bits left: already a bf16 value
}
// pieces of a pair, packed (a -> low half, b -> high half of each dword)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2)
{
    unsigned a0, a1, a2, b0, b1, b2;
    pieces(a, a0, a1, a2);
    pieces(b, b0, b1, b2);
    q0 = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
    q1 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
    q2 = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
}

// H3: second pieces of a pair = fp16(v - h0), packed like the first.  The difference is exact in fp32, so one mixed-precision fma per element
// (fp16 source x -1.0 + fp32 source, fp32 arithmetic, fp16 result written to one half of the destination) gives the same bits as
// convert-back, subtract, convert in two instructions instead of four.  (Written out: the compiler folds fma(h, -1, v) back into a subtraction.)
__device__ __forceinline__ unsigned h3_second_pieces(unsigned h0, float va, float vb)
{
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h0), "v"(va));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h0), "v"(vb));
    return r;
}

// ---- arithmetic-generic forms: `s` is the operand tensor's power-of-two scale (ignored by X6) ----
// pieces of a pair, packed (a -> low half, b -> high half of each dword); q[p] = piece p
template <class AR>
__device__ __forceinline__ void cut_pair(float a, float b, float s, unsigned (&q)[AR::NP])
{
    if constexpr (AR::SCALED) {
        const f32x2 v = {a * s, b * s};
        const f16x2 h0 = __builtin_convertvector(v, f16x2);                 // v_cvt_pk_f16_f32: round to nearest even
        q[0] = __builtin_bit_cast(unsigned, h0);
        q[1] = h3_second_pieces(q[0], v[0], v[1]);
    } else {
        split_pair(a, b, q[0], q[1], q[2]);
    }
}
// the same for values that already carry their tensor's scale
template <class AR>
__device__ __forceinline__ void cut_pair_scaled(float a, float b, unsigned (&q)[AR::NP])
{
    if constexpr (AR::SCALED) {
        const f32x2 v = {a, b};
        const f16x2 h0 = __builtin_convertvector(v, f16x2);
        q[0] = __builtin_bit_cast(unsigned, h0);
        q[1] = h3_second_pieces(q[0], v[0], v[1]);
    } else {
        split_pair(a, b, q[0], q[1], q[2]);
    }
}
// 16-bit pieces of one value
template <class AR>
__device__ __forceinline__ void cut_one(float a, float s, unsigned short (&q)[AR::NP])
{
    if constexpr (AR::SCALED) {
        const float v = a * s;
        const _Float16 h0 = (_Float16)v;
        const _Float16 h1 = (_Float16)(v - (float)h0);
        q[0] = __builtin_bit_cast(unsigned short, h0);
        q[1] = __builtin_bit_cast(unsigned short, h1);
    } else {
        unsigned p0, p1, p2;
        pieces(a, p0, p1, p2);
        q[0] = (unsigned short)(p0 >> 16); q[1] = (unsigned short)(p1 >> 16); q[2] = (unsigned short)(p2 >> 16);
    }
}
// eight consecutive k-slots -> NP fragments
template <class AR>
__device__ __forceinline__ void cut8(const float (&x)[8], float s, Frag (&f)[AR::NP])
{
    unsigned q[4][AR::NP];
#pragma unroll
    for (int i = 0; i < 4; ++i) cut_pair<AR>(x[2 * i], x[2 * i + 1], s, q[i]);
#pragma unroll
    for (int p = 0; p < AR::NP; ++p) { f[p].u.x = q[0][p]; f[p].u.y = q[1][p]; f[p].u.z = q[2][p]; f[p].u.w = q[3][p]; }
}
template <class AR>
__device__ __forceinline__ void cut8_scaled(const float (&x)[8], Frag (&f)[AR::NP])
{
    unsigned q[4][AR::NP];
#pragma unroll
    for (int i = 0; i < 4; ++i) cut_pair_scaled<AR>(x[2 * i], x[2 * i + 1], q[i]);
#pragma unroll
    for (int p = 0; p < AR::NP; ++p) { f[p].u.x = q[0][p]; f[p].u.y = q[1][p]; f[p].u.z = q[2][p]; f[p].u.w = q[3][p]; }
}
// acc += A * B over one k-block of 16, smallest terms first
template <class AR>
__device__ __forceinline__ f32x16 mac(const Frag (&a)[AR::NP], const Frag (&b)[AR::NP], f32x16 acc)
{
    if constexpr (AR::SCALED) {
        acc = MFMA16H(a[1], b[0], acc);
        acc = MFMA16H(a[0], b[1], acc);
        acc = MFMA16H(a[0], b[0], acc);
    } else {
        acc = MFMA16(a[2], b[0], acc);
        acc = MFMA16(a[1], b[1], acc);
        acc = MFMA16(a[0], b[2], acc);
        acc = MFMA16(a[1], b[0], acc);
        acc = MFMA16(a[0], b[1], acc);
        acc = MFMA16(a[0], b[0], acc);
    }
    return acc;
}
// X6 spellings used by the kernels written before the tag existed
__device__ __forceinline__ void split8(const float (&x)[8], Frag (&f)[3]) { cut8<X6>(x, 1.f, f); }
__device__ __forceinline__ f32x16 mac6(const Frag (&a)[3], const Frag (&b)[3], f32x16 acc) { return mac<X6>(a, b, acc); }


// ---- LDS images of the fused pointwise kernels (kernels_x6.hip, kernels_pw4.hip) ----
constexpr int PB_ROW = 80;                  // bytes per voxel row of a piece image: 32 bf16 + 16 (row reads conflict-free)
constexpr int PB_IMG = 32 * PB_ROW;
constexpr int PB_TB = 32 * 33;              // floats of one dX partial

constexpr int PT_ROW = 72;                  // row bytes of a wave's transpose image (only 8-byte accesses)
constexpr int PT_IMG = 32 * PT_ROW;

template <int ROW>
__device__ __forceinline__ void tr_frag(const unsigned char* img, int lane, int kb, Frag& f)
{
    // operand [row|col = channel lane&31][k-slot j of half h <-> voxel 16kb + 8(j>>2) + 4h + (j&3)] of a [voxel][channel] image
    const int li = lane & 15, gcol = (lane >> 4) & 1, h = lane >> 5;
    const unsigned char* p = img + (16 * kb + 4 * h + (li >> 2)) * ROW + (16 * gcol + 4 * (li & 3)) * 2;
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    f.hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
    f.hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 8 * ROW));
}

constexpr int PS_IMG = 32 * 64;                     // bytes of one transpose image [32 voxels][32 fp16]
constexpr int PS_TB = 32 * 32;                      // floats of one dX partial [32 voxels][32 cin]
__device__ __forceinline__ int ps_key(int row) { return (row ^ (row >> 3)) & 7; }
__device__ __forceinline__ void tr_frag_sw(const unsigned char* img, int o0, int o1, Frag& f)
{
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    f.hs[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + o0));
    f.hs[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + o1));
}


__device__ __forceinline__ void h3_second_pieces2(unsigned h0a, float a0, float a1, unsigned h0b, float b0, float b1, unsigned& ra, unsigned& rb)
{
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(ra), "=&v"(rb) : "v"(h0a), "v"(h0b), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
// both pieces of four values that already carry their tensor's scale: q0 = {pair a, pair b} first pieces, q1 = second pieces
__device__ __forceinline__ void h3_cut4_scaled(float a0, float a1, float b0, float b1, uint2& q0, uint2& q1)
{
    const f32x2 va = {a0, a1}, vb = {b0, b1};
    q0.x = __builtin_bit_cast(unsigned, __builtin_convertvector(va, f16x2));
    q0.y = __builtin_bit_cast(unsigned, __builtin_convertvector(vb, f16x2));
    h3_second_pieces2(q0.x, a0, a1, q0.y, b0, b1, q1.x, q1.y);
}


}  // namespace probav
