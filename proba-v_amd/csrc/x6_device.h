// Device helpers of the "x6" kernels (kernels_x6.hip, and the x6 variants inside kernels_mfma.hip): an fp32 value
// as three bf16 truncation pieces, and a product of two such triples as the six largest piece products.
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
#define XS_DECL unsigned long long xs_t = __builtin_amdgcn_s_memtime(), xs_acc[8] = {xs_t, 0, 0, 0, 0, 0, 0, 0}
#define XS_ACC(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); xs_acc[k] += t_ - xs_t; xs_t = t_; } while (0)
#define XS_OUT do { xs_acc[7] = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) for (int k_ = 0; k_ < 8; ++k_) g_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + k_] = xs_acc[k_]; } while (0)
#else
#define XS_DECL do { } while (0)
#define XS_ACC(k) do { } while (0)
#define XS_OUT do { } while (0)
#endif


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
union Frag { uint4 u; bf16x8 v; s16x4 hs[2]; };

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).v, (b).v, (c), 0, 0, 0)

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// the three truncation pieces of one value, as fp32 bit patterns whose low 16 bits are zero
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
// eight consecutive k-slots -> three fragments
__device__ __forceinline__ void split8(const float (&x)[8], Frag (&f)[3])
{
    split_pair(x[0], x[1], f[0].u.x, f[1].u.x, f[2].u.x);
    split_pair(x[2], x[3], f[0].u.y, f[1].u.y, f[2].u.y);
    split_pair(x[4], x[5], f[0].u.z, f[1].u.z, f[2].u.z);
    split_pair(x[6], x[7], f[0].u.w, f[1].u.w, f[2].u.w);
}
// acc += A * B over one k-block of 16, smallest terms first
__device__ __forceinline__ f32x16 mac6(const Frag (&a)[3], const Frag (&b)[3], f32x16 acc)
{
    acc = MFMA16(a[2], b[0], acc);
    acc = MFMA16(a[1], b[1], acc);
    acc = MFMA16(a[0], b[2], acc);
    acc = MFMA16(a[1], b[0], acc);
    acc = MFMA16(a[0], b[1], acc);
    acc = MFMA16(a[0], b[0], acc);
    return acc;
}

}  // namespace probav
