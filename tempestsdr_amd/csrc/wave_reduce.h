// wave_reduce.h — reductions over the 64 lanes of a wave whose result lands in lane 0, with the additions of the
// __shfl_down tree (v += lane+32, +16, +8, +4, +2, +1), but without the LDS crossbar: the two steps that cross the
// 16-lane rows are gfx950's v_permlane32_swap / v_permlane16_swap (VALU), the four inside a row are DPP row shifts
// folded into the add / min / max itself.  Six VALU instructions instead of six ds_bpermute round trips per reduction
// (scripts/micro/wave_reduce_check.hip holds lane 0 to the shuffle tree bit for bit).  Lanes other than 0 end up with
// partial results that nobody reads.
#ifndef TSDRGPU_WAVE_REDUCE_H
#define TSDRGPU_WAVE_REDUCE_H

// lane l < 32: a[l] and a[l+32]; lane l < 16 (after the first step): a[l] and a[l+16]
#define WR_SWAP32(v_, lo_, hi_)                                                                                 \
    {                                                                                                           \
        const auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(v_), __float_as_uint(v_), false, false); \
        lo_ = __uint_as_float(r_[0]);                                                                           \
        hi_ = __uint_as_float(r_[1]);                                                                           \
    }
#define WR_SWAP16(v_, lo_, hi_)                                                                                 \
    {                                                                                                           \
        const auto r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(v_), __float_as_uint(v_), false, false); \
        lo_ = __uint_as_float(r_[0]);                                                                           \
        hi_ = __uint_as_float(r_[1]);                                                                           \
    }
// lane l of a row reads lane l + N of the same row; a source beyond the row reads `old`
#define WR_ROW_SHL(v_, old_, N_) __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old_), __float_as_uint(v_), 0x100 + (N_), 0xf, 0xf, false))

__device__ __forceinline__ float wave_sum(float v)
{
    float a, b;
    WR_SWAP32(v, a, b);
    v = a + b;
    WR_SWAP16(v, a, b);
    v = a + b;
    v += WR_ROW_SHL(v, 0.f, 8);
    v += WR_ROW_SHL(v, 0.f, 4);
    v += WR_ROW_SHL(v, 0.f, 2);
    v += WR_ROW_SHL(v, 0.f, 1);
    return v;
}
// min / max: the values travel as order-preserving integer keys (the floats' order, -0 below +0; callers hold no NaN),
// so that every step is ONE integer min / max with the DPP operand folded in — the float form pays a v_mov_dpp and a
// canonicalising v_max per step
__device__ __forceinline__ int wr_key(float v)
{
    const int b = __builtin_bit_cast(int, v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float wr_unkey(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }
#define WR_SWAP32I(v_, lo_, hi_)                                                                       \
    {                                                                                                  \
        const auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)(v_), (unsigned)(v_), false, false); \
        lo_ = (int)r_[0];                                                                              \
        hi_ = (int)r_[1];                                                                              \
    }
#define WR_SWAP16I(v_, lo_, hi_)                                                                       \
    {                                                                                                  \
        const auto r_ = __builtin_amdgcn_permlane16_swap((unsigned)(v_), (unsigned)(v_), false, false); \
        lo_ = (int)r_[0];                                                                              \
        hi_ = (int)r_[1];                                                                              \
    }
#define WR_ROW_SHLI(v_, old_, N_) __builtin_amdgcn_update_dpp((int)(old_), (v_), 0x100 + (N_), 0xf, 0xf, false)

__device__ __forceinline__ float wave_min(float v)
{
    int k = wr_key(v), a, b;
    WR_SWAP32I(k, a, b);
    k = min(a, b);
    WR_SWAP16I(k, a, b);
    k = min(a, b);
    k = min(k, WR_ROW_SHLI(k, 0x7fffffff, 8));
    k = min(k, WR_ROW_SHLI(k, 0x7fffffff, 4));
    k = min(k, WR_ROW_SHLI(k, 0x7fffffff, 2));
    k = min(k, WR_ROW_SHLI(k, 0x7fffffff, 1));
    return wr_unkey(k);
}
__device__ __forceinline__ float wave_max(float v)
{
    int k = wr_key(v), a, b;
    WR_SWAP32I(k, a, b);
    k = max(a, b);
    WR_SWAP16I(k, a, b);
    k = max(a, b);
    k = max(k, WR_ROW_SHLI(k, 0x80000000, 8));
    k = max(k, WR_ROW_SHLI(k, 0x80000000, 4));
    k = max(k, WR_ROW_SHLI(k, 0x80000000, 2));
    k = max(k, WR_ROW_SHLI(k, 0x80000000, 1));
    return wr_unkey(k);
}
#endif
