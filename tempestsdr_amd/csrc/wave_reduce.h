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
__device__ __forceinline__ float wave_min(float v)
{
    float a, b;
    WR_SWAP32(v, a, b);
    v = fminf(a, b);
    WR_SWAP16(v, a, b);
    v = fminf(a, b);
    v = fminf(v, WR_ROW_SHL(v, v, 8));
    v = fminf(v, WR_ROW_SHL(v, v, 4));
    v = fminf(v, WR_ROW_SHL(v, v, 2));
    v = fminf(v, WR_ROW_SHL(v, v, 1));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    float a, b;
    WR_SWAP32(v, a, b);
    v = fmaxf(a, b);
    WR_SWAP16(v, a, b);
    v = fmaxf(a, b);
    v = fmaxf(v, WR_ROW_SHL(v, v, 8));
    v = fmaxf(v, WR_ROW_SHL(v, v, 4));
    v = fmaxf(v, WR_ROW_SHL(v, v, 2));
    v = fmaxf(v, WR_ROW_SHL(v, v, 1));
    return v;
}
#endif
