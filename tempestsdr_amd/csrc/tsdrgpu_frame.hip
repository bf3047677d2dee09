// tsdrgpu_frame.hip — frame post-processing on gfx950: dsp_post_process
// (TempestSDR/src/dsp.c:134-239) and everything it calls.
//
// The reference handles one frame at a time on one core: two full passes for
// autogain, one for the row/column collapse, the sync detector on the two
// strips, an optional roll / line paint, and the temporal IIR.  Here a BATCH of
// F consecutive frames is pushed through three kinds of launches:
//
//   k_frame_stats   (big, HBM-bound)   per-tile min/max + row/column partial sums
//   k_frame_reduce  (small)            per-frame min/max and raw strips
//   k_chain         (2 workgroups)     the frame-to-frame recurrences: autogain IIR
//                                      state, sync detector (blur, sliding-window
//                                      best fit, dx low-pass) and the framerate PLL
//   k_frame_pass    (big, HBM-bound)   normalise + roll/lines + temporal IIR, all F
//                                      frames per thread so the IIR state stays in
//                                      registers
//
// Only the recurrences are sequential; they touch O(width+height) data per frame.
#include "tsdrgpu_internal.h"

#define PIX_G 512.0f
#define PIX_B 1024.0f

#define TILE_W 256
#define TILE_H 32

struct PpState {  // dsp_autogain_t + syncdetector_t, device resident
    float lastmax, lastmin;
    int dx_x, vx_x, strip_x;
    int dx_y, vx_y, strip_y;
    int locked;
    double avg_speed;
};

struct ChainOut {  // per frame, written by k_chain, read by k_frame_pass
    float lastmin, lastmax, span;
    int dx, vx, stripx;
    int dy, vy, stripy;
    int locked, pll_fired;
    double avg_speed, frameratediff;
};

struct tsdrgpu_postproc {
    tsdrgpu_t *g;
    PpState *d_state;
    float *d_screen;   // dsp_postprocess_t.screenbuffer (IIR state)
    size_t cap_screen;
    int *d_odd;        // k_frame_pass_par's "this batch needs the frame-by-frame form" flag
    float *d_screen2;  // second IIR buffer of the fused run (read one, write the other, swap)
    size_t cap_screen2;
    float *d_dump;     // where the fused pass's lanes outside the frame store (never read)
    size_t cap_dump;
    float *d_tmp1, *d_tmp2;  // intermediates for the non-default stage orders (F frames each)
    size_t cap_tmp1, cap_tmp2;
    // statistics scratch
    float *d_bmin, *d_bmax;
    int *d_tflag;  // per tile: it holds sentinel pixels (only then are the sentinel partials written / read)
    size_t cap_tflag;
    float *d_colp, *d_rowp;
    size_t cap_bmin, cap_bmax, cap_colp, cap_rowp;
    float *d_fmin, *d_fmax;
    double *d_strip_x, *d_strip_y;  // [F][3][n]: non-sentinel sum, sentinel sum, sentinel count
    size_t cap_fmin, cap_fmax, cap_sx, cap_sy;
    float *d_work;  // chain scratch: blurred strips + prefix sums
    int exact_ties; // tsdrgpu_postproc_set_exact_ties: second chain run for toss-up decisions
    int *d_sflag;   // [F][2]: the strip of (frame, axis) holds equal entries -> take the reference-order sums
    size_t cap_sflag;
    float *d_exact; // [F][2][nmax] strips summed in the reference's order (filled for flagged frames only)
    size_t cap_exact;
    size_t cap_work;
    ChainOut *d_chain;
    ChainOut *h_chain;  // pinned mirror
    size_t cap_chain;
    int width, height;
    int lowpass_before_sync;
    int chain_has_autogain;  // within one tsdrgpu_postproc_run: the autogain record of d_chain is valid
    int last_F;
    float taps[5];
    // split runs (tsdrgpu_postproc_begin / _finish)
    hipStream_t chain_st;       // where launch_chain queues (the context's main stream unless split)
    hipEvent_t ev_stats, ev_chain;
    int pending;                // 1: begin() done, chain queued on the side stream; 2: begin() deferred everything;
                                // 3: fused run (begin_minmax) queued completely, finish() only joins the streams;
                                // 4: the flat fused run (finish() also queues the gated literal pass);
                                // PEND_BAND: a row-band run — only band_finish / band_advance close it;
                                // PEND_BAND + 1: a general band run (band_open) — band_step closes it
#define PEND_BAND 5
    const float *p_frames;
    // row-band sharding (tsdrgpu_postproc_band_begin / _finish)
    int band_y0, band_rows;     // this rank's rows [y0, y0 + rows) of every frame
    int band_mode;              // launch_chain: strips come from the exchange, no literal re-collapse is possible
    double *d_xsum;             // [F][3][W + H]: column sums of the band, row sums of its rows (zero elsewhere)
    float *d_xmax;              // [F][4]: -min, max, pixel 0 of the frame (rank of band 0, else -inf), spare
    float *d_v0;                // [F] pixel 0 of every frame after the exchange
    ChainOut *d_chain_band;     // the chain record with dy relative to the band, for the pass
    size_t cap_xsum, cap_xmax, cap_v0, cap_chain_band;
    // contract-exact band runs (tsdrgpu_postproc_band_advance): the literal strip collapse relayed band by band
    int band_stage;             // 0 = after the first exchange; 1 = relay of round 0; 2 = run 0 done; 3 = relay of round 1; 4 = pass
    int relay_step, relay_items;
    int *d_items;               // the (frame*2 + axis) ids being relayed
    int *h_flags;               // host copy of a flag array [2F]
    double *d_relay;            // [items][nmax]: sums so far (f32 values carried as f64 through the sum all-reduce)
    size_t cap_items, cap_hflags, cap_relay;
    // ... speculated: both halves of the chain are queued without asking the host, the two flag arrays are read once behind them
    // (band_speculate)
    int band_fused;             // fused band run (tsdrgpu_postproc_band_begin_minmax): 1 = waiting for the min/max exchange, 2 = the trip is done
    int band_flat;              // ... and that trip was the flat one (motion blur 0: k_frame_stats<true>), whose exceptions the literal pass redoes
    int band_spec;              // -1 = the open run was speculated, a strip held ties, the literal run is under way (its first question answered)
    PpState *d_state_save;      // the autogain / sync state the batch started from
    hipEvent_t ev_spec;         // behind the copies of the speculated run's flags
    int *h_spec_flags;          // pinned: [2F] tie flags + [2F] toss-up flags behind the speculated chain
    size_t cap_spec_flags;
    unsigned long long band_spec_runs, band_spec_replays;  // counters (tsdrgpu_postproc_band_spec_stats)
    // general band runs (tsdrgpu_postproc_band_open / _band_step): every stage order, autoshift, PLL
    struct BandOp { int op, a, b, c; } bprog[24];
    int bprog_n, bprog_pc;
    int bsync_stage, bsync_norm;   // the sync detector's sub-machine (relays)
    int bnbands, bindex, brows_max;
    int bedges[65];                // row edges of all bands
    int *d_bedges;
    const float *bsrc[4];          // the band buffers a program works on: input, tmp1, tmp2, output
    const float *brelay_src;       // the buffer whose strips are being collapsed
    float *d_gather;               // [nbands][F][rows_max][W]: the frames every rank needs rows of for the 2-D roll
    size_t cap_gather;
    const float *ext_fmin, *ext_fmax;  // per-frame min/max supplied by the caller (fused run), else null
    // dsp_autogain_t.snr as a by-product (tsdrgpu_postproc_set_snr): one value per frame of the last run
    int want_snr, snr_frames;
    double *d_snr_part;
    float *d_snr;
    size_t cap_snr_part, cap_snr;
    float *p_out;                      // the fused run's frame buffer (given to _begin_minmax; _finish must name the same)
    int *clear_with_autogain;          // a device flag the next autogain chain launch zeroes (null: none)
    int p_F, p_W, p_H;
    tsdrgpu_pp_params_t p_prm;
};

// ---------------------------------------------------------------------------
// k_frame_stats: one workgroup per TILE_W x TILE_H tile of one frame.
// Wave w takes rows w, w+4, ...; lane l takes columns l, l+64, l+128, l+192 of
// the tile, so every load instruction covers 256 contiguous bytes.
// ---------------------------------------------------------------------------
// frames are W*H floats apart and rows W floats, so a frame or row base is only 4-byte aligned in general: these
// vector types say so, and gfx950 (unaligned access mode) still moves them with one dwordx2 / dwordx4 instruction
typedef float float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));

#include "wave_reduce.h"  // wave_sum / wave_min / wave_max: the shuffle tree's additions, result in lane 0

// STORE (the fused run at motion blur 0, tsdrgpu_postproc_begin_minmax): the autogain's range is already known (min/max from
// the resampler), so the same trip also writes the normalised frame — what k_frame_pass_par would read the frame a
// second time for.  The checks that guard the frame-parallel form of the IIR (see k_frame_pass_par) are made here.
struct StatsStore {
    float *dst;
    long long dstride;
    const struct ChainOut *chain;
    const float *screen;
    int *odd;
};
// (v - lastmin) / span for a whole frame's pixels: the divisor is the same 3.3 million times.  hipcc expands an IEEE f32
// division into  s = div_scale(d), r = rcp(s), r += (1 - s r) r  |  q = n r, q += (n - s q) r, div_fmas(n - s q, r, q), div_fixup
// — eleven instructions of which the first half depend on the divisor alone, and the scalings / fix-ups act only at the ends
// of the exponent range (v_div_scale_f32: divisor subnormal or above 2^126, exponents 96 apart, quotient subnormal, numerator
// below 2^-103).  NormDiv keeps the divisor's half per frame and runs the numerator's half bare: the SAME instructions on the
// same values, so the same bits as the `/` it replaces, provided nothing would have been scaled — `ok` says so from the frame's
// two scalars alone: 2^-20 <= span <= 2^20 and 2^-20 <= |lastmin| <= 2^10.  Then a pixel |v| <= 250 gives n = v - lastmin
// with |n| <= 2^11 and, unless it is zero, |n| >= 2^-44 (two distinct floats one of which is at least 2^-20 in magnitude differ by at
// least an ulp of that one), quotients between 2^-64 and 2^31: far inside.  n = 0 gives 0 either way.  Pixels above 250 in
// magnitude (sentinels, dsp.c:80-86) are not divided; non-finite results send the batch to the literal pass (`odd`).
struct NormDiv {
    float d, r;
    bool ok;
};
__device__ __forceinline__ NormDiv norm_div_setup(float lastmin, float span)
{
    NormDiv nd;
    nd.d = span;
    nd.ok = span >= 0x1p-20f && span <= 0x1p20f && fabsf(lastmin) >= 0x1p-20f && fabsf(lastmin) <= 0x1p10f;
    const float r0 = __builtin_amdgcn_rcpf(span);
    const float e = __builtin_fmaf(-span, r0, 1.0f);
    nd.r = __builtin_fmaf(e, r0, r0);
    return nd;
}
__device__ __forceinline__ float norm_div(const NormDiv &nd, float n)
{
    float q = n * nd.r;
    q = __builtin_fmaf(__builtin_fmaf(-nd.d, q, n), nd.r, q);
    return __builtin_fmaf(__builtin_fmaf(-nd.d, q, n), nd.r, q);
}

// what the frame-parallel IIR at coefficient 0 cannot reproduce: a non-finite value (it sticks to its pixel in the reference)
// and -0.0 (the reference's sum with the old state's +0 gives +0).  One v_cmp_class: sNaN | qNaN | -inf | -0 | +inf.
__device__ __forceinline__ bool px_odd(float o) { return __builtin_amdgcn_classf(o, 0x1 | 0x2 | 0x4 | 0x20 | 0x200); }
__device__ __forceinline__ bool px_nonfinite(float o) { return __builtin_amdgcn_classf(o, 0x1 | 0x2 | 0x4 | 0x200); }

// no_sentinel: the caller knows (wave-uniformly) that none of the wave's pixels exceeds 250 in magnitude
__device__ __forceinline__ void stats_store_tile(const float (&val)[TILE_H / 4][4], const StatsStore &st, int f, int W, int H, int x0, int y0,
                                                 int lane, int wave, bool no_sentinel)
{
    const float lastmin = st.chain[f].lastmin, span = st.chain[f].span;
    float *outp = st.dst + (long long)f * st.dstride;
    bool bad = false;
    const NormDiv nd = norm_div_setup(lastmin, span);
    if (nd.ok && x0 + TILE_W <= W && y0 + TILE_H <= H) {
        // the rule: a tile inside the frame, a divisor in range — no bounds, no branches, five instructions per division
#pragma unroll
        for (int r = 0; r < TILE_H / 4; r++) {
            const int y = y0 + wave + 4 * r;
            float4_a4 t;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = val[r][j];
                const float q = norm_div(nd, v - lastmin);
                const float o = (no_sentinel || !(v > 250.0f || v < -250.0f)) ? q : v;  // (no_sentinel is uniform: the tests fold away)
                t[j] = o;
                bad |= px_odd(o);
            }
            const long long at = (long long)y * W + x0 + 4 * lane;
            // (non-temporal hints on this store, on the raw-pixel loads above and on the resampler's sample loads: measured, each alone
            // and together, -0.3 to -3 % for the pass — profiles/round6_ab_runs.txt)
            *reinterpret_cast<float4_a4 *>(outp + at) = t;
            if (f == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) bad |= px_nonfinite(st.screen[at + j]);  // the incoming state has to be finite
            }
        }
        if (__any(bad) && lane == 0) atomicOr(st.odd, 1);
        return;
    }
#pragma unroll
    for (int r = 0; r < TILE_H / 4; r++) {
        const int y = y0 + wave + 4 * r;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = val[r][j];
            o[j] = (v > 250.0f || v < -250.0f) ? v : ((v - lastmin) / span);  // dsp.c:80-86, as pass_one(PASS_NORMALISE)
        }
        const int xl = x0 + 4 * lane;
        if (y < H && xl + 4 <= W) {
            float4_a4 t;
#pragma unroll
            for (int j = 0; j < 4; j++) t[j] = o[j];
            *reinterpret_cast<float4_a4 *>(outp + (long long)y * W + xl) = t;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = xl + j;
            if (y < H && x < W) {
                if (xl + 4 > W) outp[(long long)y * W + x] = o[j];
                bad |= px_odd(o[j]);
                if (f == 0) bad |= px_nonfinite(st.screen[(long long)y * W + x]);  // the incoming state has to be finite
            }
        }
    }
    if (__any(bad) && lane == 0) atomicOr(st.odd, 1);
}

// column of a lane's j-th value inside the tile: lanes stride the 64-column quarters (every load covers 256 contiguous
// bytes), or — in the storing form — own four neighbouring columns (dwordx4 loads and stores)
#define STATS_CX(j_) (STORE ? 4 * lane + (j_) : lane + 64 * (j_))
template <bool STORE>
__global__ __launch_bounds__(256) void k_frame_stats(const float *__restrict__ frames, long long fstride, int W, int H,
                                                     int tiles_x, int tiles_y, float *__restrict__ bmin,
                                                     float *__restrict__ bmax, float *__restrict__ colp,
                                                     float *__restrict__ rowp, int *__restrict__ tflag, int want_strips, StatsStore st)
{
    // 1-D grid, XCD-aware: a tile row is 1 KB that rarely starts on a 128-byte line (W*4 is no multiple
    // of 128), so horizontally adjacent tiles share a cache line; workgroups go round-robin to the 8
    // XCDs, so ids l, l+8, l+16 ... (one XCD, back to back) are mapped to consecutive tiles.
    const unsigned total = gridDim.x;
    const unsigned l = blockIdx.x;
    const unsigned logical = (total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;
    const int tx = logical % tiles_x, ty = (logical / tiles_x) % tiles_y, f = logical / (tiles_x * tiles_y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *src = frames + (long long)f * fstride;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;

    float cns[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0}, cc[4] = {0, 0, 0, 0};
    float lo = INFINITY, hi = -INFINITY;

    // phase 1: issue all of this wave's loads (8 rows x 4 columns per lane) before touching them,
    // so that 32 requests per lane are in flight (one dwordx4 per row measured 4 % slower here)
    constexpr int ROWS = TILE_H / 4;
    float val[ROWS][4];
    const bool interior = (x0 + TILE_W <= W) && (y0 + TILE_H <= H);  // workgroup-uniform, and the rule
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int y = y0 + wave + 4 * r;
        if (STORE && interior) {  // no bounds to test: eight dwordx4 requests back to back
            const float4_a4 t = *reinterpret_cast<const float4_a4 *>(src + (long long)y * W + x0 + 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; j++) val[r][j] = t[j];
            continue;
        }
        const float *row = src + (long long)(y < H ? y : 0) * W;
        if (STORE && y < H && x0 + 4 * lane + 4 <= W) {  // the storing form: a lane's four columns are neighbours, one dwordx4
            const float4_a4 t = *reinterpret_cast<const float4_a4 *>(row + x0 + 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; j++) val[r][j] = t[j];
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + STATS_CX(j);
            val[r][j] = (y < H && x < W) ? row[x] : NAN;  // NaN = outside the frame (neither branch below takes it)
        }
    }
    // a wave whose 32 x 256 pixels are all inside the frame and hold no sentinel (the rule, decided with one
    // max3 per two pixels and a ballot) only needs the plain sums and min/max: same additions in the same
    // order as the general form below, whose sentinel accumulators would all stay zero — and, in the storing form, no
    // sentinel test per pixel
    bool plain = interior;
    if (plain) {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            m = fmaxf(fmaxf(m, fabsf(val[r][0])), fabsf(val[r][1]));
            m = fmaxf(fmaxf(m, fabsf(val[r][2])), fabsf(val[r][3]));
        }
        plain = __builtin_amdgcn_ballot_w64(m > 250.0f) == 0ull;  // wave-uniform
    }
    if (STORE) {
        if (plain) stats_store_tile(val, st, f, W, H, x0, y0, lane, wave, true);
        else stats_store_tile(val, st, f, W, H, x0, y0, lane, wave, false);
    }
    // phase 2.  Sentinel pixels (|v| > 250) are rare — raw resampler output has none — so their partial
    // sums are only reduced and written when the tile holds any (tflag), which saves two thirds of the
    // partial-sum traffic; k_frame_reduce reads them under the same flag.
    float prs[ROWS], prc[ROWS];
    if (plain) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            float rns = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = val[r][j];
                cns[j] += v;
                rns += v;
            }
            lo = fminf(fminf(lo, val[r][0]), val[r][1]);
            lo = fminf(fminf(lo, val[r][2]), val[r][3]);
            hi = fmaxf(fmaxf(hi, val[r][0]), val[r][1]);
            hi = fmaxf(fmaxf(hi, val[r][2]), val[r][3]);
            prs[r] = 0.f;
            prc[r] = 0.f;
            if (want_strips) {
                rns = wave_sum(rns);
                if (lane == 0) rowp[((long long)(f * tiles_x + tx) * 3) * H + (y0 + wave + 4 * r)] = rns;
            }
        }
    } else {
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int y = y0 + wave + 4 * r;
        float rns = 0.f, rs = 0.f, rc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float v = val[r][j];
            const bool sent = (v > 250.0f) || (v < -250.0f);  // dsp.c:57
            const bool inside = (y < H) && (x0 + STATS_CX(j) < W);
            if (inside) {
                if (sent) {
                    cs[j] += v; cc[j] += 1.f; rs += v; rc += 1.f;
                } else {
                    cns[j] += v; rns += v;
                    lo = fminf(lo, v);
                    hi = fmaxf(hi, v);
                }
            }
        }
        prs[r] = rs;
        prc[r] = rc;
        if (want_strips && y < H) {  // y is wave-uniform
            rns = wave_sum(rns);
            if (lane == 0) rowp[((long long)(f * tiles_x + tx) * 3) * H + y] = rns;
        }
    }
    }
    const int tile_sent = __syncthreads_or((cc[0] + cc[1] + cc[2] + cc[3]) != 0.f);
    if (want_strips && tile_sent) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int y = y0 + wave + 4 * r;
            if (y < H) {
                const float rs = wave_sum(prs[r]), rc = wave_sum(prc[r]);
                if (lane == 0) {
                    float *rp = rowp + ((long long)(f * tiles_x + tx) * 3) * H;
                    rp[H + y] = rs;
                    rp[2 * H + y] = rc;
                }
            }
        }
    }

    __shared__ float sh[3][4][TILE_W];
    __shared__ float shmin[4], shmax[4];
    if (want_strips) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sh[0][wave][STATS_CX(j)] = cns[j];
            if (tile_sent) {
                sh[1][wave][STATS_CX(j)] = cs[j];
                sh[2][wave][STATS_CX(j)] = cc[j];
            }
        }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) { shmin[wave] = lo; shmax[wave] = hi; }
    __syncthreads();
    if (want_strips) {
        const int x = x0 + threadIdx.x;
        if (x < W) {
            float *cp = colp + ((long long)(f * tiles_y + ty) * 3) * W;
            const int nq = tile_sent ? 3 : 1;
            for (int q = 0; q < nq; q++)
                cp[q * W + x] = sh[q][0][threadIdx.x] + sh[q][1][threadIdx.x] + sh[q][2][threadIdx.x] + sh[q][3][threadIdx.x];
        }
    }
    if (threadIdx.x == 0) {
        const long long b = (long long)f * tiles_x * tiles_y + (long long)ty * tiles_x + tx;
        bmin[b] = fminf(fminf(shmin[0], shmin[1]), fminf(shmin[2], shmin[3]));
        bmax[b] = fmaxf(fmaxf(shmax[0], shmax[1]), fmaxf(shmax[2], shmax[3]));
        tflag[b] = tile_sent;
    }
}

#undef STATS_CX

// ---------------------------------------------------------------------------
// k_frame_reduce: grid (blocks over the strip, 3, F): y=0 min/max of the frame
// (block x=0 only), y=1 column strips, y=2 row strips (f64 sums over the tile
// partials, one thread per strip element).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame_reduce(int W, int H, int tiles_x, int tiles_y, const float *__restrict__ bmin,
                                                      const float *__restrict__ bmax, const float *__restrict__ colp,
                                                      const float *__restrict__ rowp, float *__restrict__ fmin_,
                                                      float *__restrict__ fmax_, double *__restrict__ strip_x,
                                                      double *__restrict__ strip_y, const int *__restrict__ tflag,
                                                      int want_strips, int want_minmax, int tile_h)
{
    const int f = blockIdx.z;
    if (blockIdx.y == 0) {
        if (blockIdx.x != 0 || !want_minmax) return;
        const int nblk = tiles_x * tiles_y;
        float lo = INFINITY, hi = -INFINITY;
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
            lo = fminf(lo, bmin[(long long)f * nblk + b]);
            hi = fmaxf(hi, bmax[(long long)f * nblk + b]);
        }
        __shared__ float slo[4], shi[4];
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            fmin_[f] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
            fmax_[f] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        }
    } else if (want_strips) {
        const bool cols = blockIdx.y == 1;
        const int n = cols ? W : H;
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        const int parts = cols ? tiles_y : tiles_x;
        const float *src = (cols ? colp : rowp) + (long long)f * parts * 3 * n;
        double *dst = (cols ? strip_x : strip_y) + (long long)f * 3 * n;
        // tile (ty, tx) that partial p of strip element i came from
        const int *fl = tflag + (long long)f * tiles_x * tiles_y + (cols ? i / TILE_W : (i / tile_h) * tiles_x);
        const int fstep = cols ? tiles_x : 1;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        // the partials of twelve tiles are requested together (the loop used to pay a load's latency per tile: a strip
        // element has up to ~70 of them and the kernel too few threads to hide that), then added in the same order
        constexpr int RB = 12;
        for (int p0 = 0; p0 < parts; p0 += RB) {
            float v0[RB];
            int sf[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const int p = p0 + k < parts ? p0 + k : parts - 1;
                v0[k] = src[(long long)p * 3 * n + i];
                sf[k] = fl[p * fstep];
            }
#pragma unroll
            for (int k = 0; k < RB; k++) {
                if (p0 + k >= parts) break;
                a0 += (double)v0[k];
                if (sf[k]) {
                    const float *s3 = src + (long long)(p0 + k) * 3 * n + i;
                    a1 += (double)s3[n];
                    a2 += (double)s3[2 * n];
                }
            }
        }
        dst[i] = a0;
        dst[n + i] = a1;
        dst[2 * n + i] = a2;
    }
}

// ---------------------------------------------------------------------------
// The frame-to-frame recurrences.
//   k_autogain_chain  scalar IIR of min/max over the F frames (dsp.c:50-66)
//   k_strip_prepare   per frame and axis, in parallel: strip of the (autogained)
//                     frame, circular 5-tap blur (gaussian.c:18-79), f64 prefix
//                     sums and total — none of it depends on the sync state
//   k_sync_chain      2 workgroups (x, y): sliding-window best fit for the up to
//                     five strip sizes (syncdetector.c:26-119), dx low-pass, PLL
//                     (syncdetector.c:133-153) — sequential over the frames, but
//                     only O(n) prefix look-ups + one reduction per frame
// ---------------------------------------------------------------------------
#define CHAIN_T 1024
#define STRIP_MAX TSDRGPU_MAX_STRIP

__device__ __forceinline__ double wave_incl_scan(double v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

struct FitBest {
    double fit;
    int q;
    int q2;         // where that runner-up starts
    double second;  // the largest fit among the OTHER windows of this size (-1: none seen)
};
__device__ __forceinline__ FitBest better(FitBest a, FitBest b)
{
    // larger fit wins; on equal fits the earlier position (syncdetector.c:53 uses `>`); the loser and both
    // runner-ups compete for `second`
    const bool bwins = b.fit > a.fit || (b.fit == a.fit && b.q < a.q);
    FitBest w = bwins ? b : a;
    const FitBest l = bwins ? a : b;
    double s2 = w.second;
    int p2 = w.q2;
    if (l.second > s2) { s2 = l.second; p2 = l.q2; }
    if (l.q != 0x7fffffff && l.fit > s2) { s2 = l.fit; p2 = l.q; }
    w.second = s2;
    w.q2 = p2;
    return w;
}

__global__ __launch_bounds__(64) void k_autogain_chain(int F, const float *__restrict__ frames, long long fstride,
                                                       const float *__restrict__ fmin_, const float *__restrict__ fmax_,
                                                       PpState *__restrict__ state, ChainOut *__restrict__ out,
                                                       int do_autogain, float norm, int *__restrict__ clear)
{
    if (clear && threadIdx.x == 0) *clear = 0;  // the redo flag of the pass queued behind this kernel (saves a memset's launch)
    // stage the per-frame inputs in LDS in parallel, then one lane walks the recurrence
    __shared__ float sv0[256], slo[256], shi[256];
    float lastmax = state->lastmax, lastmin = state->lastmin;
    for (int base = 0; base < F; base += 256) {
        const int cnt = (F - base < 256) ? (F - base) : 256;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            sv0[i] = do_autogain ? frames[(long long)(base + i) * fstride] : 0.f;
            slo[i] = do_autogain ? fmin_[base + i] : 0.f;
            shi[i] = do_autogain ? fmax_[base + i] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < cnt; i++) {
                if (do_autogain) {
                    // dsp.c:50-66: min/max start from v[0] even when it is a sentinel
                    // (and a NaN there stays: `val > max` / `val < min` are false against it for every later pixel)
                    const bool v0nan = sv0[i] != sv0[i];
                    const float hi = v0nan ? sv0[i] : fmaxf(sv0[i], shi[i]);
                    const float lo = v0nan ? sv0[i] : fminf(sv0[i], slo[i]);
                    const float keep = 1.0f - norm;
                    lastmax = keep * lastmax + norm * hi;
                    lastmin = keep * lastmin + norm * lo;
                }
                ChainOut *o = &out[base + i];
                o->lastmin = lastmin;
                o->lastmax = lastmax;
                o->span = (lastmax == lastmin) ? 1.0f : (lastmax - lastmin);
            }
        }
    }
    if (threadIdx.x == 0 && do_autogain) {
        state->lastmax = lastmax;
        state->lastmin = lastmin;
    }
}

// scratch layout per (frame, axis): blur[n] floats, prefix[n+1] doubles, total
struct StripScratch {
    float *blur;     // [F][2][nmax]
    double *prefix;  // [F][2][nmax+1]
    double *total;   // [F][2]
    int nmax;
};

// ---------------------------------------------------------------------------
// Exact ties.  The strips above come from exact f64 sums, the reference's from f32 additions in
// raster order (dsp.c:96-110); they agree to ~1e-6 — except that where several window positions fit
// EXACTLY equally (plateaus, periodic test patterns, blank frames) the reference's winner is decided
// by its own rounding.  Equal window sums need equal strip entries, and exact f64 sums of noisy data
// (almost) never collide, so: k_strip_flag counts equal entries (hashed presence map in LDS) and flags
// (frame, axis) if there are many (or sentinels are present, or the strip is nearly flat); for flagged frames k_exact_strips redoes the collapse literally — one
// thread per column / per row adding the (autogained) pixels in f32 in the reference's order — and
// k_strip_prepare takes those.  Unflagged frames (every measured frame) pay one early-exit each.
// ---------------------------------------------------------------------------
#define FLAG_BITS 20  // 2^20-bit presence map (128 KiB of LDS): equal sums always collide, unequal ones ~n^2/2^21 times
__global__ __launch_bounds__(CHAIN_T) void k_strip_flag(int W, int H, const double *__restrict__ strip_x,
                                                        const double *__restrict__ strip_y, const ChainOut *__restrict__ chain,
                                                        int strips_normalised, int *__restrict__ sflag)
{
    __shared__ unsigned bitmap[1u << (FLAG_BITS - 5)];
    __shared__ double rlo[CHAIN_T / 64], rhi[CHAIN_T / 64];
    __shared__ int ndup, flagged;
    const int axis = blockIdx.x, f = blockIdx.y;
    const int n = axis == 0 ? W : H;
    const double *sp = (axis == 0 ? strip_x : strip_y) + (long long)f * 3 * n;
    const double cnt_all = (double)(axis == 0 ? H : W);
    const double lastmin = chain[f].lastmin, span = chain[f].span;
    for (unsigned i = threadIdx.x; i < (1u << (FLAG_BITS - 5)); i += CHAIN_T) bitmap[i] = 0u;
    if (threadIdx.x == 0) { ndup = 0; flagged = 0; }
    __syncthreads();
    int sent = 0, dup = 0;
    double lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += CHAIN_T) {
        const double ns = sp[i];
        sent |= (sp[2 * n + i] != 0.0) ? 1 : 0;
        const double v = strips_normalised ? (ns - cnt_all * lastmin) / span : ns;  // the strip entry (no sentinels)
        lo = fmin(lo, v);
        hi = fmax(hi, v);
        unsigned long long z = (unsigned long long)__double_as_longlong(ns);  // splitmix64 finaliser
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const unsigned bit = (unsigned)z & ((1u << FLAG_BITS) - 1u);
        const unsigned old = atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
        dup += (old >> (bit & 31)) & 1u;
    }
    if (dup) atomicAdd(&ndup, dup);
    // Low contrast is the other way to a coin toss: when the strip varies by less than ~1 % of its level
    // (a frame that is nearly blank after the autogain), window fits differ by less than the rounding
    // of the reference's own f32 sums, so the literal collapse is taken as well.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { rlo[threadIdx.x >> 6] = lo; rhi[threadIdx.x >> 6] = hi; }
    sent = __syncthreads_or(sent);  // also: ndup, rlo, rhi complete
    if (threadIdx.x == 0) {
        for (int w = 1; w < CHAIN_T / 64; w++) { lo = fmin(lo, rlo[w]); hi = fmax(hi, rhi[w]); }
        const double level = fmax(fabs(lo), fabs(hi));
        const bool flat = !(hi - lo > 1e-2 * level);  // also true for NaN / empty
        // The sums are exact only down to the f32 tile partials of k_frame_stats, so a noisy frame shows a
        // few accidental collisions (0-3 equal pairs per 2962-entry strip, plus ~4 of the map itself);
        // structure (plateaus, periodic patterns, blank frames) shows them by the hundred.  Flag from n/32
        // on (from one on for strips too short for accidents).
        const int limit = n < 256 ? 1 : (n / 32 > 16 ? n / 32 : 16);
        sflag[f * 2 + axis] = (ndup >= limit || sent || flat) ? 1 : 0;
    }
}

// grid (max(ceil(W/64), ceil(H/64)), 2, F).  A workgroup forms 64 sums — axis 0: columns x0..x0+63 summed down all
// rows, axis 1: rows y0..y0+63 summed along all columns — in the reference's order: one thread per sum adds f32
// values one after the other (dsp.c:103-108).  That chain is short (a few thousand dependent adds); what costs is the
// latency of the loads, so the frame is walked in tiles of 256 (along the sum) x 64 staged through LDS: all four waves
// fetch tile k+1 (64 coalesced loads per lane in flight) while wave 0 sums tile k out of LDS.
#define XS_ITEMS 16  // strips re-collapsed side by side
#define XS_LONG 64   // terms per tile: 64 x 65 floats = 16.6 KB of LDS, so that a workgroup finds room beside the FFT trips' (2 x 70 KB per CU)
#define XS_E (XS_LONG / 4)  // elements per thread and tile
#define XS_WAVES 4
__global__ __launch_bounds__(256) void k_exact_strips(const float *__restrict__ frames, long long fstride, int W, int H,
                                                      const ChainOut *__restrict__ chain, int strips_normalised,
                                                      const int *__restrict__ sflag, float *__restrict__ exact, int nmax,
                                                      const int *__restrict__ redo, const int *__restrict__ only, int nstrips)
{
    if (redo && !*redo) return;
    // grid (blocks of 64 sums, XS_ITEMS): row j of the grid takes the j-th, (j + XS_ITEMS)-th ... flagged strip of the
    // batch.  Almost every launch finds none: a grid with a row per (frame, axis) spent ~0.1 ms on the latency-bound
    // chain dispatching thousands of workgroups (66 KB of LDS each, beside the FFT's) that returned at once.
    const int lane_ = threadIdx.x & 63;
    for (int target = blockIdx.y;; target += gridDim.y) {
    int item = -1, seen = 0;
    for (int base = 0; base < nstrips && item < 0; base += 64) {
        const int i = base + lane_;
        const bool on = i < nstrips && sflag[i] && (!only || only[i]);  // (second run: the first one already made the others exact)
        unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        const int n = __builtin_popcountll(m);
        if (target < seen + n) {
            for (int k = target - seen; k > 0; k--) m &= m - 1;
            item = base + __builtin_ctzll(m);
        }
        seen += n;
    }
    if (item < 0) return;  // (workgroup-uniform: every wave sees the same flags)
    const int axis = item & 1, f = item >> 1;
    const int nsum = axis == 0 ? W : H;   // how many sums this axis has
    const int nlong = axis == 0 ? H : W;  // how many terms each sum has
    const int s0 = blockIdx.x * 64;       // first sum of this workgroup
    if (s0 >= nsum) continue;             // (this axis is the shorter one: on to the next strip)
    const float *src = frames + (long long)f * fstride;
    const float lastmin = chain[f].lastmin, span = chain[f].span;
    float *out = exact + ((long long)f * 2 + axis) * nmax;
    // tile[t][s]: term t (0..255 along the sum), sum s (0..63); rows padded to 65 so that both the transposing
    // store of axis 1 and the walk of one thread down its column are conflict free
    __shared__ float tile[XS_LONG][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[XS_E];
    const int ntiles = (nlong + XS_LONG - 1) / XS_LONG;
    float acc = 0.f;
    // element i of a thread: axis 0: term t = wave + 4 i, sum s = lane          -> pixel (x = s0 + lane, y = t0 + t)
    //                        axis 1: sum s = wave + 4 (i & 15), term t = lane + 64 (i >> 4) -> pixel (x = t0 + t, y = s0 + s)
    // either way a wave's load instruction covers 256 contiguous bytes
    // loads first, unconditionally (indices clamped into the frame), so that all 64 are in flight together; the
    // normalisation and the inside-the-frame test follow
#define XS_FETCH(t0_)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < XS_E; i++) {                                                        \
        const int t = axis == 0 ? wave + XS_WAVES * i : lane + 64 * (i >> 4);                               \
        const int sidx = axis == 0 ? lane : wave + XS_WAVES * (i & 15);                                     \
        int x = axis == 0 ? s0 + sidx : (t0_) + t, y = axis == 0 ? (t0_) + t : s0 + sidx;                   \
        x = x < W ? x : W - 1;                                                                              \
        y = y < H ? y : H - 1;                                                                              \
        v[i] = src[(long long)y * W + x];                                                                   \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < XS_E; i++) {                                                        \
        const int t = axis == 0 ? wave + XS_WAVES * i : lane + 64 * (i >> 4);                               \
        const int sidx = axis == 0 ? lane : wave + XS_WAVES * (i & 15);                                     \
        const int x = axis == 0 ? s0 + sidx : (t0_) + t, y = axis == 0 ? (t0_) + t : s0 + sidx;             \
        float val = v[i];                                                                                   \
        if (strips_normalised) val = (val > 250.0f || val < -250.0f) ? val : ((val - lastmin) / span); /* dsp.c:80-86 */ \
        v[i] = (x < W && y < H) ? val : 0.f;                                                                \
    }
    XS_FETCH(0)
    for (int k = 0; k < ntiles; k++) {
#pragma unroll
        for (int i = 0; i < XS_E; i++) {
            const int t = axis == 0 ? wave + XS_WAVES * i : lane + 64 * (i >> 4);
            const int sidx = axis == 0 ? lane : wave + XS_WAVES * (i & 15);
            tile[t][sidx] = v[i];
        }
        __syncthreads();
        if (k + 1 < ntiles) { XS_FETCH((k + 1) * XS_LONG) }
        if (threadIdx.x < 64) {
            const int cnt = nlong - k * XS_LONG < XS_LONG ? nlong - k * XS_LONG : XS_LONG;
            if (cnt == XS_LONG) {
                // the adds form one dependent chain; the LDS reads do not, so they are issued 32 ahead
#pragma unroll 32
                for (int t = 0; t < XS_LONG; t++) acc += tile[t][threadIdx.x];
            } else {
                int t = 0;
                for (; t + 8 <= cnt; t += 8) {
                    float r[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) r[u] = tile[t + u][threadIdx.x];
#pragma unroll
                    for (int u = 0; u < 8; u++) acc += r[u];
                }
                for (; t < cnt; t++) acc += tile[t][threadIdx.x];
            }
        }
        __syncthreads();  // tile consumed
    }
#undef XS_FETCH
    if (threadIdx.x < 64 && s0 + (int)threadIdx.x < nsum) out[s0 + threadIdx.x] = acc;
    __syncthreads();  // the tile is free for the next strip
    }
}

// BIG: a strip longer than STRIP_MAX entries (the reference bounds width x height, not each: 100 MS/s with a 100-line raster at
// 60 Hz is 33 333 pixels per line, TSDRLibrary.c:31,489,543-546) does not fit the LDS; the strip then sits in its slot of `exact`
// and its blur in the scratch's own copy, both in HBM — the same arithmetic in the same order, a rare geometry's speed.
template <bool BIG>
__global__ __launch_bounds__(CHAIN_T) void k_strip_prepare(int W, int H, const double *__restrict__ strip_x,
                                                           const double *__restrict__ strip_y,
                                                           const ChainOut *__restrict__ chain, StripScratch sc,
                                                           int strips_normalised, float t0, float t1, float t2, float t3,
                                                           float t4, const int *__restrict__ sflag,
                                                           float *exact, const int *__restrict__ redo)
{
    if (redo && !*redo) return;
    __shared__ float lds_data[BIG ? 1 : STRIP_MAX];
    __shared__ float lds_blur[BIG ? 1 : STRIP_MAX];
    __shared__ double wsum[16];
    const int axis = blockIdx.x, f = blockIdx.y;
    const int n = axis == 0 ? W : H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *sp = (axis == 0 ? strip_x : strip_y) + (long long)f * 3 * n;
    const double cnt_all = (double)(axis == 0 ? H : W);
    const float lastmin = chain[f].lastmin, span = chain[f].span;
    float *const gblur = sc.blur + ((long long)f * 2 + axis) * sc.nmax;
    float *const data = BIG ? exact + ((long long)f * 2 + axis) * sc.nmax : lds_data;
    float *const blur = BIG ? gblur : lds_blur;
    if (sflag[f * 2 + axis]) {  // reference-order sums, see k_strip_flag
        const float *ex = exact + ((long long)f * 2 + axis) * sc.nmax;
        if (!BIG)  // (BIG: they are where `data` points already)
            for (int i = tid; i < n; i += CHAIN_T) data[i] = ex[i];
    } else {
        for (int i = tid; i < n; i += CHAIN_T) {
            const double ns = sp[i], s = sp[n + i], c = sp[2 * n + i];
            double v;
            if (strips_normalised)  // strip of the autogained frame, from the raw frame's sums
                v = (ns - (cnt_all - c) * (double)lastmin) / (double)span + s;
            else
                v = ns + s;
            data[i] = (float)v;
        }
    }
    __syncthreads();
    // gaussianblur: out[(i+2)%n] = sum_k taps[k]*in[(i+k)%n], left to right in f32
    for (int i = tid; i < n; i += CHAIN_T) {
        const float a = data[i % n], b = data[(i + 1) % n], c = data[(i + 2) % n];
        const float d = data[(i + 3) % n], e = data[(i + 4) % n];
        const float v = a * t0 + b * t1 + c * t2 + d * t3 + e * t4;
        blur[(i + 2) % n] = v;
        if (!BIG) gblur[(i + 2) % n] = v;
    }
    __syncthreads();
    // f64 prefix sums: thread t owns [t*per, (t+1)*per)
    const int per = (n + CHAIN_T - 1) / CHAIN_T;
    const int b0 = tid * per;
    double local = 0.0;
    for (int i = b0; i < b0 + per && i < n; i++) local += (double)blur[i];
    const double incl = wave_incl_scan(local, lane);
    double excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        double w = (lane < 16) ? wsum[lane] : 0.0;
        w = wave_incl_scan(w, lane);
        if (lane < 16) wsum[lane] = w;
    }
    __syncthreads();
    double *gprefix = sc.prefix + ((long long)f * 2 + axis) * (sc.nmax + 1);
    double run = excl + (wave ? wsum[wave - 1] : 0.0);
    for (int i = b0; i < b0 + per && i < n; i++) {
        gprefix[i] = run;
        run += (double)blur[i];
    }
    if (tid == 0) {
        gprefix[n] = wsum[15];
        sc.total[f * 2 + axis] = wsum[15];
    }
}

#define TSDR_LAUNCH_STRIP_PREPARE(g_, st_, nmax_, ...)                                        \
    do {                                                                                     \
        if ((nmax_) > STRIP_MAX) TSDR_LAUNCH(g_, PROF_CHAIN, st_, k_strip_prepare<true>, __VA_ARGS__);  \
        else TSDR_LAUNCH(g_, PROF_CHAIN, st_, k_strip_prepare<false>, __VA_ARGS__);          \
    } while (0)

#define SYNC_T 256
#define CHAIN_STAGE 128  // frames whose search results k_sync_chain stages in LDS at a time

struct SearchShared {
    FitBest wbest[5][SYNC_T / 64];
    double wlo[5][SYNC_T / 64], whi[5][SYNC_T / 64];
    FitBest best[5];  // result: per candidate size the first window start with the largest fit
};

// candidate strip sizes around `cur` (RUNWITH_SIZE, syncdetector.c:60-69,90-93); 0 = not tried
__device__ __forceinline__ int sync_sizes(int cur, int minsize, int half, int sizes[5])
{
    if (cur < minsize) cur = minsize; else if (cur > half) cur = half;  // syncdetector.c:76-77
    sizes[0] = cur;
    const int trial[4] = {cur - 4, cur + 4, cur >> 1, cur << 1};
#pragma unroll
    for (int t = 0; t < 4; t++) sizes[t + 1] = (trial[t] >= minsize && trial[t] < half && trial[t] != cur) ? trial[t] : 0;
    return cur;
}

// findbestfit (syncdetector.c:26-58) for the candidate sizes at once, by the whole workgroup.
// P = prefix sums of the blurred strip (n+1 doubles).  fit(q) = ((total-S_q)/(n-s) - S_q/s)^2 is,
// with every rounding kept, a monotone function of the window sum S_q on either side of its
// zero, so its maximum over q sits at the smallest or the largest S_q.  Step 1 finds those
// extremes with adds and compares only; step 2 evaluates the reference's exact f64 expression
// (two divisions) just for the windows within a 1e-9 band of them (rounding plateaus are ~1e-16
// wide); the (fit, q) reduction applies the reference's first-maximum rule.  Result: S.best[].
__device__ void strip_search(SearchShared &S, const double *__restrict__ P, int n, const int sizes[5])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double pn_ = P[n];
    const float totalf = (float)pn_;  // narrowed by findbestfit's float parameter
    int sz[5];
    double c1[5], c2[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        sz[k] = sizes[k] > 0 ? sizes[k] : sizes[0];  // untried sizes alias the current one (results ignored)
        c1[k] = (double)(n - sz[k]);
        c2[k] = (double)sz[k];
    }
    double lo[5], hi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
#pragma unroll 4
    for (int q = tid; q < n; q += SYNC_T) {
        const double pq = P[q];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int e = q + sz[k];
            const double pe = P[e <= n ? e : e - n];
            const double sum = (e <= n) ? (pe - pq) : (pn_ - pq + pe);
            lo[k] = (sum < lo[k]) ? sum : lo[k];
            hi[k] = (sum > hi[k]) ? sum : hi[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_xor(lo[k], o, 64), h2 = __shfl_xor(hi[k], o, 64);
            lo[k] = (l2 < lo[k]) ? l2 : lo[k];
            hi[k] = (h2 > hi[k]) ? h2 : hi[k];
        }
        if (lane == 0) { S.wlo[k][wave] = lo[k]; S.whi[k][wave] = hi[k]; }
    }
    __syncthreads();
    FitBest mine[5];
    double lo_b[5], hi_b[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double l = S.wlo[k][0], h = S.whi[k][0];
#pragma unroll
        for (int w = 1; w < SYNC_T / 64; w++) {
            l = (S.wlo[k][w] < l) ? S.wlo[k][w] : l;
            h = (S.whi[k][w] > h) ? S.whi[k][w] : h;
        }
        // windows within 4e-6 of an extreme sum are evaluated: the winner is among those within 1e-9, the rest
        // supply the runner-up for the toss-up test of k_sync_chain (whose tolerance, expressed in window sums,
        // is ~4*sqrt(2/strip)*7e-7 of the sum: below 1e-6 for every strip size)
        const double band = 4e-6 * (fabs(l) + fabs(h)) + 1e-300;
        lo_b[k] = l + band;
        hi_b[k] = h - band;
        mine[k].fit = -1.0;
        mine[k].q = 0x7fffffff;
        mine[k].q2 = 0;
        mine[k].second = -1.0;
    }
#pragma unroll 4
    for (int q = tid; q < n; q += SYNC_T) {
        const double pq = P[q];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int e = q + sz[k];
            const double pe = P[e <= n ? e : e - n];
            const double sum = (e <= n) ? (pe - pq) : (pn_ - pq + pe);
            if (sum <= lo_b[k] || sum >= hi_b[k]) {
                const double d = ((double)totalf - sum) / c1[k] - sum / c2[k];
                const double fit = d * d;
                if (fit > mine[k].fit) {  // q ascending per thread
                    if (mine[k].q != 0x7fffffff) { mine[k].second = mine[k].fit; mine[k].q2 = mine[k].q; }
                    mine[k].fit = fit;
                    mine[k].q = q;
                } else if (fit > mine[k].second) {
                    mine[k].second = fit;
                    mine[k].q2 = q;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        FitBest b = mine[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            FitBest other;
            other.fit = __shfl_down(b.fit, o, 64);
            other.q = __shfl_down(b.q, o, 64);
            other.q2 = __shfl_down(b.q2, o, 64);
            other.second = __shfl_down(b.second, o, 64);
            b = better(b, other);
        }
        if (lane == 0) S.wbest[k][wave] = b;
    }
    __syncthreads();
    if (tid < 5) {
        FitBest b = S.wbest[tid][0];
#pragma unroll
        for (int w = 1; w < SYNC_T / 64; w++) b = better(b, S.wbest[tid][w]);
        S.best[tid] = b;
    }
    __syncthreads();
}

struct SpecEntry {
    FitBest best[5];
};

// Speculative, fully parallel part of the sync detector: the strip size a frame starts from is
// the previous frame's result, but it settles within a few frames and then stays — so search
// every frame of the batch for the candidate sizes around the size the batch STARTS with.
// k_sync_chain consumes these results while the prediction holds and searches on its own
// where it does not.
__global__ __launch_bounds__(SYNC_T) void k_sync_search(int W, int H, StripScratch sc, const PpState *__restrict__ state,
                                                        SpecEntry *__restrict__ spec, const int *__restrict__ redo,
                                                        const int *__restrict__ only)
{
    __shared__ SearchShared S;
    if (redo && !*redo) return;
    const int axis = blockIdx.x, f = blockIdx.y;
    // second run: the batch starts from the same state, so the speculation of every frame whose strip did not
    // change still stands
    if (only && !only[f * 2 + axis]) return;
    const int n = axis == 0 ? W : H;
    int minsize = axis == 0 ? (int)(W * 0.05f) : (int)(H * 0.01f);  // syncdetector.c:178-179
    if (minsize < 1) minsize = 1;
    int sizes[5];
    sync_sizes(axis == 0 ? state->strip_x : state->strip_y, minsize, n >> 1, sizes);
    strip_search(S, sc.prefix + ((long long)f * 2 + axis) * (sc.nmax + 1), n, sizes);
    if (threadIdx.x < 5) spec[f * 2 + axis].best[threadIdx.x] = S.best[threadIdx.x];
}

// The sequential part (syncdetector.c:95-119,133-153): pick the winning size, move dx with its
// low-pass, run the framerate PLL — a scalar recurrence that lane 0 walks frame by frame using
// the speculative search results; a frame whose starting size differs from the prediction is
// searched here by the whole workgroup.
struct ChainShared {
    int f, cur, dx, vx;
};

// What findthesweetspot (syncdetector.c:60-119) decides for one frame from the per-size search results `res`
// (sizes[] as sync_sizes gives them, cc = the clamped starting size): the winning size and window, where the two
// 1024.0 markers go, the centre of the blanking band, and — when asked — whether the reference's own rounding
// could have chosen otherwise (toss-up; see k_sync_chain).
struct SyncDecision {
    int beststart, bestsize, mark2, centre, toss;
};
__device__ __forceinline__ SyncDecision sync_decide(const FitBest *res, const int sizes[5], int cc, int n, double total, bool want_toss)
{
    double bestfit = -1.0;
    int bestq = 0, bestsize = cc, bestk = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (sizes[k] <= 0) continue;
        // sizes are tried in the reference's order with its strict `>` (k = 0 always taken)
        if (k == 0 || res[k].fit > bestfit) { bestfit = res[k].fit; bestq = res[k].q; bestsize = sizes[k]; bestk = k; }
    }
    // No window chosen: every fit of this size was NaN — a strip of ONE entry (n - size = 0: 0/0), or non-finite sums — and NaN
    // beats nothing.  The reference starts from window 0 and replaces it only by a LARGER fit (syncdetector.c:36-38,52-55), so it
    // keeps window 0; the search's "none yet" label must not reach the marker stores below (it did: a write 8 GB past the
    // strip, the GPU memory fault one-row frames ended in until round 5).
    if (bestq == 0x7fffffff) bestq = 0;
    // A strip total that is not finite — a NaN or an infinite entry, or sums beyond float's range, `totalsum` being narrowed to
    // float (syncdetector.c:26) — makes every window's fit NaN or +Inf, window 0's included, and `bestfitcurr > *bestfit` /
    // `bestfit_temp > bestfit` (syncdetector.c:52,63) are false against either: the reference keeps window 0 of the current size.
    // (The parallel search would take the first window whose fit is +Inf, which is not window 0 when that one's is NaN.)
    const float totalf = (float)total;
    const bool total_finite = fabsf(totalf) <= 3.402823466e38f;  // false for NaN and for +-Inf
    if (!total_finite) { bestq = 0; bestsize = cc; bestk = 0; }
    SyncDecision d;
    d.toss = 0;
    if (want_toss && total_finite) {
        // Would the reference's rounding have chosen otherwise?  Its strip entries (sequential f32 sums of
        // ~10^3 pixels) carry ~7e-7 of relative error each, independently; two windows differ in m
        // entries, so their sums move against each other by ~sqrt(m)*7e-7*entry, the mean difference d
        // by that times (1/rest + 1/strip), and fit = d*d by 2|d| times that.  Four sigmas.
        const double entry = fabs(total) / (double)n;
        const double sd = sqrt(bestfit > 0.0 ? bestfit : 0.0);
        const double per = 1.0 / (double)(n - bestsize) + 1.0 / (double)bestsize;
        int toss = 0;
        if (res[bestk].second >= 0.0) {
            int shift = res[bestk].q2 - bestq;
            if (shift < 0) shift = -shift;
            if (shift > n - shift) shift = n - shift;
            const int m = 2 * (shift < bestsize ? shift : bestsize);
            const double tol = 8.0 * sd * sqrt((double)m) * 7e-7 * entry * per;
            toss |= !(bestfit - res[bestk].second > tol);
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (k != bestk && sizes[k] > 0) {
                const double tol = 8.0 * sd * sqrt((double)(bestsize + sizes[k])) * 7e-7 * entry * per;
                toss |= !(fabs(bestfit - res[k].fit) > tol);
            }
        d.toss = toss;
    }
    // window start q carries the label of the index just removed (q-1); start 0 is labelled 0
    d.beststart = bestq > 0 ? bestq - 1 : 0;
    d.bestsize = bestsize;
    d.mark2 = (d.beststart + bestsize) % n;
    d.centre = (d.beststart + bestsize / 2) % n;
    return d;
}

// First run of a batch (redo == nullptr): starts from `state`, leaves a copy of that starting state in `saved`
// and marks in amb[f*2+axis] every decision whose margin over the runner-up (another window position of the
// chosen size, or another size) is inside the rounding of the reference's own f32 strip sums — those could
// fall the other way there.  Second run (redo != nullptr, after k_redo_prepare made the marked frames' strips
// exact): does nothing unless *redo, else starts again from `saved`.
__global__ __launch_bounds__(SYNC_T) void k_sync_chain(int F, int W, int H, StripScratch sc, PpState *__restrict__ state,
                                                       ChainOut *__restrict__ out, const SpecEntry *__restrict__ spec,
                                                       int pll_enabled, PpState *__restrict__ saved, int *__restrict__ amb,
                                                       const int *__restrict__ redo)
{
    __shared__ SearchShared S;
    __shared__ ChainShared C;
    if (redo && !*redo) return;
    const bool xblock = blockIdx.x == 0;
    const int axis = blockIdx.x;
    const int n = xblock ? W : H;
    const int tid = threadIdx.x;
    int minsize = xblock ? (int)(W * 0.05f) : (int)(H * 0.01f);
    if (minsize < 1) minsize = 1;
    const double lowpass = xblock ? 0.9 : 0.1;
    const int half = n >> 1;

    const PpState *from = redo ? saved : state;
    int dx = xblock ? from->dx_x : from->dx_y;
    int vx = xblock ? from->vx_x : from->vx_y;
    int cur = xblock ? from->strip_x : from->strip_y;
    double avg_speed = from->avg_speed;
    int locked = from->locked;
    if (!redo && tid == 0) {  // each block saves the fields it owns
        if (xblock) {
            saved->dx_x = dx; saved->vx_x = vx; saved->strip_x = cur;
            saved->locked = locked; saved->avg_speed = avg_speed;
        } else {
            saved->dx_y = dx; saved->vx_y = vx; saved->strip_y = cur;
        }
    }
    int pred[5];
    const int cur0 = sync_sizes(cur, minsize, half, pred);  // what k_sync_search assumed
    const bool want_toss = !redo && amb;

    // Two phases per stage of CHAIN_STAGE frames.  A (one thread per frame): everything that follows from a
    // frame's search results alone once the starting size is the predicted one — the winner among the candidate
    // sizes, the toss-up test with its f64 square roots and divisions, the markers, the blanking centre.  B (lane
    // 0): the recurrence proper (dx low-pass, vx, the PLL) — a few dozen dependent instructions per frame, which is
    // what a scalar walk on a vector unit can afford; a frame whose starting size is not the predicted one is
    // searched by the whole workgroup and decided in B.
    __shared__ SyncDecision sdec[CHAIN_STAGE];
    int f = 0;
    bool have_search = false;  // S.best holds this frame's own search
    for (int fbase = 0; fbase < F; fbase += CHAIN_STAGE) {
        const int flimit = (F - fbase < CHAIN_STAGE) ? F : fbase + CHAIN_STAGE;
        __syncthreads();  // the previous stage's entries are no longer read
        for (int i = tid; i < flimit - fbase; i += SYNC_T)
            sdec[i] = sync_decide(spec[(fbase + i) * 2 + axis].best, pred, cur0, n, sc.total[(fbase + i) * 2 + axis], want_toss);
        __syncthreads();
        while (f < flimit) {
            if (tid == 0) {
                for (; f < flimit; f++) {
                    const int cc = cur < minsize ? minsize : (cur > half ? half : cur);  // syncdetector.c:76-77
                    SyncDecision d;
                    if (have_search) {
                        int sizes[5];
                        sync_sizes(cur, minsize, half, sizes);
                        d = sync_decide(S.best, sizes, cc, n, sc.total[f * 2 + axis], want_toss);
                        have_search = false;
                    } else if (cc != cur0) {
                        break;  // misprediction: the workgroup searches frame f
                    } else {
                        d = sdec[f - fbase];
                    }
                    if (want_toss) amb[f * 2 + axis] = d.toss;
                    float *gblur = sc.blur + ((long long)f * 2 + axis) * sc.nmax;
                    gblur[d.beststart] = PIX_B;  // syncdetector.c:98-99
                    gblur[d.mark2] = PIX_B;

                    const int h2 = n / 2;
                    int centre = d.centre;
                    int ndx = dx;
                    const int rawdiff = centre - ndx;
                    if (rawdiff > h2) ndx += n;
                    else if (rawdiff < -h2) centre += n;
                    const int last = ndx;
                    // operands are far below 2^31: the reference's int64 round-and-modulo in 32 bits; both terms
                    // lie in [0, 2n), so `% n` is one conditional subtraction (the general form stays for safety)
                    ndx = (int)round(centre * lowpass + (1.0 - lowpass) * ndx);
                    if (ndx >= n) ndx -= n;
                    if (ndx >= n || ndx < 0) ndx %= n;
                    const int rawvx = ndx - last;
                    vx = (rawvx > h2) ? (n - rawvx) : ((rawvx < -h2) ? (-n - rawvx) : rawvx);
                    dx = ndx;
                    cur = d.bestsize;
                    ChainOut *o = &out[f];
                    if (xblock) {
                        // frameratepll, syncdetector.c:133-153
                        avg_speed = avg_speed * 0.99 + 0.01 * vx;
                        locked = (avg_speed < 0.5 && avg_speed > -0.5) ? 1 : 0;
                        int fired = 0;
                        double diff = 0.0;
                        if (pll_enabled && vx != 0) {
                            diff = locked ? (avg_speed * 0.000001) : (vx * 0.00001);
                            fired = 1;
                        }
                        o->dx = dx; o->vx = vx; o->stripx = cur;
                        o->locked = locked; o->pll_fired = fired;
                        o->avg_speed = avg_speed; o->frameratediff = diff;
                    } else {
                        o->dy = dx; o->vy = vx; o->stripy = cur;
                    }
                }
                C.f = f;
                C.cur = cur;
            }
            __syncthreads();
            f = C.f;
            if (f >= flimit) break;
            // frame f starts from a size the speculation did not cover
            int sizes[5];
            sync_sizes(C.cur, minsize, half, sizes);
            strip_search(S, sc.prefix + ((long long)f * 2 + axis) * (sc.nmax + 1), n, sizes);
            have_search = true;  // only lane 0's copy matters
        }
    }
    if (tid == 0) {
        if (xblock) {
            state->dx_x = dx; state->vx_x = vx; state->strip_x = cur;
            state->locked = locked; state->avg_speed = avg_speed;
        } else {
            state->dx_y = dx; state->vx_y = vx; state->strip_y = cur;
        }
    }
}

// ---------------------------------------------------------------------------
// k_frame_pass: elementwise work of dsp_post_process for F frames.
//   v = src_f[map(p)]                       map: identity | 2-D roll (syncdetector.c:187-207)
//   v = sentinel(v) ? v : (v-lastmin)/span  if NORMALISE      (dsp.c:74)
//   v = 512 on column dx / row dy           if LINES          (syncdetector.c:209-223)
//   s = s*a + v*(1.0-a)                     if IIR            (dsp.c:29-32, mixed f32/f64)
//   dst_f[p] = IIR ? s : v
// Each thread keeps its pixels' IIR state in registers across the F frames.
// ---------------------------------------------------------------------------
#define PASS_NORMALISE 1
#define PASS_ROLL 2
#define PASS_LINES 4
#define PASS_IIR 8

template <int VW> struct VecT;
template <> struct VecT<2> { typedef float2_a4 type; };
template <> struct VecT<4> { typedef float4_a4 type; };

__device__ __forceinline__ float pass_one(int flags, float v, float &s, float a, double one_minus_a, float lastmin, float span,
                                          bool on_line)
{
    if (flags & PASS_NORMALISE) {
        // the frame's divisor prepared once (NormDiv; lastmin and span are uniform and loop invariant, so the set-up is hoisted
        // out of the caller's pixel loop), `/` where the guard does not hold
        const NormDiv nd = norm_div_setup(lastmin, span);
        const float q = nd.ok ? norm_div(nd, v - lastmin) : ((v - lastmin) / span);
        v = (v > 250.0f || v < -250.0f) ? v : q;
    }
    if ((flags & PASS_LINES) && on_line) v = PIX_G;
    if (flags & PASS_IIR) {
        s = (float)((double)(s * a) + (double)v * one_minus_a);
        v = s;
    }
    return v;
}

// VW consecutive pixels per thread (one VW*4-byte load/store per frame when the layout allows it),
// two frames per loop trip so that several loads are in flight per lane.
template <int FLAGS, int VW>
__global__ __launch_bounds__(256) void k_frame_pass(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                    long long dstride, int F, int W, int H,
                                                    const ChainOut *__restrict__ chain, float *__restrict__ screen, float a,
                                                    const int *__restrict__ gate)
{
    typedef typename VecT<VW>::type vec_t;
    if (gate && !*gate) return;  // (the exact redo behind k_frame_pass_par: launched always, needed almost never)
    const int P = W * H;
    const double one_minus_a = 1.0 - a;
    const int ngroups = (P + VW - 1) / VW;
    // consecutive spans go to the same XCD (see k_frame_stats): frames are not 128-byte aligned, so
    // neighbouring workgroups share a cache line at each end of their span
    const unsigned gx = gridDim.x;
    const unsigned lb = (gx % 8u == 0u) ? (blockIdx.x % 8u) * (gx / 8u) + blockIdx.x / 8u : blockIdx.x;
    for (int grp = lb * blockDim.x + threadIdx.x; grp < ngroups; grp += gridDim.x * blockDim.x) {
        const int p0 = grp * VW;
        int x[VW], y[VW];
        float s[VW];
#pragma unroll
        for (int k = 0; k < VW; k++) {
            const int p = p0 + k;
            y[k] = p / W;
            x[k] = p - y[k] * W;
            s[k] = ((FLAGS & PASS_IIR) && p < P) ? screen[p] : 0.f;
        }
        const bool full = (p0 + VW <= P);
#pragma unroll 2
        for (int f = 0; f < F; f++) {
            const float *in = src + (long long)f * sstride;
            float *outp = dst + (long long)f * dstride;
            int dx = 0, dy = 0;
            float lastmin = 0.f, span = 1.f;
            if (FLAGS & (PASS_ROLL | PASS_LINES)) { dx = chain[f].dx; dy = chain[f].dy; }
            if (FLAGS & PASS_NORMALISE) { lastmin = chain[f].lastmin; span = chain[f].span; }
            float v[VW];
            if (!(FLAGS & PASS_ROLL) && VW > 1 && full) {
                const vec_t t = *reinterpret_cast<const vec_t *>(in + p0);
#pragma unroll
                for (int k = 0; k < VW; k++) v[k] = t[k];
            } else {
#pragma unroll
                for (int k = 0; k < VW; k++) {
                    int sp = p0 + k;
                    if (FLAGS & PASS_ROLL) {
                        int sx = x[k] + dx; if (sx >= W) sx -= W;
                        int sy = y[k] + dy; if (sy >= H) sy -= H;
                        sp = sy * W + sx;
                    }
                    v[k] = (p0 + k < P) ? in[sp] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < VW; k++) v[k] = pass_one(FLAGS, v[k], s[k], a, one_minus_a, lastmin, span, x[k] == dx || y[k] == dy);
            if (VW > 1 && full) {
                vec_t t;
#pragma unroll
                for (int k = 0; k < VW; k++) t[k] = v[k];
                *reinterpret_cast<vec_t *>(outp + p0) = t;
            } else {
#pragma unroll
                for (int k = 0; k < VW; k++)
                    if (p0 + k < P) outp[p0 + k] = v[k];
            }
        }
        if (FLAGS & PASS_IIR) {
#pragma unroll
            for (int k = 0; k < VW; k++)
                if (p0 + k < P) screen[p0 + k] = s[k];
        }
    }
}

// The same pass with one thread per FOUR PIXELS OF ONE FRAME: every frame is one contiguous stream (a thread of
// k_frame_pass walks the F frames of its four pixels, 13 MB apart) — 0.327 -> 0.298 ms for 60 frames of 2962x1125.
// For flag sets without the IIR, and for the IIR with motion blur 0: then s*a + v*(1-a) (dsp.c:29-32, f32*f32 in f32,
// the rest in f64) is v itself whenever s is finite and v is not -0.0, so a frame's output does not depend on the
// previous frame's, and the last frame's output is the new state.  The exceptions are caught, not assumed away: any
// non-finite incoming state or output (a NaN sticks to its pixel for good in the reference) or a -0.0 output raises
// *odd, and the frame-by-frame kernel, queued behind this one and gated on *odd, redoes the batch literally;
// k_pass_state copies the last frame into the state only when *odd stayed 0.  grid (spans, F).
template <int FLAGS>
__global__ __launch_bounds__(256) void k_frame_pass_par(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                        long long dstride, int W, int H, const ChainOut *__restrict__ chain,
                                                        const float *__restrict__ screen, int *__restrict__ odd)
{
    typedef typename VecT<4>::type vec_t;
    const int P = W * H;
    const int ngroups = (P + 3) / 4;
    const int f = blockIdx.y;
    const unsigned gx = gridDim.x;
    const unsigned lb = (gx % 8u == 0u) ? (blockIdx.x % 8u) * (gx / 8u) + blockIdx.x / 8u : blockIdx.x;
    const float *in = src + (long long)f * sstride;
    float *outp = dst + (long long)f * dstride;
    int dx = 0, dy = 0;
    float lastmin = 0.f, span = 1.f;
    if (FLAGS & (PASS_ROLL | PASS_LINES)) { dx = chain[f].dx; dy = chain[f].dy; }
    if (FLAGS & PASS_NORMALISE) { lastmin = chain[f].lastmin; span = chain[f].span; }
    bool bad = false;
    for (int grp = lb * blockDim.x + threadIdx.x; grp < ngroups; grp += gx * blockDim.x) {
        const int p0 = grp * 4;
        const bool full = (p0 + 4 <= P);
        int x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
        if (FLAGS & (PASS_ROLL | PASS_LINES)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = p0 + k;
                y[k] = p / W;
                x[k] = p - y[k] * W;
            }
        }
        float v[4];
        if (!(FLAGS & PASS_ROLL) && full) {
            const vec_t t = *reinterpret_cast<const vec_t *>(in + p0);
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int sp = p0 + k;
                if (FLAGS & PASS_ROLL) {
                    int sx = x[k] + dx; if (sx >= W) sx -= W;
                    int sy = y[k] + dy; if (sy >= H) sy -= H;
                    sp = sy * W + sx;
                }
                v[k] = (p0 + k < P) ? in[sp] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float none = 0.f;
            v[k] = pass_one(FLAGS & ~PASS_IIR, v[k], none, 0.f, 1.0, lastmin, span, x[k] == dx || y[k] == dy);
            if (FLAGS & PASS_IIR) bad |= !(fabsf(v[k]) <= 3.4028234664e38f) || __float_as_uint(v[k]) == 0x80000000u;
        }
        if ((FLAGS & PASS_IIR) && f == 0) {  // the incoming state only has to be finite
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (p0 + k < P) bad |= !(fabsf(screen[p0 + k]) <= 3.4028234664e38f);
        }
        if (full) {
            vec_t t;
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = v[k];
            *reinterpret_cast<vec_t *>(outp + p0) = t;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (p0 + k < P) outp[p0 + k] = v[k];
        }
    }
    if ((FLAGS & PASS_IIR) && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(odd, 1);
}

// the new IIR state behind k_frame_pass_par: the last frame's output, unless the batch has to be redone
__global__ __launch_bounds__(256) void k_pass_state(const float *__restrict__ last, float *__restrict__ screen, int P, const int *__restrict__ odd)
{
    if (*odd) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) screen[p] = last[p];
}

// ---------------------------------------------------------------------------
// Fused run (tsdrgpu_postproc_begin_minmax, default stage order without roll): the autogain's
// min/max come from the resampler, so normalise + IIR can run BEFORE the sync detector, and the
// one trip over the raw frames that does it also produces the row/column partial sums the sync
// detector needs — k_frame_stats' read of every frame disappears (16P -> 12P bytes per frame).
// Workgroup = one 256x32 tile for all F frames (IIR state of its 32 pixels per thread stays in
// registers); the statistics part is k_frame_stats' code, so the strips are bit-identical.
// The green lines (syncdetector.c:209-223) need dx/dy, which are known only afterwards:
// k_fix_lines replays the exact per-pixel recurrence over the batch for the few pixels that lie on
// a line in any frame (reading the pre-batch IIR state, which is why the state is double buffered).
// ---------------------------------------------------------------------------
// LDS-only workgroup barrier: __syncthreads() also waits for every outstanding global load/store
// (s_waitcnt vmcnt(0)), which would serialise the prefetch of the next frame behind this frame's stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define TP_WAVES 8  // 512 threads per tile
#define TP_H 16     // tile = 256 columns x 16 rows: wave w owns rows w and w+8, lane l columns l+64j
// 8 pixels per thread keep the kernel under 64 VGPRs, so four workgroups (32 waves) fit a CU and every
// tile of a 1080p frame is resident at once: the waves of other tiles hide a tile's load latency and
// its two LDS barriers per frame.  Accesses are unconditional (pixels outside the frame read pixel 0
// and store to a dump area): no divergent branches around memory operations, exact s_waitcnt counts.
__global__ __launch_bounds__(512, 7) void k_frame_tile_pass(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                            long long dstride, int F, int W, int H, int tiles_x, int tiles_y,
                                                            const ChainOut *__restrict__ chain, const float *__restrict__ screen_in,
                                                            float *__restrict__ screen_out, float a, float *__restrict__ colp,
                                                            float *__restrict__ rowp, int *__restrict__ tflag, float *__restrict__ dump)
{
    const unsigned total = gridDim.x;
    const unsigned l = blockIdx.x;
    const unsigned logical = (total % 8u == 0u) ? (l % 8u) * (total / 8u) + l / 8u : l;  // see k_frame_stats
    const int tx = logical % tiles_x, ty = logical / tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = tx * TILE_W, y0 = ty * TP_H;
    const double one_minus_a = 1.0 - a;
    constexpr int ROWS = TP_H / TP_WAVES;  // 2
    constexpr int FL = PASS_NORMALISE | PASS_IIR;
    float *const mydump = dump + threadIdx.x;

    const int base = (y0 + wave) * W + x0 + lane;
    bool rowok[ROWS], colok[4];
#pragma unroll
    for (int r = 0; r < ROWS; r++) rowok[r] = (y0 + wave + TP_WAVES * r) < H;  // wave-uniform
#pragma unroll
    for (int j = 0; j < 4; j++) colok[j] = (x0 + lane + 64 * j) < W;
#define TP_IN(r, j) (rowok[r] && colok[j])
#define TP_OFF(r, j) (TP_IN(r, j) ? base + (r) * TP_WAVES * W + 64 * (j) : 0)  /* 0 for pixels outside the frame */
    float st[ROWS][4];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) st[r][j] = screen_in[TP_OFF(r, j)];
    __shared__ float sh[3][TP_WAVES][TILE_W];  // 24 KiB
    __shared__ int wsent[2][TP_WAVES];
    for (int f = 0; f < F; f++) {
        const float *in = src + (long long)f * sstride;
        float *outp = dst + (long long)f * dstride;
        const float lastmin = chain[f].lastmin, span = chain[f].span;
        float val[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) val[r][j] = in[TP_OFF(r, j)];
        float cns[4] = {0, 0, 0, 0};
        bool has_sent = false;
        float *const rp = rowp + ((long long)(f * tiles_x + tx) * 3) * H;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int y = y0 + wave + TP_WAVES * r;
            float rns = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float v = val[r][j];
                const bool sent = (v > 250.0f) || (v < -250.0f);
                const float vn = (TP_IN(r, j) && !sent) ? v : 0.f;
                cns[j] += vn;
                rns += vn;
                has_sent |= TP_IN(r, j) && sent;
                float *t = TP_IN(r, j) ? outp + TP_OFF(r, j) : mydump;
                *t = pass_one(FL, v, st[r][j], a, one_minus_a, lastmin, span, false);
            }
            rns = wave_sum(rns);
            float *t = (lane == 0 && y < H) ? rp + y : mydump;
            *t = rns;
        }
        // sentinel pixels are rare: their sums are only formed (from the values still in registers)
        // by waves / tiles that hold any
        const int wave_sent = __any(has_sent) ? 1 : 0;
        if (lane == 0) wsent[f & 1][wave] = wave_sent;
#pragma unroll
        for (int j = 0; j < 4; j++) sh[0][wave][lane + 64 * j] = cns[j];
        if (wave_sent) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float cs = 0.f, cc = 0.f;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const float v = val[r][j];
                    const bool se = TP_IN(r, j) && ((v > 250.0f) || (v < -250.0f));
                    cs += se ? v : 0.f;
                    cc += se ? 1.f : 0.f;
                }
                sh[1][wave][lane + 64 * j] = cs;
                sh[2][wave][lane + 64 * j] = cc;
            }
        }
        lds_barrier();
        int tile_sent = 0;
#pragma unroll
        for (int w = 0; w < TP_WAVES; w++) tile_sent |= wsent[f & 1][w];
        if (tile_sent) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int y = y0 + wave + TP_WAVES * r;
                float rs = 0.f, rc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = val[r][j];
                    const bool se = TP_IN(r, j) && ((v > 250.0f) || (v < -250.0f));
                    rs += se ? v : 0.f;
                    rc += se ? 1.f : 0.f;
                }
                rs = wave_sum(rs);
                rc = wave_sum(rc);
                if (lane == 0 && y < H) {
                    rp[H + y] = rs;
                    rp[2 * H + y] = rc;
                }
            }
        }
        if (threadIdx.x < TILE_W) {
            const int x = x0 + threadIdx.x;
            float *cp = colp + ((long long)(f * tiles_y + ty) * 3) * W;
            const int nq = tile_sent ? 3 : 1;
            for (int q = 0; q < nq; q++) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < TP_WAVES; w++) {
                    // waves without sentinels did not write their q = 1, 2 rows
                    const float t = (q == 0 || wsent[f & 1][w]) ? sh[q][w][threadIdx.x] : 0.f;
                    acc += t;
                }
                float *t = (x < W) ? cp + q * W + x : mydump;
                *t = acc;
            }
            if (threadIdx.x == 0) tflag[(long long)f * tiles_x * tiles_y + (long long)ty * tiles_x + tx] = tile_sent;
        }
        lds_barrier();  // sh[] consumed
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float *t = TP_IN(r, j) ? screen_out + TP_OFF(r, j) : mydump;
            *t = st[r][j];
        }
}
#undef TP_IN
#undef TP_OFF

// grid (ceil(max(W,H)/256), 2, F): the column (axis 0) / row (axis 1) that frame g paints, unless an
// earlier frame of the batch paints the same one (then that frame's workgroups do the work)
__global__ __launch_bounds__(256) void k_fix_lines(const float *__restrict__ src, long long sstride, float *__restrict__ dst,
                                                   long long dstride, int F, int W, int H, const ChainOut *__restrict__ chain,
                                                   const float *__restrict__ screen_in, float *__restrict__ screen_out, float a)
{
    const int axis = blockIdx.y, g = blockIdx.z;
    const int mine = axis == 0 ? chain[g].dx : chain[g].dy;
    for (int e = 0; e < g; e++)
        if ((axis == 0 ? chain[e].dx : chain[e].dy) == mine) return;  // uniform over the workgroup
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = axis == 0 ? mine : i, y = axis == 0 ? i : mine;
    if (x < 0 || x >= W || y < 0 || y >= H) return;
    const long long pix = (long long)y * W + x;
    const double one_minus_a = 1.0 - a;
    float s = screen_in[pix];
    for (int f = 0; f < F; f++) {
        const ChainOut c = chain[f];
        const float v = src[(long long)f * sstride + pix];
        dst[(long long)f * dstride + pix] =
            pass_one(PASS_NORMALISE | PASS_LINES | PASS_IIR, v, s, a, one_minus_a, c.lastmin, c.span, x == c.dx || y == c.dy);
    }
    screen_out[pix] = s;
}

// The flat fused run (motion blur 0): a frame's painted column / row is PIX_G whatever lay underneath and whatever the
// old state was, as long as that state is finite ((float)((double)(s * 0) + 512.0) = 512) — and a non-finite state has
// raised the redo flag, which replaces the whole batch.  grid (ceil(max(W,H)/256), 2, F): axis 0 the column dx, axis 1
// the row dy of frame blockIdx.z.
__global__ __launch_bounds__(256) void k_paint_lines(float *__restrict__ dst, long long dstride, int W, int H, const ChainOut *__restrict__ chain)
{
    const int axis = blockIdx.y, f = blockIdx.z;
    const int at = axis == 0 ? chain[f].dx : chain[f].dy;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = axis == 0 ? at : i, y = axis == 0 ? i : at;
    if (x < 0 || x >= W || y < 0 || y >= H) return;
    dst[(long long)f * dstride + (long long)y * W + x] = PIX_G;
}

typedef void (*pass_fn)(const float *, long long, float *, long long, int, int, int, const ChainOut *, float *, float, const int *);
typedef void (*pass_par_fn)(const float *, long long, float *, long long, int, int, const ChainOut *, const float *, int *);
static pass_par_fn pick_pass_par(int flags)
{
    switch (flags) {
#define CASE(f) case f: return k_frame_pass_par<f>;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
#undef CASE
    }
    return nullptr;
}

template <int VW>
static pass_fn pick_pass_vw(int flags)
{
    switch (flags) {
#define CASE(f) case f: return k_frame_pass<f, VW>;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
#undef CASE
    }
    return nullptr;
}

static pass_fn pick_pass(int flags, int vw)
{
    return vw == 4 ? pick_pass_vw<4>(flags) : nullptr;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static void gaussian_taps(float taps[5])
{
    // gaussian.c:17-28 — CALC_GAUSSCOEFF(5,i) = expf(-2.0f*1.0f*1.0f*i*i/(5*5)), textual expansion
    const float e2 = expf(-2.0f * 1.0f * 1.0f * -2 * -2 / (5 * 5));
    const float e1 = expf(-2.0f * 1.0f * 1.0f * -1 * -1 / (5 * 5));
    const float e0 = expf(-2.0f * 1.0f * 1.0f * 0 * 0 / (5 * 5));
    const float f1 = expf(-2.0f * 1.0f * 1.0f * 1 * 1 / (5 * 5));
    const float f2 = expf(-2.0f * 1.0f * 1.0f * 2 * 2 / (5 * 5));
    const float norm = e2 + e1 + e0 + f1 + f2;
    taps[0] = e2 / norm; taps[1] = e1 / norm; taps[2] = e0 / norm; taps[3] = f1 / norm; taps[4] = f2 / norm;
}

extern "C" int tsdrgpu_postproc_create(tsdrgpu_t *g, tsdrgpu_postproc_t **out)
{
    if (!g || !out) return TSDRGPU_EINVAL;
    tsdrgpu_postproc_t *pp = (tsdrgpu_postproc_t *)calloc(1, sizeof(*pp));
    if (!pp) return TSDRGPU_ENOMEM;
    pp->g = g;
    gaussian_taps(pp->taps);
    if (hipMalloc(&pp->d_state, 2 * sizeof(PpState)) != hipSuccess) { free(pp); return TSDRGPU_ENOMEM; }  // [1]: the sync chain's batch-start copy
    if (hipEventCreateWithFlags(&pp->ev_stats, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&pp->ev_chain, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(pp->d_state);
        free(pp);
        return TSDRGPU_EHIP;
    }
    pp->exact_ties = 1;  // contract-exact sync decisions by default (tsdrgpu_postproc_set_exact_ties(pp, 0) opts out)
    *out = pp;
    return tsdrgpu_postproc_reset(pp);
}

extern "C" void tsdrgpu_postproc_destroy(tsdrgpu_postproc_t *pp)
{
    if (!pp) return;
    (void)hipStreamSynchronize(pp->g->stream2);
    (void)hipStreamSynchronize(pp->g->stream);
    (void)hipEventDestroy(pp->ev_stats);
    (void)hipEventDestroy(pp->ev_chain);
    void *bufs[] = {pp->d_state, pp->d_odd, pp->d_screen, pp->d_screen2, pp->d_dump, pp->d_tmp1, pp->d_tmp2, pp->d_bmin, pp->d_bmax, pp->d_tflag, pp->d_colp, pp->d_rowp,
                    pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_work, pp->d_chain, pp->d_sflag, pp->d_exact,
                    pp->d_xsum, pp->d_xmax, pp->d_v0, pp->d_chain_band, pp->d_items, pp->d_relay, pp->d_gather, pp->d_bedges, pp->d_snr_part, pp->d_snr};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (pp->h_chain) (void)hipHostFree(pp->h_chain);
    if (pp->d_state_save) (void)hipFree(pp->d_state_save);
    if (pp->h_spec_flags) (void)hipHostFree(pp->h_spec_flags);
    if (pp->ev_spec) (void)hipEventDestroy(pp->ev_spec);
    free(pp->h_flags);
    free(pp);
}

extern "C" int tsdrgpu_postproc_reset(tsdrgpu_postproc_t *pp)
{
    if (!pp) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending == 1 || pp->pending == 3 || pp->pending == 4)
        HIP_TRY(g, hipStreamWaitEvent(g->stream, pp->ev_chain, 0));  // an abandoned split / fused run: its side-lane work first
    pp->pending = 0;
    pp->band_stage = 0;  // (an abandoned band run, speculated or fused, leaves nothing behind either)
    pp->band_spec = pp->band_fused = pp->band_flat = 0;
    // dsp_post_process_init (dsp.c:112-132): autogain 0/0, sync detector zeroed, sizes forgotten
    HIP_TRY(g, hipMemsetAsync(pp->d_state, 0, sizeof(PpState), g->stream));
    pp->width = pp->height = 0;
    pp->lowpass_before_sync = 0;
    if (pp->d_screen) HIP_TRY(g, hipMemsetAsync(pp->d_screen, 0, pp->cap_screen * sizeof(float), g->stream));
    return TSDRGPU_OK;
}

template <class T>
static int ensure(tsdrgpu_t *g, T **buf, size_t *cap, size_t need, bool zero = false)
{
    if (*cap >= need && *buf) return TSDRGPU_OK;
    if (*buf) {
        (void)hipStreamSynchronize(g->stream);
        (void)hipFree(*buf);
    }
    *buf = nullptr;
    *cap = 0;
    if (hipMalloc((void **)buf, need * sizeof(T)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "device scratch");
    *cap = need;
    if (zero) HIP_TRY(g, hipMemsetAsync(*buf, 0, need * sizeof(T), g->stream));
    return TSDRGPU_OK;
}

// Between the two runs of the sync chain: frames with a toss-up decision whose strips are not yet the
// reference's own get them (sflag), and *redo tells the second run whether there is anything to do.
__global__ __launch_bounds__(256) void k_redo_prepare(int count, const int *__restrict__ amb, int *__restrict__ sflag, int *__restrict__ redo,
                                                      int *__restrict__ fresh)
{
    int any = 0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int now = (amb[i] && !sflag[i]) ? 1 : 0;
        fresh[i] = now;  // strips that change between the two runs
        if (now) {
            sflag[i] = 1;
            any = 1;
        }
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) *redo = any;
}

// the big pass over the frames (per-tile partials) ...
static int launch_stats_tiles(tsdrgpu_postproc_t *pp, const float *frames, long long fstride, int F, int W, int H, int want_strips)
{
    tsdrgpu_t *g = pp->g;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
    const unsigned grid = (unsigned)tiles_x * tiles_y * F;
    TSDR_LAUNCH(g, PROF_FRAME_STATS, g->stream, k_frame_stats<false>, grid, 256, frames, fstride, W, H, tiles_x, tiles_y, pp->d_bmin, pp->d_bmax, pp->d_colp,
                pp->d_rowp, pp->d_tflag, want_strips, StatsStore{});
    KERNEL_CHECK(g, "k_frame_stats");
    return TSDRGPU_OK;
}

// ... and the small fold of the partials (on `st`: the split run puts it on the side stream with the chain)
static int launch_stats_reduce(tsdrgpu_postproc_t *pp, hipStream_t st, int F, int W, int H, int want_strips)
{
    tsdrgpu_t *g = pp->g;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_frame_reduce, dim3(((W > H ? W : H) + 255) / 256, 3, F), 256, W, H, tiles_x, tiles_y, pp->d_bmin, pp->d_bmax,
                pp->d_colp, pp->d_rowp, pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_tflag, want_strips, 1, TILE_H);
    KERNEL_CHECK(g, "k_frame_reduce");
    return TSDRGPU_OK;
}

static int launch_stats(tsdrgpu_postproc_t *pp, const float *frames, long long fstride, int F, int W, int H, int want_strips)
{
    const int rc = launch_stats_tiles(pp, frames, fstride, F, W, H, want_strips);
    return rc ? rc : launch_stats_reduce(pp, pp->g->stream, F, W, H, want_strips);
}

static int launch_chain(tsdrgpu_postproc_t *pp, const float *frames, long long fstride, int F, int W, int H, int do_autogain,
                        int do_sync, int strips_normalised, const tsdrgpu_pp_params_t *prm)
{
    tsdrgpu_t *g = pp->g;
    hipStream_t st = pp->chain_st ? pp->chain_st : g->stream;
    // the autogain record (lastmin/lastmax/span per frame) is (re)written by every call: a sync-only
    // call repeats the carried state, which no later launch of that order reads
    if (do_autogain || !pp->chain_has_autogain) {
        TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, frames, fstride, pp->ext_fmin ? pp->ext_fmin : pp->d_fmin,
                    pp->ext_fmax ? pp->ext_fmax : pp->d_fmax, pp->d_state, pp->d_chain,
                                                  do_autogain, prm->lowpasscoeff, pp->clear_with_autogain);
        KERNEL_CHECK(g, "k_autogain_chain");
        pp->chain_has_autogain = 1;
    }
    if (do_autogain && pp->want_snr && !pp->band_mode) {
        // dsp_autogain_run's other result (dsp.c:69-93): mean / stdev of the frames autogain reads, beside the chain
        int rc;
        if ((rc = ensure(g, &pp->d_snr_part, &pp->cap_snr_part, (size_t)F * tsdr_snr_part_doubles()))) return rc;
        if ((rc = ensure(g, &pp->d_snr, &pp->cap_snr, (size_t)F))) return rc;
        if ((rc = tsdr_snr_batch(g, st, frames, fstride, (long long)W * H, F, pp->d_snr_part, pp->d_snr))) return rc;
        pp->snr_frames = F;
    }
    if (do_sync) {
        const int nmax = W > H ? W : H;
        StripScratch sc;
        sc.nmax = nmax;
        sc.blur = pp->d_work;
        sc.prefix = (double *)(pp->d_work + (((size_t)F * 2 * nmax + 1) & ~(size_t)1));
        sc.total = sc.prefix + (size_t)F * 2 * (nmax + 1);
        if (pp->band_mode) {
            // row-band run: a rank holds only its rows, so the literal (raster-order) re-collapse of a strip is not
            // available; every strip is taken from the exchanged f64 sums
            HIP_TRY(g, hipMemsetAsync(pp->d_sflag, 0, sizeof(int) * (size_t)F * 2, st));
        } else {
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_strip_flag, dim3(2, F), CHAIN_T, W, H, pp->d_strip_x, pp->d_strip_y, pp->d_chain, strips_normalised,
                        pp->d_sflag);
        }
        SpecEntry *spec = (SpecEntry *)(sc.total + (size_t)F * 2);
        int *d_amb = pp->d_sflag + (size_t)F * 2, *d_fresh = d_amb + (size_t)F * 2, *d_redo = d_fresh + (size_t)F * 2;
        const int *const no_gate = nullptr;
        const unsigned exact_blocks = (unsigned)(((W + 63) / 64) > ((H + 63) / 64) ? ((W + 63) / 64) : ((H + 63) / 64));
        // run 1 as speculated; with exact ties on, run 2 (five empty launches unless needed) repeats the chain for a
        // batch in which some decision was a toss-up at the precision of the strips, those frames' strips made exact
        const int runs = (pp->exact_ties && !pp->band_mode) ? 2 : 1;
        for (int run = 0; run < runs; run++) {
            const int *gate = run ? d_redo : no_gate;
            const int *only = run ? d_fresh : no_gate;
            if (run) TSDR_LAUNCH(g, PROF_CHAIN, st, k_redo_prepare, 1, 256, 2 * F, d_amb, pp->d_sflag, d_redo, d_fresh);
            if (!pp->band_mode) {
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_exact_strips, dim3(exact_blocks, 2 * F < XS_ITEMS ? 2 * F : XS_ITEMS), 256, frames, fstride, W, H,
                            pp->d_chain, strips_normalised, pp->d_sflag, pp->d_exact, nmax, gate, only, 2 * F);
                KERNEL_CHECK(g, "k_exact_strips");
            }
            TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, H, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc,
                        strips_normalised, pp->taps[0], pp->taps[1], pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, gate);
            KERNEL_CHECK(g, "k_strip_prepare");
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, H, sc, run ? pp->d_state + 1 : pp->d_state, spec, gate, only);
            KERNEL_CHECK(g, "k_sync_search");
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, H, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1,
                        (pp->exact_ties && !pp->band_mode) ? d_amb : (int *)nullptr, gate);
        }
        KERNEL_CHECK(g, "k_sync_chain");
    }
    return TSDRGPU_OK;
}

// the frame-by-frame (literal) pass; with `gate` it only runs if *gate was raised by the frame-parallel form before it
static int launch_pass_literal(tsdrgpu_postproc_t *pp, int flags, const float *src, long long sstride, float *dst, long long dstride,
                               int F, int W, int H, float a, const int *gate)
{
    tsdrgpu_t *g = pp->g;
    // four pixels per lane (one dwordx4 per frame; the vector types only claim float alignment)
    const int vw = 4;
    pass_fn fn = pick_pass(flags, vw);
    if (!fn) return tsdr_fail(g, TSDRGPU_EINVAL, "k_frame_pass", "unsupported flag combination");
    const long long P = (long long)W * H;
    long long blocks = ((P + vw - 1) / vw + 255) / 256;  // (P < 4 is a frame too: one group)
    const long long cap = (long long)g->prop.multiProcessorCount * 16;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) & ~7LL;  // a multiple of the 8 XCDs (the kernel's span order relies on it)
    TSDR_LAUNCH(g, PROF_FRAME_PASS, g->stream, fn, (unsigned)blocks, 256, src, sstride, dst, dstride, F, W, H, pp->d_chain, pp->d_screen, a, gate);
    KERNEL_CHECK(g, "k_frame_pass");
    return TSDRGPU_OK;
}

static int launch_pass(tsdrgpu_postproc_t *pp, int flags, const float *src, long long sstride, float *dst, long long dstride,
                       int F, int W, int H, float a)
{
    tsdrgpu_t *g = pp->g;
    // one stream per frame instead of one walk over the frames per pixel group, whenever a frame's output does not
    // depend on the previous one's (k_frame_pass_par) and the batch is long enough to pay for its three small launches
    static const int serial_only = getenv("TSDRGPU_PASS_SERIAL") ? 1 : 0;
    const bool iir = (flags & PASS_IIR) != 0;
    const int *gate = nullptr;
    // (with the IIR the frame-parallel form may be redone literally from `src` behind it, so `src` must still hold the
    // raw frames then: a run whose output overlaps its input takes the literal form at once)
    const bool overlap = (const float *)dst < src + (long long)F * sstride && src < (const float *)dst + (long long)F * dstride;
    if (!serial_only && F >= 8 && (!iir || (a == 0.0f && !overlap))) {
        pass_par_fn pf = pick_pass_par(flags);
        if (!pf) return tsdr_fail(g, TSDRGPU_EINVAL, "k_frame_pass_par", "unsupported flag combination");
        if (iir) {
            if (!pp->d_odd && hipMalloc(&pp->d_odd, sizeof(int)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "flag");
            HIP_TRY(g, hipMemsetAsync(pp->d_odd, 0, sizeof(int), g->stream));
        }
        const long long P = (long long)W * H;
        long long bx = ((P + 3) / 4 + 255) / 256;
        const long long cap = (long long)g->prop.multiProcessorCount * 64 / (F < 64 ? F : 64) + 8;
        if (bx > cap) bx = cap;
        bx = (bx + 7) & ~7LL;  // a multiple of the 8 XCDs (the kernel's span order relies on it)
        TSDR_LAUNCH(g, PROF_FRAME_PASS, g->stream, pf, dim3((unsigned)bx, (unsigned)F), 256, src, sstride, dst, dstride, W, H, pp->d_chain,
                    (const float *)pp->d_screen, pp->d_odd);
        KERNEL_CHECK(g, "k_frame_pass_par");
        if (!iir) return TSDRGPU_OK;
        TSDR_LAUNCH(g, PROF_FRAME_PASS, g->stream, k_pass_state, (unsigned)(g->prop.multiProcessorCount * 4), 256,
                    (const float *)(dst + (long long)(F - 1) * dstride), pp->d_screen, (int)P, (const int *)pp->d_odd);
        KERNEL_CHECK(g, "k_pass_state");
        gate = pp->d_odd;  // ... and the literal form below only runs if the flag was raised
    }
    return launch_pass_literal(pp, flags, src, sstride, dst, dstride, F, W, H, a, gate);
}

// buffers and per-run state for F frames of W x H (everything before the first launch of a run)
static int pp_prepare(tsdrgpu_postproc_t *pp, int F, int W, int H, const tsdrgpu_pp_params_t *prm)
{
    tsdrgpu_t *g = pp->g;
    const size_t P = (size_t)W * H;
    int rc;
    if (P > (size_t)TSDRGPU_MAX_FRAME_PIXELS)  // the reference's own bound (MAX_ARR_SIZE, TSDRLibrary.c:31,489); pixel indices are ints here
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc", "width*height above 4000*4000 pixels");

    // buffer (re)sizing, dsp.c:152-173: the screen buffer is zeroed only when it has to grow
    if (H != pp->height || W != pp->width) {
        pp->height = H;
        pp->width = W;
        if (P > pp->cap_screen && (rc = ensure(g, &pp->d_screen, &pp->cap_screen, P, true))) return rc;
    }
    if (prm->lowpass_before_sync || prm->autogain_after_proc) {
        if ((rc = ensure(g, &pp->d_tmp1, &pp->cap_tmp1, P * (size_t)F, true))) return rc;
        if ((rc = ensure(g, &pp->d_tmp2, &pp->cap_tmp2, P * (size_t)F, true))) return rc;
    }
    if (pp->lowpass_before_sync != prm->lowpass_before_sync) {  // dsp.c:178-186
        pp->lowpass_before_sync = prm->lowpass_before_sync;
        HIP_TRY(g, hipMemsetAsync(pp->d_screen, 0, P * sizeof(float), g->stream));
    }

    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
    const size_t nblk = (size_t)F * tiles_x * tiles_y;
    if ((rc = ensure(g, &pp->d_bmin, &pp->cap_bmin, nblk))) return rc;
    if ((rc = ensure(g, &pp->d_bmax, &pp->cap_bmax, nblk))) return rc;
    if ((rc = ensure(g, &pp->d_tflag, &pp->cap_tflag, nblk))) return rc;
    if ((rc = ensure(g, &pp->d_colp, &pp->cap_colp, (size_t)F * tiles_y * 3 * W))) return rc;
    if ((rc = ensure(g, &pp->d_rowp, &pp->cap_rowp, (size_t)F * tiles_x * 3 * H))) return rc;
    if ((rc = ensure(g, &pp->d_fmin, &pp->cap_fmin, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_fmax, &pp->cap_fmax, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_strip_x, &pp->cap_sx, (size_t)F * 3 * W))) return rc;
    if ((rc = ensure(g, &pp->d_strip_y, &pp->cap_sy, (size_t)F * 3 * H))) return rc;
    {
        const size_t nmax = (size_t)(W > H ? W : H);
        // [F][2][nmax] floats + [F][2][nmax+1] doubles + [F][2] doubles + [F][2] speculative search results
        const size_t floats = (((size_t)F * 2 * nmax + 1) & ~(size_t)1) + 2 * ((size_t)F * 2 * (nmax + 1) + (size_t)F * 2) +
                              (size_t)F * 2 * (sizeof(SpecEntry) / sizeof(float)) + 16;
        if ((rc = ensure(g, &pp->d_work, &pp->cap_work, floats))) return rc;
        if ((rc = ensure(g, &pp->d_sflag, &pp->cap_sflag, (size_t)F * 6 + 4))) return rc;  // flags, toss-up marks, fresh marks, redo
        if ((rc = ensure(g, &pp->d_exact, &pp->cap_exact, (size_t)F * 2 * nmax))) return rc;
    }
    pp->chain_has_autogain = 0;
    pp->last_F = F;
    if ((size_t)F > pp->cap_chain) {
        if (pp->d_chain) { (void)hipStreamSynchronize(g->stream); (void)hipFree(pp->d_chain); (void)hipHostFree(pp->h_chain); }
        pp->d_chain = nullptr; pp->h_chain = nullptr; pp->cap_chain = 0;
        if (hipMalloc(&pp->d_chain, sizeof(ChainOut) * F) != hipSuccess || hipHostMalloc(&pp->h_chain, sizeof(ChainOut) * F, hipHostMallocDefault) != hipSuccess)
            return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "chain buffers");
        pp->cap_chain = F;
    }
    return TSDRGPU_OK;
}

static void pp_convert_info(tsdrgpu_postproc_t *pp, int F, tsdrgpu_pp_frameinfo_t *h_info);
static int pp_copy_info(tsdrgpu_postproc_t *pp, int F, tsdrgpu_pp_frameinfo_t *h_info)
{
    tsdrgpu_t *g = pp->g;
    // the chain record is complete after the last k_chain; convert on the host after the sync
    HIP_TRY(g, hipMemcpyAsync(pp->h_chain, pp->d_chain, sizeof(ChainOut) * F, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    pp_convert_info(pp, F, h_info);
    return TSDRGPU_OK;
}

static void pp_convert_info(tsdrgpu_postproc_t *pp, int F, tsdrgpu_pp_frameinfo_t *h_info)
{
    for (int f = 0; f < F; f++) {
        const ChainOut &c = pp->h_chain[f];
        tsdrgpu_pp_frameinfo_t &o = h_info[f];
        o.lastmin = c.lastmin; o.lastmax = c.lastmax;
        o.dx = c.dx; o.vx = c.vx; o.stripx = c.stripx;
        o.dy = c.dy; o.vy = c.vy; o.stripy = c.stripy;
        o.locked = c.locked; o.pll_fired = c.pll_fired;
        o.avg_speed = c.avg_speed; o.frameratediff = c.frameratediff;
    }
}

extern "C" int tsdrgpu_postproc_run(tsdrgpu_postproc_t *pp, const float *d_frames, int F, int W, int H,
                                    const tsdrgpu_pp_params_t *prm, float *d_out, tsdrgpu_pp_frameinfo_t *h_info)
{
    if (!pp || !d_frames || !d_out || !prm || F < 0 || W <= 0 || H <= 0)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_run", "bad argument") : TSDRGPU_EINVAL;
    if (pp->pending) return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_run", "a split run is open: call tsdrgpu_postproc_finish first");
    if (F == 0) return TSDRGPU_OK;
    tsdrgpu_t *g = pp->g;
    const size_t P = (size_t)W * H;
    int rc;
    if ((rc = pp_prepare(pp, F, W, H, prm))) return rc;

    const float a = prm->motionblur;
    const int lbs = prm->lowpass_before_sync, aap = prm->autogain_after_proc;
    const int map = prm->autoshift ? PASS_ROLL : 0;
    const long long Ps = (long long)P;

    if (!lbs && !aap) {
        // order A (library default): autogain -> collapse -> sync -> IIR; result = screenbuffer
        const int lines = (!prm->autoshift && a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        if ((rc = launch_stats(pp, d_frames, Ps, F, W, H, 1))) return rc;
        if ((rc = launch_chain(pp, d_frames, Ps, F, W, H, 1, 1, 1, prm))) return rc;
        if ((rc = launch_pass(pp, PASS_NORMALISE | map | lines | PASS_IIR, d_frames, Ps, d_out, Ps, F, W, H, a))) return rc;
    } else if (!lbs && aap) {
        // order B: collapse(raw) -> sync -> IIR -> autogain(screen); result = sendbuffer
        const int lines = (!prm->autoshift && a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        if ((rc = launch_stats(pp, d_frames, Ps, F, W, H, 1))) return rc;
        if ((rc = launch_chain(pp, d_frames, Ps, F, W, H, 0, 1, 0, prm))) return rc;
        if ((rc = launch_pass(pp, map | lines | PASS_IIR, d_frames, Ps, pp->d_tmp1, Ps, F, W, H, a))) return rc;
        if ((rc = launch_stats(pp, pp->d_tmp1, Ps, F, W, H, 0))) return rc;
        if ((rc = launch_chain(pp, pp->d_tmp1, Ps, F, W, H, 1, 0, 0, prm))) return rc;
        if ((rc = launch_pass(pp, PASS_NORMALISE, pp->d_tmp1, Ps, d_out, Ps, F, W, H, a))) return rc;
    } else if (lbs && !aap) {
        // order C (GUI default): autogain -> IIR -> collapse(screen) -> sync; result = corrected / screen
        const int lines = (!prm->autoshift && !prm->superresolution) ? PASS_LINES : 0;
        if ((rc = launch_stats(pp, d_frames, Ps, F, W, H, 0))) return rc;
        if ((rc = launch_chain(pp, d_frames, Ps, F, W, H, 1, 0, 0, prm))) return rc;
        if ((rc = launch_pass(pp, PASS_NORMALISE | PASS_IIR, d_frames, Ps, pp->d_tmp1, Ps, F, W, H, a))) return rc;
        if ((rc = launch_stats(pp, pp->d_tmp1, Ps, F, W, H, 1))) return rc;
        if ((rc = launch_chain(pp, pp->d_tmp1, Ps, F, W, H, 0, 1, 0, prm))) return rc;
        if ((rc = launch_pass(pp, map | lines, pp->d_tmp1, Ps, d_out, Ps, F, W, H, a))) return rc;
    } else {
        // order D: IIR -> collapse(screen) -> sync -> autogain(sync result); result = sendbuffer
        const int lines = (!prm->autoshift && !prm->superresolution) ? PASS_LINES : 0;
        if ((rc = launch_pass(pp, PASS_IIR, d_frames, Ps, pp->d_tmp1, Ps, F, W, H, a))) return rc;
        if ((rc = launch_stats(pp, pp->d_tmp1, Ps, F, W, H, 1))) return rc;
        if ((rc = launch_chain(pp, pp->d_tmp1, Ps, F, W, H, 0, 1, 0, prm))) return rc;
        if ((rc = launch_pass(pp, map | lines, pp->d_tmp1, Ps, pp->d_tmp2, Ps, F, W, H, a))) return rc;
        if ((rc = launch_stats(pp, pp->d_tmp2, Ps, F, W, H, 0))) return rc;
        if ((rc = launch_chain(pp, pp->d_tmp2, Ps, F, W, H, 1, 0, 0, prm))) return rc;
        if ((rc = launch_pass(pp, PASS_NORMALISE, pp->d_tmp2, Ps, d_out, Ps, F, W, H, a))) return rc;
    }

    if (h_info) return pp_copy_info(pp, F, h_info);
    return TSDRGPU_OK;
}

// Split form of the default stage order.  begin(): statistics on the main stream, then the short,
// latency-bound chain kernels (autogain IIR, strip blur, sync search, PLL: ~0.1 ms of mostly idle GPU
// per batch) on the context's side stream; whatever the caller queues on the main stream before
// finish() — e.g. tsdrgpu_autocorr_run on the same samples — runs meanwhile.  finish(): the main
// stream waits for the chain, then the normalise/IIR pass.  Results are identical to
// tsdrgpu_postproc_run; for the other three stage orders begin() only records its arguments.
extern "C" int tsdrgpu_postproc_begin(tsdrgpu_postproc_t *pp, const float *d_frames, int F, int W, int H,
                                      const tsdrgpu_pp_params_t *prm)
{
    if (!pp || !d_frames || !prm || F < 0 || W <= 0 || H <= 0)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_begin", "bad argument") : TSDRGPU_EINVAL;
    if (pp->pending) return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_begin", "a split run is already open");
    tsdrgpu_t *g = pp->g;
    pp->p_frames = d_frames;
    pp->p_F = F; pp->p_W = W; pp->p_H = H;
    pp->p_prm = *prm;
    if (F == 0 || prm->lowpass_before_sync || prm->autogain_after_proc) {
        pp->pending = 2;
        return TSDRGPU_OK;
    }
    int rc;
    if ((rc = pp_prepare(pp, F, W, H, prm))) return rc;
    const long long Ps = (long long)W * H;
    if ((rc = launch_stats_tiles(pp, d_frames, Ps, F, W, H, 1))) return rc;
    HIP_TRY(g, hipEventRecord(pp->ev_stats, g->stream));
    HIP_TRY(g, hipStreamWaitEvent(g->stream2, pp->ev_stats, 0));
    if ((rc = launch_stats_reduce(pp, g->stream2, F, W, H, 1))) return rc;
    pp->chain_st = g->stream2;
    rc = launch_chain(pp, d_frames, Ps, F, W, H, 1, 1, 1, prm);
    pp->chain_st = nullptr;
    if (rc) return rc;
    HIP_TRY(g, hipEventRecord(pp->ev_chain, g->stream2));
    pp->pending = 1;
    return TSDRGPU_OK;
}

// Fused run: see k_frame_tile_pass.  Everything is queued here; finish() only joins the streams.
extern "C" int tsdrgpu_postproc_begin_minmax(tsdrgpu_postproc_t *pp, const float *d_frames, int F, int W, int H,
                                             const tsdrgpu_pp_params_t *prm, const float *d_fmin, const float *d_fmax,
                                             float *d_out)
{
    if (!pp || !d_frames || !prm || !d_fmin || !d_fmax || !d_out || F < 0 || W <= 0 || H <= 0)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_begin_minmax", "bad argument") : TSDRGPU_EINVAL;
    if (pp->pending) return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_begin_minmax", "a split run is already open");
    tsdrgpu_t *g = pp->g;
    if (F == 0 || prm->lowpass_before_sync || prm->autogain_after_proc || prm->autoshift) {
        // other stage orders take their statistics from processed frames, and the roll needs dx/dy before
        // the pass: the plain split run handles those (it ignores the supplied min/max)
        const int rc = tsdrgpu_postproc_begin(pp, d_frames, F, W, H, prm);
        return rc;
    }
    pp->p_frames = d_frames;
    pp->p_out = d_out;
    pp->p_F = F; pp->p_W = W; pp->p_H = H;
    pp->p_prm = *prm;
    int rc;
    if ((rc = pp_prepare(pp, F, W, H, prm))) return rc;
    const size_t P = (size_t)W * H;
    if ((rc = ensure(g, &pp->d_screen2, &pp->cap_screen2, pp->cap_screen > P ? pp->cap_screen : P, true))) return rc;
    if ((rc = ensure(g, &pp->d_dump, &pp->cap_dump, (size_t)1024))) return rc;
    const long long Ps = (long long)P;
    const float a = prm->motionblur;
    const int lines = (a == 0.0f && !prm->superresolution) ? 1 : 0;  // autoshift is off on this path
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TP_H - 1) / TP_H;  // this path's tiles are 16 rows high
    if ((rc = ensure(g, &pp->d_tflag, &pp->cap_tflag, (size_t)F * tiles_x * tiles_y))) return rc;
    if ((rc = ensure(g, &pp->d_colp, &pp->cap_colp, (size_t)F * tiles_y * 3 * W))) return rc;

    // autogain IIR from the supplied min/max, then normalise + IIR + partial sums in one trip
    static const int tiles_only = getenv("TSDRGPU_FUSE_TILES") ? 1 : 0;
    const bool overlap = (const float *)d_out < d_frames + (long long)F * Ps && d_frames < (const float *)d_out + (long long)F * Ps;
    const bool flat = a == 0.0f && F >= 8 && !overlap && !tiles_only;
    if (flat && !pp->d_odd && hipMalloc(&pp->d_odd, sizeof(int)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "flag");
    pp->ext_fmin = d_fmin;
    pp->ext_fmax = d_fmax;
    pp->clear_with_autogain = flat ? pp->d_odd : nullptr;  // the flat trip's redo flag is zeroed by the chain kernel in front of it
    rc = launch_chain(pp, d_frames, Ps, F, W, H, 1, 0, 1, prm);
    pp->clear_with_autogain = nullptr;
    pp->ext_fmin = pp->ext_fmax = nullptr;
    if (rc) return rc;
    // Motion blur 0: a frame's output does not depend on the previous one's, so the trip is a FLAT kernel over (tile, frame)
    // — k_frame_stats<true>: the statistics kernel that also stores the normalised pixels — instead of tiles that walk
    // the frames.  Same guard as launch_pass: the exceptions (non-finite values, -0.0) raise *d_odd and the literal
    // pass, queued by _finish behind the sync detector and gated on the flag, redoes the batch from the raw frames.
    if (flat) {
        const int ftiles_y = (H + TILE_H - 1) / TILE_H;
        StatsStore st;
        st.dst = d_out; st.dstride = Ps; st.chain = pp->d_chain; st.screen = pp->d_screen; st.odd = pp->d_odd;
        TSDR_LAUNCH(g, PROF_FRAME_PASS, g->stream, k_frame_stats<true>, (unsigned)(tiles_x * ftiles_y * F), 256, d_frames, Ps, W, H, tiles_x, ftiles_y,
                    pp->d_bmin, pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_tflag, 1, st);
        KERNEL_CHECK(g, "k_frame_stats<store>");
        TSDR_LAUNCH(g, PROF_FRAME_REDUCE, g->stream, k_frame_reduce, dim3(((W > H ? W : H) + 255) / 256, 3, F), 256, W, H, tiles_x, ftiles_y, pp->d_bmin,
                    pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_tflag, 1, 0, TILE_H);
        KERNEL_CHECK(g, "k_frame_reduce");
        HIP_TRY(g, hipEventRecord(pp->ev_stats, g->stream));
        HIP_TRY(g, hipStreamWaitEvent(g->stream2, pp->ev_stats, 0));
        pp->chain_st = g->stream2;
        pp->ext_fmin = d_fmin;
        pp->ext_fmax = d_fmax;
        rc = launch_chain(pp, d_frames, Ps, F, W, H, 0, 1, 1, prm);
        pp->ext_fmin = pp->ext_fmax = nullptr;
        pp->chain_st = nullptr;
        if (rc) return rc;
        if (lines) {  // the painted lines
            TSDR_LAUNCH(g, PROF_CHAIN, g->stream2, k_paint_lines, dim3(((W > H ? W : H) + 255) / 256, 2, F), 256, d_out, Ps, W, H, pp->d_chain);
            KERNEL_CHECK(g, "k_paint_lines");
        }
        // the new state: the last frame as it now stands, unless the batch is about to be redone
        TSDR_LAUNCH(g, PROF_CHAIN, g->stream2, k_pass_state, (unsigned)(g->prop.multiProcessorCount * 4), 256,
                    (const float *)(d_out + (long long)(F - 1) * Ps), pp->d_screen, (int)P, (const int *)pp->d_odd);
        KERNEL_CHECK(g, "k_pass_state");
        HIP_TRY(g, hipEventRecord(pp->ev_chain, g->stream2));
        pp->pending = 4;
        return TSDRGPU_OK;
    }
    TSDR_LAUNCH(g, PROF_FRAME_PASS, g->stream, k_frame_tile_pass, (unsigned)(tiles_x * tiles_y), 512, d_frames, Ps, d_out, Ps, F, W, H, tiles_x,
                tiles_y, pp->d_chain, pp->d_screen, pp->d_screen2, a, pp->d_colp, pp->d_rowp, pp->d_tflag, pp->d_dump);
    KERNEL_CHECK(g, "k_frame_tile_pass");
    {   // the state the next run reads is the buffer just written
        float *t = pp->d_screen; pp->d_screen = pp->d_screen2; pp->d_screen2 = t;
        const size_t c = pp->cap_screen; pp->cap_screen = pp->cap_screen2; pp->cap_screen2 = c;
        // the reference keeps whatever lies beyond width*height in its single buffer (it matters after a later
        // resolution change, dsp.c:152-173): carry that tail over to the buffer that is now current
        const size_t common = pp->cap_screen < pp->cap_screen2 ? pp->cap_screen : pp->cap_screen2;
        if (common > P)
            HIP_TRY(g, hipMemcpyAsync(pp->d_screen + P, pp->d_screen2 + P, (common - P) * sizeof(float), hipMemcpyDeviceToDevice, g->stream));
    }
    // strips in line (bandwidth work: it would only fight the caller's kernels for HBM on the side stream) ...
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, g->stream, k_frame_reduce, dim3(((W > H ? W : H) + 255) / 256, 3, F), 256, W, H, tiles_x, tiles_y, pp->d_bmin,
                pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_tflag, 1, 0, TP_H);
    KERNEL_CHECK(g, "k_frame_reduce");
    HIP_TRY(g, hipEventRecord(pp->ev_stats, g->stream));
    // ... the latency-bound sync detector (+ the painted lines) on the side stream
    HIP_TRY(g, hipStreamWaitEvent(g->stream2, pp->ev_stats, 0));
    pp->chain_st = g->stream2;
    pp->ext_fmin = d_fmin;
    pp->ext_fmax = d_fmax;
    rc = launch_chain(pp, d_frames, Ps, F, W, H, 0, 1, 1, prm);
    pp->ext_fmin = pp->ext_fmax = nullptr;
    pp->chain_st = nullptr;
    if (rc) return rc;
    if (lines) {
        // d_screen2 now holds the pre-batch state, d_screen the post-batch one
        TSDR_LAUNCH(g, PROF_CHAIN, g->stream2, k_fix_lines, dim3(((W > H ? W : H) + 255) / 256, 2, F), 256, d_frames, Ps, d_out, Ps, F, W, H,
                    pp->d_chain, pp->d_screen2, pp->d_screen, a);
        KERNEL_CHECK(g, "k_fix_lines");
    }
    HIP_TRY(g, hipEventRecord(pp->ev_chain, g->stream2));
    pp->pending = 3;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_finish(tsdrgpu_postproc_t *pp, float *d_out, tsdrgpu_pp_frameinfo_t *h_info)
{
    if (!pp || !d_out) return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_finish", "bad argument") : TSDRGPU_EINVAL;
    if (!pp->pending) return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_finish", "no split run is open");
    if (pp->pending >= PEND_BAND)
        return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_finish", "a band run is open: tsdrgpu_postproc_band_finish / _band_advance / _band_step close it");
    tsdrgpu_t *g = pp->g;
    const int mode = pp->pending;
    if ((mode == 3 || mode == 4) && d_out != pp->p_out)  // the run stays open: the caller can still finish it properly
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_finish", "the fused run's frames are in the buffer given to _begin_minmax: name the same one");
    pp->pending = 0;
    const int F = pp->p_F, W = pp->p_W, H = pp->p_H;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    if (mode == 2) return tsdrgpu_postproc_run(pp, pp->p_frames, F, W, H, prm, d_out, h_info);
    HIP_TRY(g, hipStreamWaitEvent(g->stream, pp->ev_chain, 0));
    if (mode == 3) {  // fused run: the frames are already in the d_out given to begin_minmax
        if (h_info) return pp_copy_info(pp, F, h_info);
        return TSDRGPU_OK;
    }
    if (mode == 4) {  // flat fused run: ... unless a frame held one of the exceptions; then the literal pass redoes the batch
        const float a0 = prm->motionblur;
        const int lines0 = (a0 == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        const long long Pl = (long long)W * H;
        int rc0;
        if ((rc0 = launch_pass_literal(pp, PASS_NORMALISE | lines0 | PASS_IIR, pp->p_frames, Pl, d_out, Pl, F, W, H, a0, pp->d_odd))) return rc0;
        if (h_info) return pp_copy_info(pp, F, h_info);
        return TSDRGPU_OK;
    }
    const float a = prm->motionblur;
    const long long Ps = (long long)W * H;
    const int map = prm->autoshift ? PASS_ROLL : 0;
    const int lines = (!prm->autoshift && a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
    int rc;
    if ((rc = launch_pass(pp, PASS_NORMALISE | map | lines | PASS_IIR, pp->p_frames, Ps, d_out, Ps, F, W, H, a))) return rc;
    if (h_info) return pp_copy_info(pp, F, h_info);
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// Row-band sharding of the frame path (dsp.c:41-110, syncdetector.c:171-225).  Temporal sharding is
// impossible (IIR state, autogain, sync state are frame-to-frame recurrences) but rows shard: rank r holds rows
// [y0, y0+rows) of every frame, its IIR state and its part of the output.  What a frame needs from the other ranks is
// tiny: min / max / pixel 0 (autogain, dsp.c:50-66) and the two collapsed strips (dsp.c:96-110) — column sums add up
// over the bands, row sums are concatenated (added here with zeros elsewhere, so one sum all-reduce does both).
// begin: band statistics -> exchange buffers; the CALLER all-reduces them in place (sum for d_xsum, max for d_xmax:
// tsdrgpu_comm_allreduce_f64 / _f32max over RCCL); finish: every rank runs the (tiny, replicated) chain on the
// identical strips and the normalise / lines / IIR pass on its rows.  With bands that start on multiples of 32 rows
// the tile partial sums are the single-GPU run's, f64 sums of <= a few hundred f32 partials are exact, so strips, sync
// decisions and therefore frames are bit-identical to the single-GPU run in its fast mode (no literal re-collapse of
// toss-up strips: that walks a column through every band in order).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_band_pack(int F, int W, int Htot, int y0, int rows, const double *__restrict__ strip_x,
                                                   const double *__restrict__ strip_y, const float *__restrict__ fmin_,
                                                   const float *__restrict__ fmax_, const float *__restrict__ frames, long long fstride,
                                                   double *__restrict__ xsum, float *__restrict__ xmax)
{
    const int f = blockIdx.z, q = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W + Htot) {
        double v;
        if (i < W) v = strip_x[((long long)f * 3 + q) * W + i];
        else {
            const int y = i - W;
            v = (y >= y0 && y < y0 + rows) ? strip_y[((long long)f * 3 + q) * rows + (y - y0)] : 0.0;
        }
        xsum[((long long)f * 3 + q) * (W + Htot) + i] = v;
    }
    if (i == 0 && q == 0 && xmax) {  // (no xmax: the fused band run exchanged the range before its trip)
        xmax[f * 4 + 0] = -fmin_[f];
        xmax[f * 4 + 1] = fmax_[f];
        const float p0_ = (y0 == 0) ? frames[(long long)f * fstride] : -INFINITY;  // dsp.c:50-51: v[0] seeds min and max
        // a max all-reduce drops a NaN (fmaxf semantics), and a NaN at pixel 0 is what poisons the reference's autogain for good
        // (dsp.c:50-59: no comparison ever replaces it): it travels as a flag in the fourth slot
        xmax[f * 4 + 2] = (p0_ != p0_) ? -INFINITY : p0_;
        xmax[f * 4 + 3] = (p0_ != p0_) ? 1.0f : -INFINITY;
    }
}

__global__ __launch_bounds__(256) void k_band_unpack(int F, int W, int Htot, const double *__restrict__ xsum, const float *__restrict__ xmax,
                                                     double *__restrict__ strip_x, double *__restrict__ strip_y, float *__restrict__ fmin_,
                                                     float *__restrict__ fmax_, float *__restrict__ v0)
{
    const int f = blockIdx.z, q = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W + Htot) {
        const double v = xsum[((long long)f * 3 + q) * (W + Htot) + i];
        if (i < W) strip_x[((long long)f * 3 + q) * W + i] = v;
        else strip_y[((long long)f * 3 + q) * Htot + (i - W)] = v;
    }
    if (i == 0 && q == 0) {
        fmin_[f] = -xmax[f * 4 + 0];
        fmax_[f] = xmax[f * 4 + 1];
        v0[f] = (xmax[f * 4 + 3] > 0.0f) ? __int_as_float(0x7fc00000) : xmax[f * 4 + 2];
    }
}

__global__ void k_band_chain(const ChainOut *__restrict__ chain, ChainOut *__restrict__ out, int F, int y0)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    ChainOut c = chain[f];
    c.dy -= y0;  // the pass compares band-local row numbers
    out[f] = c;
}

extern "C" int tsdrgpu_postproc_band_begin(tsdrgpu_postproc_t *pp, const float *d_band, int F, int W, int Htot, int y0, int rows,
                                           const tsdrgpu_pp_params_t *prm, double **d_xsum, int64_t *n_xsum, float **d_xmax, int64_t *n_xmax)
{
    if (!pp || !d_band || !prm || F <= 0 || W <= 0 || Htot <= 0 || y0 < 0 || rows <= 0 || y0 + rows > Htot)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_begin", "a split run is already open");
    if (prm->lowpass_before_sync || prm->autogain_after_proc || prm->autoshift || prm->pll)
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin",
                         "row bands support the library-default stage order without autoshift (the 2-D roll needs every row) and without the PLL");
    if (y0 % TILE_H) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin", "a band must start on a multiple of 32 rows");
    int rc;
    if ((rc = pp_prepare(pp, F, W, Htot, prm))) return rc;
    if ((rc = ensure(g, &pp->d_xsum, &pp->cap_xsum, (size_t)F * 3 * (W + Htot)))) return rc;
    if ((rc = ensure(g, &pp->d_xmax, &pp->cap_xmax, (size_t)F * 4))) return rc;
    if ((rc = ensure(g, &pp->d_v0, &pp->cap_v0, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_chain_band, &pp->cap_chain_band, (size_t)F))) return rc;
    const long long Pb = (long long)W * rows;
    if ((rc = launch_stats(pp, d_band, Pb, F, W, rows, 1))) return rc;  // strips of the band: [F][3][W] and [F][3][rows]
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, g->stream, k_band_pack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, y0, rows, pp->d_strip_x, pp->d_strip_y,
                pp->d_fmin, pp->d_fmax, d_band, Pb, pp->d_xsum, pp->d_xmax);
    KERNEL_CHECK(g, "k_band_pack");
    pp->p_frames = d_band;
    pp->p_F = F; pp->p_W = W; pp->p_H = Htot;
    pp->p_prm = *prm;
    pp->band_y0 = y0;
    pp->band_rows = rows;
    pp->band_stage = 0;
    pp->band_spec = 0;
    pp->band_fused = 0;
    pp->brelay_src = nullptr;
    pp->pending = PEND_BAND;
    if (d_xsum) *d_xsum = pp->d_xsum;
    if (n_xsum) *n_xsum = (int64_t)F * 3 * (W + Htot);
    if (d_xmax) *d_xmax = pp->d_xmax;
    if (n_xmax) *n_xmax = (int64_t)F * 4;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_band_finish(tsdrgpu_postproc_t *pp, float *d_out_band, tsdrgpu_pp_frameinfo_t *h_info)
{
    if (!pp || !d_out_band) return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_finish", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending != PEND_BAND) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_finish", "no band run is open");
    if (pp->band_fused) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_finish", "a fused band run is open: tsdrgpu_postproc_band_advance closes it");
    pp->pending = 0;
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H, y0 = pp->band_y0, rows = pp->band_rows;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, g->stream, k_band_unpack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_xmax, pp->d_strip_x,
                pp->d_strip_y, pp->d_fmin, pp->d_fmax, pp->d_v0);
    KERNEL_CHECK(g, "k_band_unpack");
    pp->band_mode = 1;
    int rc = launch_chain(pp, pp->d_v0, 1, F, W, Htot, 1, 1, 1, prm);  // "frames" = pixel 0 of every frame, stride 1
    pp->band_mode = 0;
    if (rc) return rc;
    TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_band_chain, (F + 63) / 64, 64, pp->d_chain, pp->d_chain_band, F, y0);
    KERNEL_CHECK(g, "k_band_chain");
    const float a = prm->motionblur;
    const int lines = (a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
    const long long Pb = (long long)W * rows;
    ChainOut *full = pp->d_chain;
    pp->d_chain = pp->d_chain_band;  // what the pass reads
    rc = launch_pass(pp, PASS_NORMALISE | lines | PASS_IIR, pp->p_frames, Pb, d_out_band, Pb, F, W, rows, a);
    pp->d_chain = full;
    if (rc) return rc;
    if (h_info) return pp_copy_info(pp, F, h_info);
    return TSDRGPU_OK;
}

// (the exchange kernels of the min/max: shared by the fused band run below and the general band runs further down)
__global__ __launch_bounds__(256) void k_band_pack_mm(int F, int y0, const float *__restrict__ fmin_, const float *__restrict__ fmax_,
                                                      const float *__restrict__ frames, long long fstride, float *__restrict__ xmax)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    xmax[f * 4 + 0] = -fmin_[f];
    xmax[f * 4 + 1] = fmax_[f];
    const float p0_ = (y0 == 0) ? frames[(long long)f * fstride] : -INFINITY;  // dsp.c:50-51: v[0] seeds min and max
    xmax[f * 4 + 2] = (p0_ != p0_) ? -INFINITY : p0_;  // (a NaN travels as a flag in the fourth slot: see k_band_pack)
    xmax[f * 4 + 3] = (p0_ != p0_) ? 1.0f : -INFINITY;
}

__global__ __launch_bounds__(256) void k_band_unpack_mm(int F, const float *__restrict__ xmax, float *__restrict__ fmin_, float *__restrict__ fmax_,
                                                        float *__restrict__ v0)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    fmin_[f] = -xmax[f * 4 + 0];
    fmax_[f] = xmax[f * 4 + 1];
    v0[f] = (xmax[f * 4 + 3] > 0.0f) ? __int_as_float(0x7fc00000) : xmax[f * 4 + 2];
}

__global__ __launch_bounds__(256) void k_band_unpack_sum(int F, int W, int Htot, const double *__restrict__ xsum, double *__restrict__ strip_x,
                                                         double *__restrict__ strip_y)
{
    const int f = blockIdx.z, q = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W + Htot) return;
    const double v = xsum[((long long)f * 3 + q) * (W + Htot) + i];
    if (i < W) strip_x[((long long)f * 3 + q) * W + i] = v;
    else strip_y[((long long)f * 3 + q) * Htot + (i - W)] = v;
}

// ---------------------------------------------------------------------------
// The FUSED band run (the band form of tsdrgpu_postproc_begin_minmax): the resampler's band form leaves every frame's min/max over
// this band's pixels (frame tracking in tsdrgpu_resample_band), so the range can be exchanged BEFORE the band is read and ONE trip
// over the raw band both writes the normalised / IIR'd rows and gathers the strip partials — 12P bytes per band pixel instead of
// the 16P of statistics + pass (measured on one rank, configs[4]: the whole gap between the band path and the single-GPU run,
// profiles/round6_ab_runs.txt).  Order of calls, every rank alike:
//   tsdrgpu_postproc_band_begin_minmax   {-min, max, pixel 0} of every frame -> d_xmax       caller: max all-reduce
//   tsdrgpu_postproc_band_fused          autogain recurrence, the trip, strip partials -> d_xsum   caller: sum all-reduce
//   tsdrgpu_postproc_band_advance        the (replicated, contract-exact) sync chain with its relays; then only the painted lines
// The trip is k_frame_tile_pass on the band's rows (tiles of 16 rows: a band starts on a multiple of 32, so its tiles are the
// single-GPU fused run's tiles and its partial sums that run's), lines by k_fix_lines from the state the batch started with.
// Library-default stage order, no autoshift, no PLL — like tsdrgpu_postproc_band_begin.  Reference: dsp.c:41-110,134-239.
// ---------------------------------------------------------------------------
extern "C" int tsdrgpu_postproc_band_begin_minmax(tsdrgpu_postproc_t *pp, const float *d_band, int F, int W, int Htot, int y0, int rows,
                                                  const tsdrgpu_pp_params_t *prm, const float *d_fmin, const float *d_fmax, float **d_xmax,
                                                  int64_t *n_xmax)
{
    if (!pp || !d_band || !prm || !d_fmin || !d_fmax || F <= 0 || W <= 0 || Htot <= 0 || y0 < 0 || rows <= 0 || y0 + rows > Htot)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin_minmax", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_begin_minmax", "a split run is already open");
    if (prm->lowpass_before_sync || prm->autogain_after_proc || prm->autoshift || prm->pll)
        return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin_minmax",
                         "row bands support the library-default stage order without autoshift (the 2-D roll needs every row) and without the PLL");
    if (y0 % TILE_H) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_begin_minmax", "a band must start on a multiple of 32 rows");
    int rc;
    if ((rc = pp_prepare(pp, F, W, Htot, prm))) return rc;
    const size_t Pb = (size_t)W * rows;
    if ((rc = ensure(g, &pp->d_xsum, &pp->cap_xsum, (size_t)F * 3 * (W + Htot)))) return rc;
    if ((rc = ensure(g, &pp->d_xmax, &pp->cap_xmax, (size_t)F * 4))) return rc;
    if ((rc = ensure(g, &pp->d_v0, &pp->cap_v0, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_chain_band, &pp->cap_chain_band, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_screen2, &pp->cap_screen2, pp->cap_screen > Pb ? pp->cap_screen : Pb, true))) return rc;
    if ((rc = ensure(g, &pp->d_dump, &pp->cap_dump, (size_t)1024))) return rc;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (rows + TP_H - 1) / TP_H;  // the trip's tiles are 16 rows high
    if ((rc = ensure(g, &pp->d_tflag, &pp->cap_tflag, (size_t)F * tiles_x * tiles_y))) return rc;
    if ((rc = ensure(g, &pp->d_colp, &pp->cap_colp, (size_t)F * tiles_y * 3 * W))) return rc;
    // the band's share of the range (the caller's arrays: tsdrgpu_resampler_frame_minmax) + pixel 0 from the rank that holds it
    HIP_TRY(g, hipMemcpyAsync(pp->d_fmin, d_fmin, sizeof(float) * (size_t)F, hipMemcpyDeviceToDevice, g->stream));
    HIP_TRY(g, hipMemcpyAsync(pp->d_fmax, d_fmax, sizeof(float) * (size_t)F, hipMemcpyDeviceToDevice, g->stream));
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, g->stream, k_band_pack_mm, (F + 255) / 256, 256, F, y0, pp->d_fmin, pp->d_fmax, d_band, (long long)Pb, pp->d_xmax);
    KERNEL_CHECK(g, "k_band_pack_mm");
    pp->p_frames = d_band;
    pp->p_F = F; pp->p_W = W; pp->p_H = Htot;
    pp->p_prm = *prm;
    pp->band_y0 = y0;
    pp->band_rows = rows;
    pp->band_stage = 0;
    pp->band_spec = 0;
    pp->band_fused = 1;
    pp->brelay_src = nullptr;
    pp->pending = PEND_BAND;
    if (d_xmax) *d_xmax = pp->d_xmax;
    if (n_xmax) *n_xmax = (int64_t)F * 4;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_band_fused(tsdrgpu_postproc_t *pp, float *d_out_band, double **d_xsum, int64_t *n_xsum)
{
    if (!pp || !d_out_band) return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_fused", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending != PEND_BAND || pp->band_fused != 1)
        return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_fused", "no fused band run is waiting for its trip (tsdrgpu_postproc_band_begin_minmax first)");
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H, y0 = pp->band_y0, rows = pp->band_rows;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    const long long Pb = (long long)W * rows;
    const float *d_band = pp->p_frames;
    const bool overlap = (const float *)d_out_band < d_band + (long long)F * Pb && d_band < (const float *)d_out_band + (long long)F * Pb;
    if (overlap) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_fused", "the output overlaps the band (the relays and the lines read the raw band afterwards)");
    hipStream_t st = g->stream;
    // the exchanged range -> the autogain recurrence (replicated: every rank the same values)
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack_mm, (F + 255) / 256, 256, F, pp->d_xmax, pp->d_fmin, pp->d_fmax, pp->d_v0);
    KERNEL_CHECK(g, "k_band_unpack_mm");
    // Motion blur 0 and a batch long enough: a frame's output does not depend on the previous one's, so the trip is the FLAT kernel over
    // (tile, frame) — k_frame_stats<true>, like tsdrgpu_postproc_begin_minmax's — instead of tiles that walk the frames (measured on the
    // headline stream in one band: 1.27 ms per pass for the two-trip run, 1.42 with the walking tiles).  Its exceptions (a -0.0 or
    // non-finite pixel, a non-finite incoming state) raise d_odd, and the literal pass, queued behind the chain and gated on the flag,
    // then redoes this band's rows from the raw band — per-pixel recurrences: a band decides for itself.
    static const int tiles_only = getenv("TSDRGPU_FUSE_TILES") ? 1 : 0;
    const bool flat = prm->motionblur == 0.0f && F >= 8 && !tiles_only;
    if (flat && !pp->d_odd && hipMalloc(&pp->d_odd, sizeof(int)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "flag");
    TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, (const float *)pp->d_v0, 1LL, pp->d_fmin, pp->d_fmax, pp->d_state, pp->d_chain, 1,
                prm->lowpasscoeff, flat ? pp->d_odd : (int *)nullptr);  // (the flat trip's redo flag is zeroed by the chain kernel in front of it)
    KERNEL_CHECK(g, "k_autogain_chain");
    pp->band_flat = flat ? 1 : 0;
    if (flat) {
        const int tiles_x = (W + TILE_W - 1) / TILE_W, ftiles_y = (rows + TILE_H - 1) / TILE_H;
        StatsStore ss;
        ss.dst = d_out_band; ss.dstride = Pb; ss.chain = pp->d_chain; ss.screen = pp->d_screen; ss.odd = pp->d_odd;
        TSDR_LAUNCH(g, PROF_FRAME_PASS, st, k_frame_stats<true>, (unsigned)(tiles_x * ftiles_y * F), 256, d_band, Pb, W, rows, tiles_x, ftiles_y, pp->d_bmin,
                    pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_tflag, 1, ss);
        KERNEL_CHECK(g, "k_frame_stats<store>");
        TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_frame_reduce, dim3(((W > rows ? W : rows) + 255) / 256, 3, F), 256, W, rows, tiles_x, ftiles_y, pp->d_bmin,
                    pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_tflag, 1, 0, TILE_H);
        KERNEL_CHECK(g, "k_frame_reduce");
        TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_pack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, y0, rows, pp->d_strip_x, pp->d_strip_y,
                    pp->d_fmin, pp->d_fmax, d_band, Pb, pp->d_xsum, (float *)nullptr);
        KERNEL_CHECK(g, "k_band_pack");
        pp->p_out = d_out_band;
        pp->band_fused = 2;
        if (d_xsum) *d_xsum = pp->d_xsum;
        if (n_xsum) *n_xsum = (int64_t)F * 3 * (W + Htot);
        return TSDRGPU_OK;
    }
    // the trip: normalise + IIR into d_out_band, strip partials of the band's rows
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (rows + TP_H - 1) / TP_H;
    TSDR_LAUNCH(g, PROF_FRAME_PASS, st, k_frame_tile_pass, (unsigned)(tiles_x * tiles_y), 512, d_band, Pb, d_out_band, Pb, F, W, rows, tiles_x, tiles_y,
                pp->d_chain, pp->d_screen, pp->d_screen2, prm->motionblur, pp->d_colp, pp->d_rowp, pp->d_tflag, pp->d_dump);
    KERNEL_CHECK(g, "k_frame_tile_pass");
    {   // the state the next run reads is the buffer just written (d_screen2 keeps the state the batch started with: the lines need it)
        float *t = pp->d_screen; pp->d_screen = pp->d_screen2; pp->d_screen2 = t;
        const size_t c = pp->cap_screen; pp->cap_screen = pp->cap_screen2; pp->cap_screen2 = c;
        const size_t common = pp->cap_screen < pp->cap_screen2 ? pp->cap_screen : pp->cap_screen2;
        if (common > (size_t)Pb)
            HIP_TRY(g, hipMemcpyAsync(pp->d_screen + Pb, pp->d_screen2 + Pb, (common - (size_t)Pb) * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_frame_reduce, dim3(((W > rows ? W : rows) + 255) / 256, 3, F), 256, W, rows, tiles_x, tiles_y, pp->d_bmin,
                pp->d_bmax, pp->d_colp, pp->d_rowp, pp->d_fmin, pp->d_fmax, pp->d_strip_x, pp->d_strip_y, pp->d_tflag, 1, 0, TP_H);
    KERNEL_CHECK(g, "k_frame_reduce");
    TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_pack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, y0, rows, pp->d_strip_x, pp->d_strip_y,
                pp->d_fmin, pp->d_fmax, d_band, Pb, pp->d_xsum, (float *)nullptr);
    KERNEL_CHECK(g, "k_band_pack");
    pp->p_out = d_out_band;
    pp->band_fused = 2;
    if (d_xsum) *d_xsum = pp->d_xsum;
    if (n_xsum) *n_xsum = (int64_t)F * 3 * (W + Htot);
    return TSDRGPU_OK;
}

// what is left of a fused band run once the chain has decided: the painted lines of this band's rows (syncdetector.c:209-223)
static int band_fused_lines(tsdrgpu_postproc_t *pp, float *d_out_band)
{
    tsdrgpu_t *g = pp->g;
    const int F = pp->p_F, W = pp->p_W, y0 = pp->band_y0, rows = pp->band_rows;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    if (d_out_band != pp->p_out) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_advance", "not the buffer the fused trip wrote");
    TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_band_chain, (F + 63) / 64, 64, pp->d_chain, pp->d_chain_band, F, y0);
    KERNEL_CHECK(g, "k_band_chain");
    const float a = prm->motionblur;
    if (pp->band_flat) {  // the flat trip: the lines are constants, the new state is the last frame — unless the batch is redone literally
        const long long Pb = (long long)W * rows;
        const int lines = (a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        if (lines) {
            TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_paint_lines, dim3(((W > rows ? W : rows) + 255) / 256, 2, F), 256, d_out_band, Pb, W, rows, pp->d_chain_band);
            KERNEL_CHECK(g, "k_paint_lines");
        }
        TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_pass_state, (unsigned)(g->prop.multiProcessorCount * 4), 256,
                    (const float *)(d_out_band + (long long)(F - 1) * Pb), pp->d_screen, (int)Pb, (const int *)pp->d_odd);
        KERNEL_CHECK(g, "k_pass_state");
        ChainOut *full = pp->d_chain;
        pp->d_chain = pp->d_chain_band;  // what the pass reads
        const int rc = launch_pass_literal(pp, PASS_NORMALISE | lines | PASS_IIR, pp->p_frames, Pb, d_out_band, Pb, F, W, rows, a, pp->d_odd);
        pp->d_chain = full;
        pp->band_flat = 0;
        return rc;
    }
    if (a == 0.0f && !prm->superresolution) {
        const long long Pb = (long long)W * rows;
        // d_screen2 holds the state the batch started with, d_screen the one the trip left
        TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_fix_lines, dim3(((W > rows ? W : rows) + 255) / 256, 2, F), 256, pp->p_frames, Pb, d_out_band, Pb, F, W, rows,
                    pp->d_chain_band, pp->d_screen2, pp->d_screen, a);
        KERNEL_CHECK(g, "k_fix_lines");
    }
    return TSDRGPU_OK;
}

// ---------------------------------------------------------------------------
// Contract-exact row bands.  The single-GPU run redoes the collapse of a strip literally — f32 additions in raster
// order, dsp.c:96-110 — where the strip holds exact ties (k_strip_flag) and where a sync decision was a toss-up at the
// precision of the f64 tile sums (k_sync_chain's margin test).  A column sum in that order runs through every band
// from the top row to the bottom one, so the bands take turns: at step s the band with index s continues every
// requested sum over its own rows from the values it received (a row sum lies in one band: that band forms it from
// zero) while all other ranks contribute zeros, and one sum all-reduce hands the result on.  After as many steps as
// there are bands every rank holds the reference's own strips for the requested (frame, axis) items and the replicated
// chain goes on exactly like the single-GPU one (run 0, then run 1 for the toss-ups).  Which items are requested is
// decided from the exchanged strips, identically on every rank, so the ranks agree on every collective without talking.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_band_relay(const int *__restrict__ items, int nmax, const float *__restrict__ band, long long fstride,
                                                    int W, int rows, int y0, const ChainOut *__restrict__ chain, int strips_normalised,
                                                    double *__restrict__ X)
{
    const int item = items[blockIdx.y];
    const int axis = item & 1, f = item >> 1;
    const float *src = band + (long long)f * fstride;
    const float lastmin = chain[f].lastmin, span = chain[f].span;
    double *x = X + (long long)blockIdx.y * nmax;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // The additions of one sum form a dependent chain in the reference's order; the loads do not, so they are issued a
    // block ahead (the first version, one load per addition, spent ~2 ms per relay step at 2962x2250 waiting for them).
#define RELAY_NORM(val_) (strips_normalised ? (((val_) > 250.0f || (val_) < -250.0f) ? (val_) : (((val_) - lastmin) / span)) : (val_)) /* dsp.c:80-86 */
    if (axis == 0) {  // column i: continue down this band's rows
        if (i >= W) return;
        float acc = (float)x[i];
        int y = 0;
        for (; y + 8 <= rows; y += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(long long)(y + u) * W + i];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += RELAY_NORM(v[u]);
        }
        for (; y < rows; y++) {
            const float val = src[(long long)y * W + i];
            acc += RELAY_NORM(val);
        }
        x[i] = (double)acc;
    } else {  // row y0 + i: all of it lies in this band
        if (i >= rows) return;
        float acc = 0.f;
        const float *row = src + (long long)i * W;
        int c = 0;
        for (; c + 16 <= W; c += 16) {
            float4_a4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const float4_a4 *>(row + c + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                acc += RELAY_NORM(q[u][0]);
                acc += RELAY_NORM(q[u][1]);
                acc += RELAY_NORM(q[u][2]);
                acc += RELAY_NORM(q[u][3]);
            }
        }
        for (; c < W; c++) {
            const float val = row[c];
            acc += RELAY_NORM(val);
        }
        x[y0 + i] = (double)acc;
    }
#undef RELAY_NORM
}

__global__ __launch_bounds__(256) void k_band_relay_take(const int *__restrict__ items, int nmax, int W, int H, const double *__restrict__ X,
                                                         float *__restrict__ exact)
{
    const int item = items[blockIdx.y];
    const int n = (item & 1) ? H : W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) exact[(long long)item * nmax + i] = (float)X[(long long)blockIdx.y * nmax + i];
}

// host copy of a [2F] device flag array -> the list of set entries on the device (rare path: synchronises)
static int band_collect_items(tsdrgpu_postproc_t *pp, const int *d_flags, int count, int *nitems)
{
    tsdrgpu_t *g = pp->g;
    int rc;
    if (pp->cap_hflags < (size_t)count) {
        free(pp->h_flags);
        pp->h_flags = (int *)malloc(sizeof(int) * (size_t)count * 2);
        pp->cap_hflags = pp->h_flags ? (size_t)count : 0;
        if (!pp->h_flags) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "band flags");
    }
    if ((rc = ensure(g, &pp->d_items, &pp->cap_items, (size_t)count))) return rc;
    HIP_TRY(g, hipMemcpyAsync(pp->h_flags, d_flags, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    int *list = pp->h_flags + count;
    int n = 0;
    for (int i = 0; i < count; i++)
        if (pp->h_flags[i]) list[n++] = i;
    if (n) {
        HIP_TRY(g, hipMemcpyAsync(pp->d_items, list, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, g->stream));
        HIP_TRY(g, hipStreamSynchronize(g->stream));  // `list` is reused
    }
    *nitems = n;
    return TSDRGPU_OK;
}

// one relay step: this band's turn -> continue the sums, otherwise contribute zeros
static int band_relay_step(tsdrgpu_postproc_t *pp, int band_index)
{
    tsdrgpu_t *g = pp->g;
    const int W = pp->p_W, H = pp->p_H, nmax = W > H ? W : H;
    const size_t bytes = sizeof(double) * (size_t)pp->relay_items * nmax;
    // step 0 starts every sum at zero; a rank whose turn it is not adds nothing to the all-reduce; the rank whose turn
    // it is continues from what the previous step's all-reduce left in the buffer
    if (pp->relay_step != band_index || pp->relay_step == 0) HIP_TRY(g, hipMemsetAsync(pp->d_relay, 0, bytes, g->stream));
    if (pp->relay_step == band_index) {
        const int longest = W > pp->band_rows ? W : pp->band_rows;
        TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_band_relay, dim3((longest + 255) / 256, pp->relay_items), 256, pp->d_items, nmax,
                    pp->brelay_src ? pp->brelay_src : pp->p_frames, (long long)W * pp->band_rows, W, pp->band_rows, pp->band_y0, pp->d_chain,
                    pp->brelay_src ? pp->bsync_norm : 1, pp->d_relay);
        KERNEL_CHECK(g, "k_band_relay");
    }
    return TSDRGPU_OK;
}

// the pass over this band's rows, which ends a band run
static int band_run_pass(tsdrgpu_postproc_t *pp, float *d_out_band)
{
    tsdrgpu_t *g = pp->g;
    const int F = pp->p_F, W = pp->p_W, y0 = pp->band_y0, rows = pp->band_rows;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_band_chain, (F + 63) / 64, 64, pp->d_chain, pp->d_chain_band, F, y0);
    KERNEL_CHECK(g, "k_band_chain");
    const float a = prm->motionblur;
    const int lines = (a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
    const long long Pb = (long long)W * rows;
    ChainOut *full = pp->d_chain;
    pp->d_chain = pp->d_chain_band;  // what the pass reads
    const int rc = launch_pass(pp, PASS_NORMALISE | lines | PASS_IIR, pp->p_frames, Pb, d_out_band, Pb, F, W, rows, a);
    pp->d_chain = full;
    return rc;
}

// The common batch holds no strip with exact ties and no toss-up decision, so neither relay is needed — but finding that out
// took the host two round trips per batch (the flag array behind k_strip_flag, then the one behind run 0 of the chain), each one
// a drained queue.  Speculated form: both halves of the chain — exchanged statistics -> autogain recurrence -> tie flags -> run 0
// of the sync chain -> toss-up flags — are queued as if no strip held ties, both flag arrays are copied out behind them and the
// host waits ONCE.  Every rank sees the same flags (the strips are replicated), so every rank goes the same way:
//   no flag at all      -> the pass (no relay, no collective, one round trip instead of two)
//   toss-ups only       -> what was queued IS the literal run up to its second question: on with the relay of round 1
//   strips with ties    -> run 0 saw strips that the relay had not made exact: the autogain / sync state saved before the first
//                          launch is put back and the literal run takes the batch from the start (its first question is answered)
// (A first form also queued the PASS ahead of the answer and replayed the whole batch when a flag fired: on configs[4]'s synthetic
// stream one batch in nine holds a toss-up, and a wasted pass costs more than a round trip saves — 1.42 against 1.36 ms per pass
// on one rank, profiles/round6_ab_runs.txt.)
// h_spec_flags: [0, 2F) the tie flags, [2F, 4F) the toss-ups.  Reference: syncdetector.c:26-153, dsp.c:96-110.
static int band_speculate(tsdrgpu_postproc_t *pp, int *ties, int *tossups)
{
    tsdrgpu_t *g = pp->g;
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    const int nmax = W > Htot ? W : Htot;
    hipStream_t st = g->stream;
    if (!pp->d_state_save && hipMalloc(&pp->d_state_save, sizeof(PpState)) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "band state copy");
    if (pp->cap_spec_flags < (size_t)4 * F) {
        if (pp->h_spec_flags) { (void)hipStreamSynchronize(st); (void)hipHostFree(pp->h_spec_flags); }
        pp->h_spec_flags = nullptr;
        pp->cap_spec_flags = 0;
        if (hipHostMalloc(&pp->h_spec_flags, sizeof(int) * 4 * (size_t)F, hipHostMallocDefault) != hipSuccess)
            return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "band flags (pinned)");
        pp->cap_spec_flags = (size_t)4 * F;
    }
    if (!pp->ev_spec && hipEventCreateWithFlags(&pp->ev_spec, hipEventDisableTiming) != hipSuccess) return tsdr_fail(g, TSDRGPU_EHIP, "postproc", "event");
    HIP_TRY(g, hipMemcpyAsync(pp->d_state_save, pp->d_state, sizeof(PpState), hipMemcpyDeviceToDevice, st));

    StripScratch sc;
    sc.nmax = nmax;
    sc.blur = pp->d_work;
    sc.prefix = (double *)(pp->d_work + (((size_t)F * 2 * nmax + 1) & ~(size_t)1));
    sc.total = sc.prefix + (size_t)F * 2 * (nmax + 1);
    SpecEntry *spec = (SpecEntry *)(sc.total + (size_t)F * 2);
    int *d_amb = pp->d_sflag + (size_t)F * 2, *d_fresh = d_amb + (size_t)F * 2, *d_redo = d_fresh + (size_t)F * 2;
    const int *const no_gate = nullptr;
    // stage 0 and stage 2 of the literal run, without its questions
    if (pp->band_fused) {  // (the range was exchanged and the recurrence run before the trip)
        TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack_sum, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_strip_x, pp->d_strip_y);
        KERNEL_CHECK(g, "k_band_unpack_sum");
    } else {
        TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_xmax, pp->d_strip_x,
                    pp->d_strip_y, pp->d_fmin, pp->d_fmax, pp->d_v0);
        TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, (const float *)pp->d_v0, 1LL, pp->d_fmin, pp->d_fmax, pp->d_state, pp->d_chain, 1,
                    prm->lowpasscoeff, (int *)nullptr);
        KERNEL_CHECK(g, "k_autogain_chain");
    }
    TSDR_LAUNCH(g, PROF_CHAIN, st, k_strip_flag, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, 1, pp->d_sflag);
    KERNEL_CHECK(g, "k_strip_flag");
    HIP_TRY(g, hipMemcpyAsync(pp->h_spec_flags, pp->d_sflag, sizeof(int) * 2 * (size_t)F, hipMemcpyDeviceToHost, st));
    // (a flagged strip makes k_strip_prepare take d_exact's entries, which nobody has filled yet: whatever the chain then decides
    // is thrown away — window indices stay inside the strip whatever the sums are)
    TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc, 1, pp->taps[0], pp->taps[1],
                pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, no_gate);
    TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, Htot, sc, pp->d_state, spec, no_gate, no_gate);
    TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, Htot, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1, d_amb, no_gate);
    KERNEL_CHECK(g, "k_sync_chain");
    TSDR_LAUNCH(g, PROF_CHAIN, st, k_redo_prepare, 1, 256, 2 * F, d_amb, pp->d_sflag, d_redo, d_fresh);
    KERNEL_CHECK(g, "k_redo_prepare");
    HIP_TRY(g, hipMemcpyAsync(pp->h_spec_flags + 2 * (size_t)F, d_fresh, sizeof(int) * 2 * (size_t)F, hipMemcpyDeviceToHost, st));
    HIP_TRY(g, hipEventRecord(pp->ev_spec, st));
    HIP_TRY(g, hipEventSynchronize(pp->ev_spec));
    int t = 0, u = 0;
    for (int i = 0; i < 2 * F; i++) {
        t |= pp->h_spec_flags[i];
        u |= pp->h_spec_flags[2 * F + i];
    }
    pp->band_spec_runs++;
    *ties = t ? 1 : 0;
    *tossups = u ? 1 : 0;
    if (t) {
        pp->band_spec_replays++;
        HIP_TRY(g, hipMemcpyAsync(pp->d_state, pp->d_state_save, sizeof(PpState), hipMemcpyDeviceToDevice, st));
    }
    return TSDRGPU_OK;
}

// the set entries of a flag array the host already holds -> the relay's item list on the device
static int band_items_from_host(tsdrgpu_postproc_t *pp, const int *h_flags, int count, int *nitems)
{
    tsdrgpu_t *g = pp->g;
    int rc;
    if (pp->cap_hflags < (size_t)count) {
        free(pp->h_flags);
        pp->h_flags = (int *)malloc(sizeof(int) * (size_t)count * 2);
        pp->cap_hflags = pp->h_flags ? (size_t)count : 0;
        if (!pp->h_flags) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "band flags");
    }
    if ((rc = ensure(g, &pp->d_items, &pp->cap_items, (size_t)count))) return rc;
    int *list = pp->h_flags + count;
    int n = 0;
    for (int i = 0; i < count; i++)
        if (h_flags[i]) list[n++] = i;
    if (n) {
        HIP_TRY(g, hipMemcpyAsync(pp->d_items, list, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, g->stream));
        HIP_TRY(g, hipStreamSynchronize(g->stream));  // `list` is reused
    }
    *nitems = n;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_band_spec_stats(tsdrgpu_postproc_t *pp, uint64_t *runs, uint64_t *replays)
{
    if (!pp) return TSDRGPU_EINVAL;
    if (runs) *runs = pp->band_spec_runs;
    if (replays) *replays = pp->band_spec_replays;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_band_advance(tsdrgpu_postproc_t *pp, float *d_out_band, int band_index, int nbands, double **d_buf,
                                             int64_t *n_buf, int *h_more, tsdrgpu_pp_frameinfo_t *h_info)
{
    if (!pp || !d_out_band || !h_more || nbands < 1 || band_index < 0 || band_index >= nbands)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_advance", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending != PEND_BAND) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_advance", "no band run is open");
    if (pp->band_fused == 1) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_advance", "the fused band run has not made its trip yet (tsdrgpu_postproc_band_fused)");
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    const int nmax = W > Htot ? W : Htot;
    hipStream_t st = g->stream;
    *h_more = 0;
    int rc;
    StripScratch sc;
    sc.nmax = nmax;
    sc.blur = pp->d_work;
    sc.prefix = (double *)(pp->d_work + (((size_t)F * 2 * nmax + 1) & ~(size_t)1));
    sc.total = sc.prefix + (size_t)F * 2 * (nmax + 1);
    SpecEntry *spec = (SpecEntry *)(sc.total + (size_t)F * 2);
    int *d_amb = pp->d_sflag + (size_t)F * 2, *d_fresh = d_amb + (size_t)F * 2, *d_redo = d_fresh + (size_t)F * 2;
    const int *const no_gate = nullptr;
    const int exact = pp->exact_ties;

    // TSDRGPU_BAND_SPECULATE=0: every batch takes the literal run (two host round trips; for A/B runs and for the tests of that run)
    static const int speculate = [] { const char *e = getenv("TSDRGPU_BAND_SPECULATE"); return e ? atoi(e) : 1; }();
    if (pp->band_stage == 0 && exact && speculate && !pp->band_spec) {
        int ties = 0, tossups = 0;
        if ((rc = band_speculate(pp, &ties, &tossups))) return rc;
        if (ties) {
            pp->band_spec = -1;  // from the start, literally; stage 0's question is answered (h_spec_flags)
        } else if (tossups) {
            if ((rc = band_items_from_host(pp, pp->h_spec_flags + 2 * (size_t)F, 2 * F, &pp->relay_items))) return rc;
            pp->relay_step = 0;
            pp->band_stage = 3;
        } else {
            pp->band_stage = 4;
        }
    }

    for (;;) {
        switch (pp->band_stage) {
        case 0: {  // the exchanged statistics -> autogain recurrence, tie flags
            if (pp->band_fused) {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack_sum, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_strip_x,
                            pp->d_strip_y);
                KERNEL_CHECK(g, "k_band_unpack_sum");
            } else {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_xmax,
                            pp->d_strip_x, pp->d_strip_y, pp->d_fmin, pp->d_fmax, pp->d_v0);
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, (const float *)pp->d_v0, 1LL, pp->d_fmin, pp->d_fmax, pp->d_state, pp->d_chain, 1,
                            prm->lowpasscoeff, (int *)nullptr);
                KERNEL_CHECK(g, "k_autogain_chain");
            }
            pp->relay_items = 0;
            if (exact) {
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_strip_flag, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, 1, pp->d_sflag);
                KERNEL_CHECK(g, "k_strip_flag");
                if (pp->band_spec == -1) rc = band_items_from_host(pp, pp->h_spec_flags, 2 * F, &pp->relay_items);  // (the same flags: the same strips)
                else rc = band_collect_items(pp, pp->d_sflag, 2 * F, &pp->relay_items);
                if (rc) return rc;
            } else {
                HIP_TRY(g, hipMemsetAsync(pp->d_sflag, 0, sizeof(int) * (size_t)F * 2, st));
            }
            pp->relay_step = 0;
            pp->band_stage = pp->relay_items ? 1 : 2;
            break;
        }
        case 1:
        case 3: {  // relay: one step per call, the caller all-reduces the buffer in between
            if (pp->relay_step == 0 && (rc = ensure(g, &pp->d_relay, &pp->cap_relay, (size_t)pp->relay_items * nmax))) return rc;
            if (pp->relay_step < nbands) {
                if ((rc = band_relay_step(pp, band_index))) return rc;
                pp->relay_step++;
                if (d_buf) *d_buf = pp->d_relay;
                if (n_buf) *n_buf = (int64_t)pp->relay_items * nmax;
                *h_more = 1;
                return TSDRGPU_OK;
            }
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_band_relay_take, dim3((nmax + 255) / 256, pp->relay_items), 256, pp->d_items, nmax, W, Htot, pp->d_relay,
                        pp->d_exact);
            KERNEL_CHECK(g, "k_band_relay_take");
            pp->band_stage = pp->band_stage == 1 ? 2 : 4;
            if (pp->band_stage == 4) {  // run 1 of the chain: only the strips that changed, from the saved state
                TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc, 1, pp->taps[0],
                            pp->taps[1], pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, (const int *)d_redo);
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, Htot, sc, pp->d_state + 1, spec, (const int *)d_redo, (const int *)d_fresh);
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, Htot, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1, d_amb,
                            (const int *)d_redo);
                KERNEL_CHECK(g, "k_sync_chain");
            }
            break;
        }
        case 2: {  // run 0 of the chain; with exact ties on, its toss-ups are the second relay's items
            TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc, 1, pp->taps[0],
                        pp->taps[1], pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, no_gate);
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, Htot, sc, pp->d_state, spec, no_gate, no_gate);
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, Htot, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1,
                        exact ? d_amb : (int *)nullptr, no_gate);
            KERNEL_CHECK(g, "k_sync_chain");
            pp->relay_items = 0;
            if (exact) {
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_redo_prepare, 1, 256, 2 * F, d_amb, pp->d_sflag, d_redo, d_fresh);
                KERNEL_CHECK(g, "k_redo_prepare");
                if ((rc = band_collect_items(pp, d_fresh, 2 * F, &pp->relay_items))) return rc;
            }
            pp->relay_step = 0;
            pp->band_stage = pp->relay_items ? 3 : 4;
            break;
        }
        default: {  // the pass over this band's rows
            const int fused = pp->band_fused;
            pp->pending = 0;
            pp->band_stage = 0;
            pp->band_spec = 0;
            pp->band_fused = 0;
            if ((rc = fused ? band_fused_lines(pp, d_out_band) : band_run_pass(pp, d_out_band))) return rc;
            if (h_info) return pp_copy_info(pp, F, h_info);
            return TSDRGPU_OK;
        }
        }
    }
}

// ---------------------------------------------------------------------------
// General band runs: every stage order of dsp_post_process (dsp.c:134-239), autoshift (syncdetector.c:187-207) and the
// frame-rate PLL (syncdetector.c:133-153) with the frame path sharded by rows.  tsdrgpu_postproc_band_open compiles the
// stage order into a short program of steps; tsdrgpu_postproc_band_step runs it until it needs the other ranks and tells
// the caller which collective to make (sum / max all-reduce of a small buffer, or an all-gather of the frames when the
// 2-D roll needs rows of other bands), then goes on.  The replicated parts (autogain recurrence, sync detector, PLL) see
// identical inputs on every rank, so every rank takes the same path and the same decisions.  Contract-exact like the
// single-GPU run: strips that hold ties or toss-ups are collapsed literally, relayed band by band.
// ---------------------------------------------------------------------------
enum { BOP_STATS = 1, BOP_XSUM, BOP_XMAX, BOP_UNPACK, BOP_AUTOGAIN, BOP_SYNC, BOP_GATHER, BOP_PASS, BOP_END };

// this band's rows of every frame into its slot of the gather buffer [band][F][rows_max][W]
__global__ __launch_bounds__(256) void k_band_gather_pack(const float *__restrict__ band, int F, int W, int rows, int rows_max, float *__restrict__ slot)
{
    const long long n = (long long)F * rows * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / ((long long)rows * W), r = i - f * rows * W;
        slot[f * rows_max * W + r] = band[i];
    }
}

// The pass with the 2-D roll (syncdetector.c:187-207) for a band: output pixel (x, y0 + y) of frame f takes the pixel
// (x + dx, y0 + y + dy) (both wrapped) of the FULL frame, found in the gather buffer through the band edges.
template <int FLAGS>
__global__ __launch_bounds__(256) void k_band_roll_pass(const float *__restrict__ gath, const int *__restrict__ edges, int nbands, int rows_max,
                                                        float *__restrict__ dst, long long dstride, int F, int W, int Htot, int y0, int rows,
                                                        const ChainOut *__restrict__ chain, float *__restrict__ screen, float a)
{
    __shared__ int ed[65];
    for (int i = threadIdx.x; i <= nbands; i += blockDim.x) ed[i] = edges[i];
    __syncthreads();
    const int Pb = W * rows;
    const double one_minus_a = 1.0 - a;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Pb; p += gridDim.x * blockDim.x) {
        const int y = p / W, x = p - y * W;
        float s = (FLAGS & PASS_IIR) ? screen[p] : 0.f;
        for (int f = 0; f < F; f++) {
            const int dx = chain[f].dx, dy = chain[f].dy;
            float lastmin = 0.f, span = 1.f;
            if (FLAGS & PASS_NORMALISE) { lastmin = chain[f].lastmin; span = chain[f].span; }
            int sx = x + dx; if (sx >= W) sx -= W;
            int sy = y0 + y + dy; if (sy >= Htot) sy -= Htot;
            int b = 0;
            while (b + 1 < nbands && sy >= ed[b + 1]) b++;
            const float v = gath[(((long long)b * F + f) * rows_max + (sy - ed[b])) * W + sx];
            dst[(long long)f * dstride + p] = pass_one(FLAGS, v, s, a, one_minus_a, lastmin, span, false);
        }
        if (FLAGS & PASS_IIR) screen[p] = s;
    }
}

extern "C" int tsdrgpu_postproc_band_open(tsdrgpu_postproc_t *pp, const float *d_band, int F, int W, int Htot, const int *edges, int nbands,
                                          int band_index, const tsdrgpu_pp_params_t *prm)
{
    if (!pp || !d_band || !prm || !edges || F <= 0 || W <= 0 || Htot <= 0 || nbands < 1 || nbands > 64 || band_index < 0 || band_index >= nbands)
        return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_open", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_open", "a split run is already open");
    if (edges[0] != 0 || edges[nbands] != Htot) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_open", "the band edges must run from 0 to the frame height");
    int rows_max = 0;
    for (int b = 0; b < nbands; b++) {
        if (edges[b + 1] <= edges[b] || edges[b] % TILE_H) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_open", "bands must be non-empty and start on multiples of 32 rows");
        if (edges[b + 1] - edges[b] > rows_max) rows_max = edges[b + 1] - edges[b];
        pp->bedges[b] = edges[b];
    }
    pp->bedges[nbands] = Htot;
    const int y0 = edges[band_index], rows = edges[band_index + 1] - y0;
    int rc;
    if ((rc = pp_prepare(pp, F, W, Htot, prm))) return rc;
    if ((rc = ensure(g, &pp->d_xsum, &pp->cap_xsum, (size_t)F * 3 * (W + Htot)))) return rc;
    if ((rc = ensure(g, &pp->d_xmax, &pp->cap_xmax, (size_t)F * 4))) return rc;
    if ((rc = ensure(g, &pp->d_v0, &pp->cap_v0, (size_t)F))) return rc;
    if ((rc = ensure(g, &pp->d_chain_band, &pp->cap_chain_band, (size_t)F))) return rc;
    if (!pp->d_bedges && hipMalloc(&pp->d_bedges, sizeof(int) * 65) != hipSuccess) return tsdr_fail(g, TSDRGPU_ENOMEM, "postproc", "band edges");
    HIP_TRY(g, hipMemcpyAsync(pp->d_bedges, pp->bedges, sizeof(int) * (size_t)(nbands + 1), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));  // (bedges may be rewritten by the next open)
    if (prm->autoshift && (rc = ensure(g, &pp->d_gather, &pp->cap_gather, (size_t)nbands * F * rows_max * W))) return rc;
    pp->p_frames = d_band;
    pp->p_F = F; pp->p_W = W; pp->p_H = Htot;
    pp->p_prm = *prm;
    pp->band_y0 = y0;
    pp->band_rows = rows;
    pp->bnbands = nbands;
    pp->bindex = band_index;
    pp->brows_max = rows_max;
    pp->bsrc[0] = d_band;
    pp->bsrc[1] = pp->d_tmp1;
    pp->bsrc[2] = pp->d_tmp2;
    pp->bsrc[3] = nullptr;  // the caller's output band, known at the first step

    // the program: dsp.c:134-239's four orders.  BOP_PASS: a = flags, b = source buffer, c = destination buffer
    const float a = prm->motionblur;
    const int lbs = prm->lowpass_before_sync, aap = prm->autogain_after_proc, roll = prm->autoshift ? PASS_ROLL : 0;
    int n = 0;
    auto emit = [&](int op, int x = 0, int y = 0, int z = 0) { pp->bprog[n].op = op; pp->bprog[n].a = x; pp->bprog[n].b = y; pp->bprog[n].c = z; n++; };
    // sync on the strips of buffer `src` (normalised strips or not), then the pass with the roll / lines
    auto sync_steps = [&](int src, int norm) {
        emit(BOP_STATS, src, 1); emit(BOP_XSUM); emit(BOP_UNPACK, 1);
        emit(BOP_SYNC, src, norm);
    };
    auto autogain_steps = [&](int src) {
        emit(BOP_STATS, src, 0); emit(BOP_XMAX); emit(BOP_UNPACK, 0);
        emit(BOP_AUTOGAIN, 1);
    };
    if (!lbs && !aap) {
        const int lines = (!prm->autoshift && a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        // one statistics read serves both: strips and min / max
        emit(BOP_STATS, 0, 1); emit(BOP_XSUM); emit(BOP_XMAX); emit(BOP_UNPACK, 2);
        emit(BOP_AUTOGAIN, 1);
        emit(BOP_SYNC, 0, 1);
        if (roll) emit(BOP_GATHER, 0);
        emit(BOP_PASS, PASS_NORMALISE | roll | lines | PASS_IIR, 0, 3);
    } else if (!lbs && aap) {
        const int lines = (!prm->autoshift && a == 0.0f && !prm->superresolution) ? PASS_LINES : 0;
        sync_steps(0, 0);
        if (roll) emit(BOP_GATHER, 0);
        emit(BOP_PASS, roll | lines | PASS_IIR, 0, 1);
        autogain_steps(1);
        emit(BOP_PASS, PASS_NORMALISE, 1, 3);
    } else if (lbs && !aap) {
        const int lines = (!prm->autoshift && !prm->superresolution) ? PASS_LINES : 0;
        autogain_steps(0);
        emit(BOP_PASS, PASS_NORMALISE | PASS_IIR, 0, 1);
        sync_steps(1, 0);
        if (roll) emit(BOP_GATHER, 1);
        emit(BOP_PASS, roll | lines, 1, 3);
    } else {
        const int lines = (!prm->autoshift && !prm->superresolution) ? PASS_LINES : 0;
        emit(BOP_PASS, PASS_IIR, 0, 1);
        sync_steps(1, 0);
        if (roll) emit(BOP_GATHER, 1);
        emit(BOP_PASS, roll | lines, 1, 2);
        autogain_steps(2);
        emit(BOP_PASS, PASS_NORMALISE, 2, 3);
    }
    emit(BOP_END);
    pp->bprog_n = n;
    pp->bprog_pc = 0;
    pp->bsync_stage = 0;
    pp->pending = PEND_BAND + 1;
    return TSDRGPU_OK;
}

// the sync detector on the exchanged strips, with the relays of the literal collapse; returns 1 when the caller has to
// all-reduce pp->d_relay (sum), 0 when the detector is done
static int band_sync_advance(tsdrgpu_postproc_t *pp, int *need_exchange)
{
    tsdrgpu_t *g = pp->g;
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    const int nmax = W > Htot ? W : Htot;
    const int norm = pp->bsync_norm;
    hipStream_t st = g->stream;
    StripScratch sc;
    sc.nmax = nmax;
    sc.blur = pp->d_work;
    sc.prefix = (double *)(pp->d_work + (((size_t)F * 2 * nmax + 1) & ~(size_t)1));
    sc.total = sc.prefix + (size_t)F * 2 * (nmax + 1);
    SpecEntry *spec = (SpecEntry *)(sc.total + (size_t)F * 2);
    int *d_amb = pp->d_sflag + (size_t)F * 2, *d_fresh = d_amb + (size_t)F * 2, *d_redo = d_fresh + (size_t)F * 2;
    const int *const no_gate = nullptr;
    const int exact = pp->exact_ties;
    int rc;
    *need_exchange = 0;
    for (;;) {
        switch (pp->bsync_stage) {
        case 0: {  // tie flags on the exchanged strips
            if (!pp->chain_has_autogain) {  // a sync-only step: the chain record repeats the carried autogain state
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, (const float *)pp->d_v0, 1LL, pp->d_fmin, pp->d_fmax, pp->d_state, pp->d_chain, 0,
                            prm->lowpasscoeff, (int *)nullptr);
                KERNEL_CHECK(g, "k_autogain_chain");
                pp->chain_has_autogain = 1;
            }
            pp->relay_items = 0;
            if (exact) {
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_strip_flag, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, norm, pp->d_sflag);
                KERNEL_CHECK(g, "k_strip_flag");
                if ((rc = band_collect_items(pp, pp->d_sflag, 2 * F, &pp->relay_items))) return rc;
            } else {
                HIP_TRY(g, hipMemsetAsync(pp->d_sflag, 0, sizeof(int) * (size_t)F * 2, st));
            }
            pp->relay_step = 0;
            pp->bsync_stage = pp->relay_items ? 1 : 2;
            break;
        }
        case 1:
        case 3: {  // relay: one step per call, the caller all-reduces the buffer in between
            if (pp->relay_step == 0 && (rc = ensure(g, &pp->d_relay, &pp->cap_relay, (size_t)pp->relay_items * nmax))) return rc;
            if (pp->relay_step < pp->bnbands) {
                if ((rc = band_relay_step(pp, pp->bindex))) return rc;
                pp->relay_step++;
                *need_exchange = 1;
                return TSDRGPU_OK;
            }
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_band_relay_take, dim3((nmax + 255) / 256, pp->relay_items), 256, pp->d_items, nmax, W, Htot, pp->d_relay,
                        pp->d_exact);
            KERNEL_CHECK(g, "k_band_relay_take");
            pp->bsync_stage = pp->bsync_stage == 1 ? 2 : 4;
            if (pp->bsync_stage == 4) {  // run 1 of the chain: only the strips that changed, from the saved state
                TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc, norm, pp->taps[0],
                            pp->taps[1], pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, (const int *)d_redo);
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, Htot, sc, pp->d_state + 1, spec, (const int *)d_redo, (const int *)d_fresh);
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, Htot, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1, d_amb,
                            (const int *)d_redo);
                KERNEL_CHECK(g, "k_sync_chain");
            }
            break;
        }
        case 2: {  // run 0 of the chain; with exact ties on, its toss-ups are the second relay's items
            TSDR_LAUNCH_STRIP_PREPARE(g, st, sc.nmax, dim3(2, F), CHAIN_T, W, Htot, pp->d_strip_x, pp->d_strip_y, pp->d_chain, sc, norm, pp->taps[0],
                        pp->taps[1], pp->taps[2], pp->taps[3], pp->taps[4], pp->d_sflag, pp->d_exact, no_gate);
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_search, dim3(2, F), SYNC_T, W, Htot, sc, pp->d_state, spec, no_gate, no_gate);
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_sync_chain, 2, SYNC_T, F, W, Htot, sc, pp->d_state, pp->d_chain, spec, prm->pll, pp->d_state + 1,
                        exact ? d_amb : (int *)nullptr, no_gate);
            KERNEL_CHECK(g, "k_sync_chain");
            pp->relay_items = 0;
            if (exact) {
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_redo_prepare, 1, 256, 2 * F, d_amb, pp->d_sflag, d_redo, d_fresh);
                KERNEL_CHECK(g, "k_redo_prepare");
                if ((rc = band_collect_items(pp, d_fresh, 2 * F, &pp->relay_items))) return rc;
            }
            pp->relay_step = 0;
            pp->bsync_stage = pp->relay_items ? 3 : 4;
            break;
        }
        default:
            pp->bsync_stage = 0;
            return TSDRGPU_OK;
        }
    }
}

extern "C" int tsdrgpu_postproc_band_step(tsdrgpu_postproc_t *pp, float *d_out_band, tsdrgpu_band_exchange_t *x, tsdrgpu_pp_frameinfo_t *h_info)
{
    if (!pp || !d_out_band || !x) return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_band_step", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending != PEND_BAND + 1) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_band_step", "no band run is open (tsdrgpu_postproc_band_open)");
    const int F = pp->p_F, W = pp->p_W, Htot = pp->p_H, y0 = pp->band_y0, rows = pp->band_rows;
    const tsdrgpu_pp_params_t *prm = &pp->p_prm;
    const long long Pb = (long long)W * rows;
    hipStream_t st = g->stream;
    pp->bsrc[3] = d_out_band;
    x->kind = TSDRGPU_BAND_DONE;
    x->d_buf = nullptr;
    x->count = 0;
    int rc;
    for (;;) {
        const auto op = pp->bprog[pp->bprog_pc];
        switch (op.op) {
        case BOP_STATS: {  // a = source buffer, b = with strips
            const float *src = pp->bsrc[op.a];
            if ((rc = launch_stats(pp, src, Pb, F, W, rows, op.b))) return rc;
            if (op.b) {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_pack, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, y0, rows, pp->d_strip_x, pp->d_strip_y,
                            pp->d_fmin, pp->d_fmax, src, Pb, pp->d_xsum, pp->d_xmax);
                KERNEL_CHECK(g, "k_band_pack");
            } else {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_pack_mm, (F + 255) / 256, 256, F, y0, pp->d_fmin, pp->d_fmax, src, Pb, pp->d_xmax);
                KERNEL_CHECK(g, "k_band_pack_mm");
            }
            pp->bprog_pc++;
            break;
        }
        case BOP_XSUM:
            pp->bprog_pc++;
            x->kind = TSDRGPU_BAND_SUM_F64; x->d_buf = pp->d_xsum; x->count = (int64_t)F * 3 * (W + Htot);
            return TSDRGPU_OK;
        case BOP_XMAX:
            pp->bprog_pc++;
            x->kind = TSDRGPU_BAND_MAX_F32; x->d_buf = pp->d_xmax; x->count = (int64_t)F * 4;
            return TSDRGPU_OK;
        case BOP_UNPACK:  // a: 0 = min/max, 1 = strips, 2 = both
            if (op.a != 0) {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack_sum, dim3((W + Htot + 255) / 256, 3, F), 256, F, W, Htot, pp->d_xsum, pp->d_strip_x, pp->d_strip_y);
                KERNEL_CHECK(g, "k_band_unpack_sum");
            }
            if (op.a != 1) {
                TSDR_LAUNCH(g, PROF_FRAME_REDUCE, st, k_band_unpack_mm, (F + 255) / 256, 256, F, pp->d_xmax, pp->d_fmin, pp->d_fmax, pp->d_v0);
                KERNEL_CHECK(g, "k_band_unpack_mm");
            }
            pp->bprog_pc++;
            break;
        case BOP_AUTOGAIN:
            TSDR_LAUNCH(g, PROF_CHAIN, st, k_autogain_chain, 1, 64, F, (const float *)pp->d_v0, 1LL, pp->d_fmin, pp->d_fmax, pp->d_state, pp->d_chain, 1,
                        prm->lowpasscoeff, (int *)nullptr);
            KERNEL_CHECK(g, "k_autogain_chain");
            pp->chain_has_autogain = 1;
            pp->bprog_pc++;
            break;
        case BOP_SYNC: {  // a = the buffer the strips came from, b = strips normalised
            pp->brelay_src = pp->bsrc[op.a];
            pp->bsync_norm = op.b;
            int need = 0;
            if ((rc = band_sync_advance(pp, &need))) return rc;
            if (need) {
                x->kind = TSDRGPU_BAND_SUM_F64; x->d_buf = pp->d_relay; x->count = (int64_t)pp->relay_items * (W > Htot ? W : Htot);
                return TSDRGPU_OK;
            }
            pp->bprog_pc++;
            break;
        }
        case BOP_GATHER: {  // a = the buffer whose frames are rolled
            float *slot = pp->d_gather + (size_t)pp->bindex * F * pp->brows_max * W;
            TSDR_LAUNCH(g, PROF_FRAME_PASS, st, k_band_gather_pack, (unsigned)(g->prop.multiProcessorCount * 8), 256, pp->bsrc[op.a], F, W, rows, pp->brows_max, slot);
            KERNEL_CHECK(g, "k_band_gather_pack");
            pp->bprog_pc++;
            x->kind = TSDRGPU_BAND_ALLGATHER_F32; x->d_buf = pp->d_gather; x->count = (int64_t)F * pp->brows_max * W;
            return TSDRGPU_OK;
        }
        case BOP_PASS: {  // a = flags, b = source, c = destination
            const float a = prm->motionblur;
            float *dst = const_cast<float *>(pp->bsrc[op.c]);
            if (op.a & PASS_ROLL) {
                const unsigned grid = (unsigned)(g->prop.multiProcessorCount * 8);
                const int fl = op.a & ~PASS_LINES;
                if (fl == (PASS_NORMALISE | PASS_ROLL | PASS_IIR))
                    TSDR_LAUNCH(g, PROF_FRAME_PASS, st, (k_band_roll_pass<PASS_NORMALISE | PASS_IIR>), grid, 256, pp->d_gather, pp->d_bedges, pp->bnbands, pp->brows_max, dst, Pb,
                                F, W, Htot, y0, rows, pp->d_chain, pp->d_screen, a);
                else if (fl == (PASS_ROLL | PASS_IIR))
                    TSDR_LAUNCH(g, PROF_FRAME_PASS, st, (k_band_roll_pass<PASS_IIR>), grid, 256, pp->d_gather, pp->d_bedges, pp->bnbands, pp->brows_max, dst, Pb, F, W, Htot,
                                y0, rows, pp->d_chain, pp->d_screen, a);
                else
                    TSDR_LAUNCH(g, PROF_FRAME_PASS, st, (k_band_roll_pass<0>), grid, 256, pp->d_gather, pp->d_bedges, pp->bnbands, pp->brows_max, dst, Pb, F, W, Htot, y0,
                                rows, pp->d_chain, pp->d_screen, a);
                KERNEL_CHECK(g, "k_band_roll_pass");
            } else {
                // the flat pass on the band as a frame of `rows` rows; the painted lines compare band-local row numbers
                TSDR_LAUNCH(g, PROF_CHAIN, st, k_band_chain, (F + 63) / 64, 64, pp->d_chain, pp->d_chain_band, F, y0);
                KERNEL_CHECK(g, "k_band_chain");
                ChainOut *full = pp->d_chain;
                pp->d_chain = pp->d_chain_band;
                rc = launch_pass(pp, op.a, pp->bsrc[op.b], Pb, dst, Pb, F, W, rows, a);
                pp->d_chain = full;
                if (rc) return rc;
            }
            pp->bprog_pc++;
            break;
        }
        default:  // BOP_END
            pp->pending = 0;
            pp->brelay_src = nullptr;
            if (h_info) return pp_copy_info(pp, F, h_info);
            return TSDRGPU_OK;
        }
    }
}

__global__ void k_info_pack(const ChainOut *__restrict__ chain, tsdrgpu_pp_frameinfo_t *__restrict__ out, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const ChainOut c = chain[f];
    tsdrgpu_pp_frameinfo_t o;
    o.lastmin = c.lastmin; o.lastmax = c.lastmax;
    o.dx = c.dx; o.vx = c.vx; o.stripx = c.stripx;
    o.dy = c.dy; o.vy = c.vy; o.stripy = c.stripy;
    o.locked = c.locked; o.pll_fired = c.pll_fired;
    o.avg_speed = c.avg_speed; o.frameratediff = c.frameratediff;
    out[f] = o;
}

extern "C" int tsdrgpu_postproc_info_pack(tsdrgpu_postproc_t *pp, tsdrgpu_pp_frameinfo_t *d_info, int nframes)
{
    if (!pp || !d_info || nframes < 0) return TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (nframes > pp->last_F || !pp->d_chain) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_info_pack", "no run of that many frames");
    if (pp->pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_info_pack", "a split run is open");
    if (nframes == 0) return TSDRGPU_OK;
    TSDR_LAUNCH(g, PROF_CHAIN, g->stream, k_info_pack, (nframes + 63) / 64, 64, pp->d_chain, d_info, nframes);
    KERNEL_CHECK(g, "k_info_pack");
    return TSDRGPU_OK;
}

// dsp_autogain_t.snr (dsp.c:69-93) as a by-product of every run from now on: one more read of the frames autogain
// works on, queued beside the chain.  Off by default: the reference computes the field and never reads it (dsp.c:234).
extern "C" int tsdrgpu_postproc_set_snr(tsdrgpu_postproc_t *pp, int on)
{
    if (!pp) return TSDRGPU_EINVAL;
    if (pp->pending) return tsdr_fail(pp->g, TSDRGPU_ESTATE, "tsdrgpu_postproc_set_snr", "a split run is open");
    pp->want_snr = on ? 1 : 0;
    pp->snr_frames = 0;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_snr(tsdrgpu_postproc_t *pp, float *h_snr, int nframes)
{
    if (!pp || !h_snr || nframes < 0) return pp ? tsdr_fail(pp->g, TSDRGPU_EINVAL, "tsdrgpu_postproc_snr", "bad argument") : TSDRGPU_EINVAL;
    tsdrgpu_t *g = pp->g;
    if (pp->pending) return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_snr", "a split run is open: call tsdrgpu_postproc_finish first");
    if (!pp->want_snr || nframes > pp->snr_frames)
        return tsdr_fail(g, TSDRGPU_ESTATE, "tsdrgpu_postproc_snr", "the last run did not produce that many values (tsdrgpu_postproc_set_snr, band runs do not)");
    if (nframes == 0) return TSDRGPU_OK;
    // the values were queued on the chain's stream, which the main stream has joined by the end of every run
    HIP_TRY(g, hipMemcpyAsync(h_snr, pp->d_snr, sizeof(float) * nframes, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_set_exact_ties(tsdrgpu_postproc_t *pp, int on)
{
    if (!pp) return TSDRGPU_EINVAL;
    pp->exact_ties = on ? 1 : 0;
    return TSDRGPU_OK;
}

// statistics of the last run's toss-up handling (synchronises): decisions marked as toss-ups by the first chain run,
// strips that were re-collapsed in the reference's order because of them, strips flagged for that up front
extern "C" int tsdrgpu_postproc_redo_stats(tsdrgpu_postproc_t *pp, int *h_tossups, int *h_recollapsed, int *h_flagged_upfront)
{
    if (!pp || !pp->d_sflag || pp->last_F <= 0) return TSDRGPU_ESTATE;
    tsdrgpu_t *g = pp->g;
    const int F = pp->last_F;
    int *h = (int *)malloc(sizeof(int) * (size_t)F * 6);
    if (!h) return TSDRGPU_ENOMEM;
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    HIP_TRY(g, hipMemcpyAsync(h, pp->d_sflag, sizeof(int) * (size_t)F * 6, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    int t = 0, r = 0, u = 0;
    for (int i = 0; i < 2 * F; i++) {
        t += h[2 * F + i] ? 1 : 0;
        r += (pp->exact_ties && h[4 * F + i]) ? 1 : 0;
        u += (h[i] && !(pp->exact_ties && h[4 * F + i])) ? 1 : 0;
    }
    free(h);
    if (h_tossups) *h_tossups = pp->exact_ties ? t : 0;
    if (h_recollapsed) *h_recollapsed = r;
    if (h_flagged_upfront) *h_flagged_upfront = u;
    return TSDRGPU_OK;
}

// the same, raw: h receives [strip flagged][toss-up][re-collapsed in run 2], 2*F ints each (frame-major, axis minor)
extern "C" int tsdrgpu_postproc_redo_raw(tsdrgpu_postproc_t *pp, int *h, int cap_ints, int *h_frames)
{
    if (!pp || !pp->d_sflag || pp->last_F <= 0 || !h) return TSDRGPU_ESTATE;
    tsdrgpu_t *g = pp->g;
    const int F = pp->last_F;
    if (cap_ints < 6 * F) return tsdr_fail(g, TSDRGPU_EINVAL, "tsdrgpu_postproc_redo_raw", "buffer too small");
    HIP_TRY(g, hipStreamSynchronize(g->stream2));
    HIP_TRY(g, hipMemcpyAsync(h, pp->d_sflag, sizeof(int) * (size_t)F * 6, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    if (h_frames) *h_frames = F;
    return TSDRGPU_OK;
}

extern "C" int tsdrgpu_postproc_strips(tsdrgpu_postproc_t *pp, float *h_colsum, float *h_rowsum)
{
    if (!pp || !pp->d_work || pp->width <= 0 || pp->last_F <= 0) return TSDRGPU_ESTATE;
    tsdrgpu_t *g = pp->g;
    const int W = pp->width, H = pp->height;
    const size_t nmax = (size_t)(W > H ? W : H);
    const float *last = pp->d_work + (size_t)(pp->last_F - 1) * 2 * nmax;  // blurred strips (+markers) of the last frame
    if (h_colsum) HIP_TRY(g, hipMemcpyAsync(h_colsum, last, sizeof(float) * W, hipMemcpyDeviceToHost, g->stream));
    if (h_rowsum) HIP_TRY(g, hipMemcpyAsync(h_rowsum, last + nmax, sizeof(float) * H, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    return TSDRGPU_OK;
}
