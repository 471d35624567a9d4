// Per-frame pipeline on a resident mosaic: Stitcher.warp_target + Stitcher.merge + the two bounding boxes of
// Stitcher.stitch (reference stitcher.py:216-225 -> utils/proc.py:48-82, :192-254, :34-45, :307-312) as ONE
// entry point that runs 3 (gradient blender) or 2 (alpha blender) dependent launches and returns every
// per-frame result in one small read-back.
//
// The reference builds two padded canvases per call and scans the whole canvas three times.  Here the mosaic
// lives in a caller-owned arena that is zero outside the mosaic; the frame is warped only over Q, a
// conservative canvas rectangle around the warped quad, and the merged canvas is never materialised: its
// bounding box is  bbox(old mosaic) U bbox(pixels written)  as long as no pixel went from gray != 0 to
// gray == 0 (counted on the device; the caller rescans in that rare case).
//
//   gradient blender (utils/proc.py:192-254)
//     K1 frame_warp_kernel      one workgroup per canvas row of Q: warped pixels -> staging image, the row's
//                               gray != 0 extent (-> bbox(warped), stitcher.py:220 / proc.py:194) and channel-0
//                               extent (-> left/right feather edges, proc.py:213-218), a 1-bit/pixel mask of
//                               channel 0 != 0; plain stores only, nothing to initialise
//     K2 frame_extent_kernel    every workgroup folds the row table into bbox(warped) and the ROI
//                               (bbox +- 50 px clipped to the canvas, proc.py:199-201); column extents of the
//                               horizontally feathered mask (proc.py:230-236) from the bit mask
//     K3 frame_gradient_kernel  feathered blend in place on the mosaic inside the ROI (proc.py:211-253), bbox
//                               of the written gray != 0 pixels, zeroed / dirty pixel counts
//   alpha blender (utils/proc.py:48-82)
//     K1 frame_warp_alpha_kernel  warp + blend in place, per-row tables of both bounding boxes and counts
//     K2 frame_alpha_finish_kernel  folds the tables
#include "isx_common.h"
#include "isx_image_dev.h"
#include <string.h>

// result block (int32), device and pinned host copies
enum {
    R_WBBOX = 0,   // 5: bbox of gray != 0 warped pixels, canvas coordinates {x0,y0,x1,y1,found}
    R_ROI = 5,     // 4: gradient ROI {gx0,gy0,gx1,gy1} (canvas coordinates)
    R_MBBOX = 9,   // 5: bbox of gray != 0 pixels among the pixels written into the mosaic
    R_ZEROED = 14, // pixels whose gray was != 0 before the blend and is 0 after it
    R_DIRTY = 15,  // pixels written with gray == 0 but a byte != 0 (must not survive outside the crop)
    R_ROWS_OK = 16, // gradient: every channel-0 != 0 pixel of the ROI's rows lies inside the ROI's columns
    R_DONE = 17,   // the blend kernel ran (0: nothing warped, or rows_ok == 0 -> the caller takes the slow path)
    R_INTS = ISX_FRAME_RESULT_INTS
};

struct isx_ctx {
    int device;
    uint8_t *wbuf;
    size_t wbuf_cap;
    int32_t *rows; // gradient: 4 x qh {GL, GR, L0, R0}; alpha: 8 x qh
    size_t rows_cap;
    int32_t *cols; // 2 x qw {top, bottom}
    size_t cols_cap;
    uint64_t *mask; // qh x nchunk x 4 ballots
    size_t mask_cap;
    int32_t *res_dev;
    int32_t *res_host; // pinned
};

template <typename T>
static int ctx_ensure(T **p, size_t *cap, size_t need) {
    if (*cap >= need) return ISX_OK;
    if (*p) ISX_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 2;
    ISX_HIP_TRY(hipMalloc((void **)p, want * sizeof(T)));
    *cap = want;
    return ISX_OK;
}

extern "C" int isx_ctx_create(isx_ctx **out) {
    ISX_REQUIRE(out, "null out");
    isx_ctx *c = new isx_ctx();
    memset(c, 0, sizeof(*c));
    if (hipGetDevice(&c->device) != hipSuccess || hipMalloc((void **)&c->res_dev, R_INTS * 4) != hipSuccess ||
        hipHostMalloc((void **)&c->res_host, R_INTS * 4, hipHostMallocDefault) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        if (c->res_dev) (void)hipFree(c->res_dev);
        delete c;
        return isx_fail(ISX_E_HIP, "isx_ctx_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return ISX_OK;
}

extern "C" int isx_ctx_destroy(isx_ctx *c) {
    if (!c) return ISX_OK;
    if (c->wbuf) (void)hipFree(c->wbuf);
    if (c->rows) (void)hipFree(c->rows);
    if (c->cols) (void)hipFree(c->cols);
    if (c->mask) (void)hipFree(c->mask);
    if (c->res_dev) (void)hipFree(c->res_dev);
    if (c->res_host) (void)hipHostFree(c->res_host);
    delete c;
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// K1 (gradient path).  grid = qh workgroups of 256; wave w takes the 256-pixel chunks w, w+4, ... of its row.
// ------------------------------------------------------------------------------------------------
template <int C, bool PERSP, bool CLASSIC>
__global__ void __launch_bounds__(256) frame_warp_kernel(const uint8_t *__restrict__ src, const WarpParams P,
                                                         uint8_t *__restrict__ wbuf, int nchunk,
                                                         int32_t *__restrict__ rows, uint64_t *__restrict__ mask,
                                                         int32_t *__restrict__ cols) {
    const int ly = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int gl = INT_MAX, gr = -1, l0 = INT_MAX, r0 = -1;
    for (int chunk = wave; chunk < nchunk; chunk += 4) {
        const int lx = chunk * 256 + 4 * lane;
        uint32_t px[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            px[i] = lx + i >= P.dw ? 0u
                    : CLASSIC  ? warp_pixel_classic<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly)
                               : warp_pixel<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly);
        if (lx < P.dw) quad_store<C>(wbuf + (int64_t)ly * P.dpitch, lx, P.dw, px);
        uint64_t b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int X = P.dx0 + lx + i;
            if (gray_px<C>(px[i]) != 0) {
                gl = min(gl, X);
                gr = max(gr, X);
            }
            const bool c0 = (px[i] & 255u) != 0u;
            if (c0) {
                l0 = min(l0, X);
                r0 = max(r0, X);
            }
            b[i] = __ballot(c0);
        }
        if (lane == 0) {
            uint64_t *m = mask + ((int64_t)ly * nchunk + chunk) * 4;
            m[0] = b[0], m[1] = b[1], m[2] = b[2], m[3] = b[3];
        }
    }
    __shared__ int red[4][4];
    gl = wave_min(gl), gr = wave_max(gr), l0 = wave_min(l0), r0 = wave_max(r0);
    if (lane == 0) red[wave][0] = gl, red[wave][1] = gr, red[wave][2] = l0, red[wave][3] = r0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            gl = min(gl, red[w][0]);
            gr = max(gr, red[w][1]);
            l0 = min(l0, red[w][2]);
            r0 = max(r0, red[w][3]);
        }
        rows[ly] = gl;
        rows[P.dh + ly] = gr;
        rows[2 * P.dh + ly] = l0;
        rows[3 * P.dh + ly] = r0;
    }
    // column-extent accumulators of K2 (this kernel precedes it on the stream)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P.dw; i += gridDim.x * 256) {
        cols[i] = INT_MAX;
        cols[P.dw + i] = -1;
    }
}

// block-wide min/max of four values; every thread of the 256-thread block calls it and gets the result
__device__ __forceinline__ void block_minmax4(int &a, int &b, int &c, int &d, int (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_min(a), b = wave_min(b), c = wave_max(c), d = wave_max(d);
    __syncthreads(); // red may still be read from a previous call
    if (lane == 0) red[wave][0] = a, red[wave][1] = b, red[wave][2] = c, red[wave][3] = d;
    __syncthreads();
    a = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    b = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
    c = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2]));
    d = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
}

// ------------------------------------------------------------------------------------------------
// K2 (gradient path).  grid = (nchunk, ceil(qh / 64)).  Every workgroup folds the row table (a few KB from
// L2) so that no grid-wide synchronisation is needed; workgroup (0,0) publishes the result block.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) frame_extent_kernel(const int32_t *__restrict__ rows, int qx0, int qy0, int qw,
                                                           int qh, int canvas_w, int canvas_h, int margin,
                                                           const uint64_t *__restrict__ mask, int nchunk,
                                                           int32_t *__restrict__ cols, int32_t *__restrict__ res) {
    __shared__ int red[4][4];
    __shared__ uint64_t wm[64][4];
    __shared__ int sl[64], sr[64];
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1;
    for (int ly = threadIdx.x; ly < qh; ly += 256) {
        const int gr = rows[qh + ly];
        if (gr >= 0) {
            x0 = min(x0, rows[ly]);
            x1 = max(x1, gr + 1);
            y0 = min(y0, qy0 + ly);
            y1 = max(y1, qy0 + ly + 1);
        }
    }
    block_minmax4(x0, y0, x1, y1, red);
    const bool found = x1 >= 0;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (!found) {
        if (first) {
            res[R_WBBOX + 0] = res[R_WBBOX + 1] = INT_MAX;
            res[R_WBBOX + 2] = res[R_WBBOX + 3] = -1;
            res[R_WBBOX + 4] = 0;
            res[R_ROI + 0] = res[R_ROI + 1] = res[R_ROI + 2] = res[R_ROI + 3] = 0;
            res[R_MBBOX + 0] = res[R_MBBOX + 1] = INT_MAX;
            res[R_MBBOX + 2] = res[R_MBBOX + 3] = -1;
            res[R_MBBOX + 4] = 0;
            res[R_ZEROED] = res[R_DIRTY] = 0;
            res[R_ROWS_OK] = 1;
            res[R_DONE] = 0;
        }
        return;
    }
    const int gx0 = max(0, x0 - margin), gy0 = max(0, y0 - margin);
    const int gx1 = min(canvas_w, x1 + margin), gy1 = min(canvas_h, y1 + margin);
    // rows_ok: no channel-0 pixel of an ROI row outside the ROI's columns (then the row extents K1 took
    // over all of Q are the ROI's own, proc.py:213-218)
    int bad = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int ly = max(0, gy0 - qy0) + threadIdx.x; ly < min(qh, gy1 - qy0); ly += 256) {
        const int r0 = rows[3 * qh + ly];
        if (r0 >= 0 && (rows[2 * qh + ly] < gx0 || r0 >= gx1)) bad = 1;
    }
    block_minmax4(d1, d2, bad, d3, red);
    if (first) {
        res[R_WBBOX + 0] = x0, res[R_WBBOX + 1] = y0, res[R_WBBOX + 2] = x1, res[R_WBBOX + 3] = y1, res[R_WBBOX + 4] = 1;
        res[R_ROI + 0] = gx0, res[R_ROI + 1] = gy0, res[R_ROI + 2] = gx1, res[R_ROI + 3] = gy1;
        res[R_MBBOX + 0] = res[R_MBBOX + 1] = INT_MAX;
        res[R_MBBOX + 2] = res[R_MBBOX + 3] = -1;
        res[R_MBBOX + 4] = 0;
        res[R_ZEROED] = res[R_DIRTY] = 0;
        res[R_ROWS_OK] = !bad;
        res[R_DONE] = 0;
    }
    if (bad) return;
    // column extents over this workgroup's 64 rows x 256 columns
    const int chunk = blockIdx.x, lyb = blockIdx.y * 64;
    {
        const int r = threadIdx.x >> 2, k = threadIdx.x & 3, ly = lyb + r;
        const bool in = ly < qh && qy0 + ly >= gy0 && qy0 + ly < gy1;
        wm[r][k] = in ? mask[((int64_t)ly * nchunk + chunk) * 4 + k] : 0ull;
        if (threadIdx.x < 64) {
            const int l2 = lyb + threadIdx.x;
            const bool in2 = l2 < qh;
            sl[threadIdx.x] = in2 ? rows[2 * qh + l2] : INT_MAX;
            sr[threadIdx.x] = in2 ? rows[3 * qh + l2] : -1;
        }
    }
    __syncthreads();
    const int lx = chunk * 256 + threadIdx.x, X = qx0 + lx;
    if (lx >= qw || X < gx0 || X >= gx1) return;
    const int i = threadIdx.x >> 2, k = threadIdx.x & 3;
    int top = INT_MAX, bot = -1;
    for (int r = 0; r < 64; r++) {
        // the horizontally feathered mask is > 0 where channel 0 is != 0 and x is neither edge of its row
        if (((wm[r][k] >> i) & 1ull) && X != sl[r] && X != sr[r]) {
            top = min(top, qy0 + lyb + r);
            bot = qy0 + lyb + r;
        }
    }
    if (bot >= 0) {
        atomicMin(&cols[lx], top);
        atomicMax(&cols[qw + lx], bot);
    }
}

// ------------------------------------------------------------------------------------------------
// K3 (gradient path).  Tile = 256 pixels x 8 rows of Q; only pixels inside the ROI are blended.
// mosaic_q = the arena pixel under canvas (qx0, qy0).
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) frame_gradient_kernel(const uint8_t *__restrict__ wbuf, int64_t wpitch,
                                                             uint8_t *mosaic_q, int64_t mpitch, int qx0, int qy0,
                                                             int qw, int qh, const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ cols, int feather, int ox0,
                                                             int oy0, int ox1, int oy1, int32_t *__restrict__ res) {
    if (!res[R_WBBOX + 4] || !res[R_ROWS_OK]) return;
    const int gx0 = res[R_ROI + 0], gy0 = res[R_ROI + 1], gx1 = res[R_ROI + 2], gy1 = res[R_ROI + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = blockIdx.x * 256 + 4 * lane, X = qx0 + lx;
    const int lyb = blockIdx.y * 8 + wave;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) res[R_DONE] = 1;
    // tile entirely outside the ROI (block-uniform)
    if (qx0 + (int)blockIdx.x * 256 >= gx1 || qx0 + (int)blockIdx.x * 256 + 256 <= gx0 || qy0 + (int)blockIdx.y * 8 >= gy1 ||
        qy0 + (int)blockIdx.y * 8 + 8 <= gy0)
        return;
    int minx = INT_MAX, miny = INT_MAX, maxx = -1, maxy = -1, zeroed = 0, dirty = 0;
    if (lx < qw && X < gx1 && X + 4 > gx0) {
        uint32_t pw[2][4], pr[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int ly = min(lyb + 4 * i, qh - 1);
            quad_load<C>(wbuf + (int64_t)ly * wpitch, lx, qw, pw[i]);
            quad_load<C>(mosaic_q + (int64_t)ly * mpitch, lx, qw, pr[i]);
        }
        const double f = (double)feather;
        int T[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xc = min(lx + k, qw - 1);
            T[k] = cols[xc];
            B[k] = cols[qw + xc];
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int ly = lyb + 4 * i, Y = qy0 + ly;
            if (ly >= qh || Y < gy0 || Y >= gy1) continue;
            const int L = rows[2 * qh + ly], R = rows[3 * qh + ly];
            uint32_t po[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xx = X + k;
                po[k] = pr[i][k];
                if (xx < gx0 || xx >= gx1 || lx + k >= qw) continue;
                // factor chain of proc.py:211-246 in ROI-relative form: [L, min(L+f, W)) and [max(0, R-f), R]
                double m = 1.0;
                if (R >= 0) {
                    const int xe = L + feather < gx1 ? L + feather : gx1;
                    if (xx >= L && xx < xe) m *= (double)(xx - L) / f;
                    const int xb = R - feather > gx0 ? R - feather : gx0;
                    if (xx >= xb && xx <= R) m *= (double)(R - xx) / f;
                }
                if (B[k] >= 0) {
                    const int ye = T[k] + feather < gy1 ? T[k] + feather : gy1;
                    if (Y >= T[k] && Y < ye) m *= (double)(Y - T[k]) / f;
                    const int yb = B[k] - feather > gy0 ? B[k] - feather : gy0;
                    if (Y >= yb && Y <= B[k]) m *= (double)(B[k] - Y) / f;
                }
                uint32_t o = 0;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int wv = (pw[i][k] >> (8 * c)) & 255, rv = (pr[i][k] >> (8 * c)) & 255;
                    const double mk = wv > 0 ? m : 0.0;
                    const double r = (double)wv * mk + (double)rv * (1 - mk);
                    o |= (uint32_t)(uint8_t)(int)r << (8 * c);
                }
                po[k] = o;
                const int go = gray_px<C>(o);
                if (go != 0) {
                    minx = min(minx, xx);
                    maxx = max(maxx, xx + 1);
                    miny = min(miny, Y);
                    maxy = max(maxy, Y + 1);
                } else {
                    if (o != 0u) dirty++;
                    if (gray_px<C>(pr[i][k]) != 0) zeroed++;
                }
            }
            quad_store<C>(mosaic_q + (int64_t)ly * mpitch, lx, qw, po);
        }
    }
    // merged bbox: only what extends the old mosaic's box [ox0,ox1) x [oy0,oy1) needs to be published
    minx = wave_min(minx), miny = wave_min(miny), maxx = wave_max(maxx), maxy = wave_max(maxy);
    if (lane == 0 && maxx >= 0) {
        const bool have_old = ox1 > ox0;
        if (!have_old || minx < ox0) atomicMin(&res[R_MBBOX + 0], minx);
        if (!have_old || miny < oy0) atomicMin(&res[R_MBBOX + 1], miny);
        if (!have_old || maxx > ox1) atomicMax(&res[R_MBBOX + 2], maxx);
        if (!have_old || maxy > oy1) atomicMax(&res[R_MBBOX + 3], maxy);
        if (!have_old || minx < ox0 || miny < oy0 || maxx > ox1 || maxy > oy1) res[R_MBBOX + 4] = 1;
    }
    const uint64_t bz = __ballot(zeroed != 0), bd = __ballot(dirty != 0);
    if (bz | bd) {
        for (int m = 1; m < 64; m <<= 1) {
            zeroed += __shfl_xor(zeroed, m);
            dirty += __shfl_xor(dirty, m);
        }
        if (lane == 0) {
            if (zeroed) atomicAdd(&res[R_ZEROED], zeroed);
            if (dirty) atomicAdd(&res[R_DIRTY], dirty);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// alpha path K1: warp + seamless_merge in place over all of Q (outside Q the warped image is 0 and the merged
// canvas equals the mosaic).  Per-row table (8 ints): warped gray extent, merged gray extent, zeroed, dirty.
// ------------------------------------------------------------------------------------------------
template <int C, bool PERSP, bool CLASSIC>
__global__ void __launch_bounds__(256) frame_warp_alpha_kernel(const uint8_t *__restrict__ src, const WarpParams P,
                                                               uint8_t *mosaic_q, int64_t mpitch, int nchunk,
                                                               double contrib, double one_minus,
                                                               int32_t *__restrict__ rows) {
    const int ly = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int gl = INT_MAX, gr = -1, ml = INT_MAX, mr = -1, zeroed = 0, dirty = 0;
    uint8_t *mrow = mosaic_q + (int64_t)ly * mpitch;
    for (int chunk = wave; chunk < nchunk; chunk += 4) {
        const int lx = chunk * 256 + 4 * lane;
        if (lx >= P.dw) continue;
        uint32_t px[4], pb[4], po[4];
        quad_load<C>(mrow, lx, P.dw, pb);
#pragma unroll
        for (int i = 0; i < 4; i++)
            px[i] = lx + i >= P.dw ? 0u
                    : CLASSIC  ? warp_pixel_classic<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly)
                               : warp_pixel<C, PERSP>(src, P, P.dx0 + lx + i, P.dy0 + ly);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            po[i] = alpha_px<C>(px[i], pb[i], contrib, one_minus);
            const int X = P.dx0 + lx + i;
            if (lx + i >= P.dw) continue;
            if (gray_px<C>(px[i]) != 0) {
                gl = min(gl, X);
                gr = max(gr, X);
            }
            if (gray_px<C>(po[i]) != 0) {
                ml = min(ml, X);
                mr = max(mr, X);
            } else {
                if (po[i] != 0u) dirty++;
                if (gray_px<C>(pb[i]) != 0) zeroed++;
            }
        }
        quad_store<C>(mrow, lx, P.dw, po);
    }
    __shared__ int red[4][6];
    gl = wave_min(gl), gr = wave_max(gr), ml = wave_min(ml), mr = wave_max(mr);
    for (int m = 1; m < 64; m <<= 1) {
        zeroed += __shfl_xor(zeroed, m);
        dirty += __shfl_xor(dirty, m);
    }
    if (lane == 0)
        red[wave][0] = gl, red[wave][1] = gr, red[wave][2] = ml, red[wave][3] = mr, red[wave][4] = zeroed,
        red[wave][5] = dirty;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            gl = min(gl, red[w][0]);
            gr = max(gr, red[w][1]);
            ml = min(ml, red[w][2]);
            mr = max(mr, red[w][3]);
            zeroed += red[w][4];
            dirty += red[w][5];
        }
        rows[ly] = gl;
        rows[P.dh + ly] = gr;
        rows[2 * P.dh + ly] = ml;
        rows[3 * P.dh + ly] = mr;
        rows[4 * P.dh + ly] = zeroed;
        rows[5 * P.dh + ly] = dirty;
    }
}

__global__ void __launch_bounds__(256) frame_alpha_finish_kernel(const int32_t *__restrict__ rows, int qy0, int qh,
                                                                 int32_t *__restrict__ res) {
    __shared__ int red[4][4];
    int x0 = INT_MAX, y0 = INT_MAX, x1 = -1, y1 = -1, mx0 = INT_MAX, my0 = INT_MAX, mx1 = -1, my1 = -1, z = 0, d = 0;
    for (int ly = threadIdx.x; ly < qh; ly += 256) {
        const int gr = rows[qh + ly], mr = rows[3 * qh + ly];
        if (gr >= 0) {
            x0 = min(x0, rows[ly]);
            x1 = max(x1, gr + 1);
            y0 = min(y0, qy0 + ly);
            y1 = max(y1, qy0 + ly + 1);
        }
        if (mr >= 0) {
            mx0 = min(mx0, rows[2 * qh + ly]);
            mx1 = max(mx1, mr + 1);
            my0 = min(my0, qy0 + ly);
            my1 = max(my1, qy0 + ly + 1);
        }
        z += rows[4 * qh + ly];
        d += rows[5 * qh + ly];
    }
    block_minmax4(x0, y0, x1, y1, red);
    block_minmax4(mx0, my0, mx1, my1, red);
    int zz = 0, dd = 0, z2 = z > 0x3FFFFFFF ? 0x3FFFFFFF : z, d2 = d > 0x3FFFFFFF ? 0x3FFFFFFF : d;
    block_minmax4(zz, dd, z2, d2, red); // max over threads is enough: the caller only tests != 0
    if (threadIdx.x == 0) {
        res[R_WBBOX + 0] = x0, res[R_WBBOX + 1] = y0, res[R_WBBOX + 2] = x1, res[R_WBBOX + 3] = y1;
        res[R_WBBOX + 4] = x1 >= 0;
        res[R_ROI + 0] = res[R_ROI + 1] = res[R_ROI + 2] = res[R_ROI + 3] = 0;
        res[R_MBBOX + 0] = mx0, res[R_MBBOX + 1] = my0, res[R_MBBOX + 2] = mx1, res[R_MBBOX + 3] = my1;
        res[R_MBBOX + 4] = mx1 >= 0;
        res[R_ZEROED] = z2;
        res[R_DIRTY] = d2;
        res[R_ROWS_OK] = 1;
        res[R_DONE] = 1;
    }
}

static int fill_warp_params(WarpParams &P, const double *M_host, int model, int arith, int canvas_w, int canvas_h,
                            int sh, int sw, int64_t spitch, int src_x, int src_y, int qx0, int qy0, int qw, int qh,
                            int64_t dpitch) {
    double iM[9];
    if (model == ISX_MODEL_HOMOGRAPHY)
        invert_3x3_cv(M_host, iM);
    else
        invert_affine_cv(M_host, iM);
    for (int i = 0; i < 9; i++) P.m[i] = (float)iM[i], P.im[i] = iM[i];
    P.bw0 = 64;
    if (arith == ISX_WARP_CLASSIC && model == ISX_MODEL_HOMOGRAPHY) {
        const int bh0 = canvas_h < 16 ? canvas_h : 16;
        P.bw0 = 1024 / bh0 < canvas_w ? 1024 / bh0 : canvas_w;
    }
    P.src_x = src_x, P.src_y = src_y, P.sh = sh, P.sw = sw;
    P.dx0 = qx0, P.dy0 = qy0, P.dw = qw, P.dh = qh;
    P.spitch = spitch, P.dpitch = dpitch;
    return ISX_OK;
}

extern "C" int isx_frame_blend(isx_ctx *ctx, const uint8_t *src_dev, int sh, int sw, int c, int64_t spitch, int src_x,
                               int src_y, const double *M_host, int model, int arith, int canvas_w, int canvas_h,
                               uint8_t *mosaic_q_dev, int64_t mpitch, int q_x0, int q_y0, int q_w, int q_h,
                               const int32_t *old_bbox_host, int blender, double ref_contrib, int feather,
                               int32_t *result_host, void *stream) {
    ISX_REQUIRE(ctx && result_host, "null ctx / result_host");
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(model == ISX_MODEL_AFFINE_PARTIAL || model == ISX_MODEL_HOMOGRAPHY, "bad model %d", model);
    ISX_REQUIRE(arith == ISX_WARP_LINEAR_F32 || arith == ISX_WARP_CLASSIC, "bad warp arithmetic %d", arith);
    ISX_REQUIRE(blender == ISX_BLEND_ALPHA || blender == ISX_BLEND_GRADIENT, "bad blender %d", blender);
    ISX_REQUIRE(M_host, "null M_host");
    ISX_REQUIRE(sh > 0 && sw > 0 && spitch >= (int64_t)sw * c, "bad source geometry");
    ISX_REQUIRE(canvas_w > 0 && canvas_h > 0 && q_x0 >= 0 && q_y0 >= 0 && q_w >= 0 && q_h >= 0 &&
                    q_x0 + q_w <= canvas_w && q_y0 + q_h <= canvas_h,
                "Q = (%d,%d %dx%d) outside the %dx%d canvas", q_x0, q_y0, q_w, q_h, canvas_w, canvas_h);
    ISX_REQUIRE(q_w <= ISX_FRAME_MAX_Q && q_h <= ISX_FRAME_MAX_Q, "Q larger than %d pixels on a side", ISX_FRAME_MAX_Q);
    ISX_REQUIRE(blender == ISX_BLEND_ALPHA || feather > 0, "feather must be positive");
    int cur = -1;
    ISX_HIP_TRY(hipGetDevice(&cur));
    ISX_REQUIRE(cur == ctx->device, "ctx belongs to device %d but device %d is current", ctx->device, cur);
    hipStream_t st = (hipStream_t)stream;
    int32_t *r = result_host;
    if (q_w == 0 || q_h == 0) { // nothing of the frame lands on the canvas
        memset(r, 0, R_INTS * 4);
        r[R_WBBOX + 0] = r[R_WBBOX + 1] = r[R_MBBOX + 0] = r[R_MBBOX + 1] = INT_MAX;
        r[R_WBBOX + 2] = r[R_WBBOX + 3] = r[R_MBBOX + 2] = r[R_MBBOX + 3] = -1;
        r[R_ROWS_OK] = 1;
        return ISX_OK;
    }
    ISX_REQUIRE(src_dev && mosaic_q_dev && mpitch >= (int64_t)q_w * c, "null device pointer / bad mosaic pitch");
    int ob[4] = {0, 0, 0, 0};
    if (old_bbox_host) memcpy(ob, old_bbox_host, sizeof(ob));
    const int nchunk = (q_w + 255) / 256;
    const bool persp = model == ISX_MODEL_HOMOGRAPHY, classic = arith == ISX_WARP_CLASSIC;
    int rc;
    WarpParams P;
    if (blender == ISX_BLEND_GRADIENT) {
        const int64_t wpitch = (int64_t)q_w * c;
        if ((rc = ctx_ensure(&ctx->wbuf, &ctx->wbuf_cap, (size_t)wpitch * q_h + 16)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->rows, &ctx->rows_cap, (size_t)8 * q_h)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->cols, &ctx->cols_cap, (size_t)2 * q_w)) != ISX_OK) return rc;
        if ((rc = ctx_ensure(&ctx->mask, &ctx->mask_cap, (size_t)q_h * nchunk * 4)) != ISX_OK) return rc;
        fill_warp_params(P, M_host, model, arith, canvas_w, canvas_h, sh, sw, spitch, src_x, src_y, q_x0, q_y0, q_w, q_h, wpitch);
#define ISX_FW_LAUNCH(CC, PP, AA)                                                                                   \
    hipLaunchKernelGGL((frame_warp_kernel<CC, PP, AA>), dim3(q_h), dim3(256), 0, st, src_dev, P, ctx->wbuf, nchunk, \
                       ctx->rows, ctx->mask, ctx->cols)
        if (c == 3) {
            if (persp && classic) ISX_FW_LAUNCH(3, true, true);
            else if (persp) ISX_FW_LAUNCH(3, true, false);
            else if (classic) ISX_FW_LAUNCH(3, false, true);
            else ISX_FW_LAUNCH(3, false, false);
        } else {
            if (persp && classic) ISX_FW_LAUNCH(1, true, true);
            else if (persp) ISX_FW_LAUNCH(1, true, false);
            else if (classic) ISX_FW_LAUNCH(1, false, true);
            else ISX_FW_LAUNCH(1, false, false);
        }
#undef ISX_FW_LAUNCH
        hipLaunchKernelGGL(frame_extent_kernel, dim3(nchunk, (q_h + 63) / 64), dim3(256), 0, st, ctx->rows, q_x0, q_y0, q_w,
                           q_h, canvas_w, canvas_h, 50, ctx->mask, nchunk, ctx->cols, ctx->res_dev);
        dim3 gb(nchunk, (q_h + 7) / 8);
        if (c == 3)
            hipLaunchKernelGGL((frame_gradient_kernel<3>), gb, dim3(256), 0, st, ctx->wbuf, wpitch, mosaic_q_dev, mpitch, q_x0,
                               q_y0, q_w, q_h, ctx->rows, ctx->cols, feather, ob[0], ob[1], ob[2], ob[3], ctx->res_dev);
        else
            hipLaunchKernelGGL((frame_gradient_kernel<1>), gb, dim3(256), 0, st, ctx->wbuf, wpitch, mosaic_q_dev, mpitch, q_x0,
                               q_y0, q_w, q_h, ctx->rows, ctx->cols, feather, ob[0], ob[1], ob[2], ob[3], ctx->res_dev);
    } else {
        if ((rc = ctx_ensure(&ctx->rows, &ctx->rows_cap, (size_t)8 * q_h)) != ISX_OK) return rc;
        fill_warp_params(P, M_host, model, arith, canvas_w, canvas_h, sh, sw, spitch, src_x, src_y, q_x0, q_y0, q_w, q_h, mpitch);
        const double one_minus = 1 - ref_contrib; // Python: (1-ref_image_contrib)
#define ISX_FA_LAUNCH(CC, PP, AA)                                                                                      \
    hipLaunchKernelGGL((frame_warp_alpha_kernel<CC, PP, AA>), dim3(q_h), dim3(256), 0, st, src_dev, P, mosaic_q_dev, mpitch, \
                       nchunk, ref_contrib, one_minus, ctx->rows)
        if (c == 3) {
            if (persp && classic) ISX_FA_LAUNCH(3, true, true);
            else if (persp) ISX_FA_LAUNCH(3, true, false);
            else if (classic) ISX_FA_LAUNCH(3, false, true);
            else ISX_FA_LAUNCH(3, false, false);
        } else {
            if (persp && classic) ISX_FA_LAUNCH(1, true, true);
            else if (persp) ISX_FA_LAUNCH(1, true, false);
            else if (classic) ISX_FA_LAUNCH(1, false, true);
            else ISX_FA_LAUNCH(1, false, false);
        }
#undef ISX_FA_LAUNCH
        hipLaunchKernelGGL(frame_alpha_finish_kernel, dim3(1), dim3(256), 0, st, ctx->rows, q_y0, q_h, ctx->res_dev);
    }
    if ((rc = isx_launch_status("frame kernels")) != ISX_OK) return rc;
    ISX_HIP_TRY(hipMemcpyAsync(ctx->res_host, ctx->res_dev, R_INTS * 4, hipMemcpyDeviceToHost, st));
    ISX_HIP_TRY(hipStreamSynchronize(st));
    memcpy(r, ctx->res_host, R_INTS * 4);
    return ISX_OK;
}

// The staging image of the last gradient-path call (the warped frame over Q, pitch q_w * c): what the caller's
// slow path blends from when the result says rows_ok == 0.
extern "C" int isx_frame_warped(isx_ctx *ctx, const uint8_t **wbuf_dev_out) {
    ISX_REQUIRE(ctx && wbuf_dev_out, "null argument");
    *wbuf_dev_out = ctx->wbuf;
    return ISX_OK;
}
