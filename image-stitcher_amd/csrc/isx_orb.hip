// On-GPU ORB-style feature detection.  Stands where the reference calls
// cv2.ORB_create().detectAndCompute (stitcher.py:92-100, :163, :193-194, :263) -- but it is NOT cv2.ORB: cv2's
// learned pair pattern, pyramid and float arithmetic cannot be restated here, so parity with cv2 is unpinned by
// construction and the stitchers use this detector only when asked to (detector=HipOrb()).  What it computes is
// DEFINED, in integer arithmetic, by tests/np_orb.py, and these kernels match that definition bit for bit:
//
//   K1 orb_gray_hblur_kernel   gray (the library's BGR2GRAY), horizontal 7-tap blur sums (u16), 1 bit/pixel gray == 0
//   K2 orb_vblur_kernel        vertical 7 taps + the one rounding -> blurred u8
//   K3 orb_fast_harris_kernel  FAST-9 on the radius-3 circle (compass-point quick reject per pixel, full test on the
//                              survivors); the tile's corners are then scored
//                              wave-cooperatively: Harris R = 25(ac - b^2) - (a + c)^2 over a 7x7 block of Sobel products
//                              (int64) and the "no black pixel within `clean` px" test on the bit rows; key = bits of
//                              float32(R) into a per-pixel key map
//   K4 orb_nms_kernel          radius-3 non-maximum suppression on the key map (ties to the lower index); the survivors of
//                              a 32x32 tile go to its 64 slots (two survivors are >= 4 px apart, so 64 always suffice)
//   (host: the nfeatures largest keys, ties to the lower index -- a few thousand candidates)
//   K5 orb_describe_kernel     one wave per keypoint: intensity-centroid moments over the radius-15 disc, orientation
//                              bin = argmax of 32 integer dot products, 256 steered-BRIEF bits on the blurred image
//
// Everything is HBM/L2-bound byte work on a few MB; no MFMA.  Single scale (no pyramid).
#include "isx_common.h"
#include "isx_image_dev.h"

#define ORB_E 16      // pixels closer than this to the border are never keypoints (patch radius 15)
#define ORB_TILE 32   // NMS tile
#define ORB_SLOTS 64  // survivors per NMS tile
#define ORB_TW 64     // K3 tile: 64 x 16 pixels, 4 per thread
#define ORB_TH 16
#define ORB_HALO 4    // FAST radius 3, Harris radius 3 + 1
#define ORB_MAX_CORNERS 1024

// (constexpr, not __constant__: with the loop unrolled the offsets fold into the LDS instructions' immediates)
static constexpr int c_orb_circle[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0},  {3, 1},   {2, 2},   {1, 3},
                                            {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};
__constant__ int c_orb_dirs[32][2] = {
    {16384, 0},      {16069, 3196},    {15137, 6270},    {13623, 9102},    {11585, 11585},   {9102, 13623},   {6270, 15137},
    {3196, 16069},   {0, 16384},       {-3196, 16069},   {-6270, 15137},   {-9102, 13623},   {-11585, 11585}, {-13623, 9102},
    {-15137, 6270},  {-16069, 3196},   {-16384, 0},      {-16069, -3196},  {-15137, -6270},  {-13623, -9102}, {-11585, -11585},
    {-9102, -13623}, {-6270, -15137},  {-3196, -16069},  {0, -16384},      {3196, -16069},   {6270, -15137},  {9102, -13623},
    {11585, -11585}, {13623, -9102},   {15137, -6270},   {16069, -3196}};
__constant__ int c_orb_umax[16] = {15, 14, 14, 14, 14, 14, 13, 13, 12, 12, 11, 10, 9, 7, 5, 0};

// workspace layout (all offsets multiples of 256 bytes)
struct OrbWs {
    uint8_t *g, *bl;
    uint16_t *hs;
    uint32_t *key;
    uint64_t *zb;
    int zw; // u64 words per bit row
};
static inline int64_t orb_align(int64_t v) { return (v + 255) & ~(int64_t)255; }
static inline int64_t orb_ws_bytes(int h, int w) {
    const int64_t n = (int64_t)h * w, zw = (w + 63) / 64;
    return orb_align(n) * 2 + orb_align(n * 2) + orb_align(n * 4) + orb_align((int64_t)h * zw * 8);
}
static inline OrbWs orb_ws(void *base, int h, int w) {
    const int64_t n = (int64_t)h * w;
    OrbWs s;
    uint8_t *p = (uint8_t *)base;
    s.g = p, p += orb_align(n);
    s.bl = p, p += orb_align(n);
    s.hs = (uint16_t *)p, p += orb_align(n * 2);
    s.key = (uint32_t *)p, p += orb_align(n * 4);
    s.zb = (uint64_t *)p;
    s.zw = (w + 63) / 64;
    return s;
}

extern "C" int64_t isx_orb_workspace_bytes(int h, int w) {
    if (h <= 0 || w <= 0 || h > 32767 || w > 32767) return 0;
    return orb_ws_bytes(h, w);
}

template <int C>
__device__ __forceinline__ int orb_gray_at(const uint8_t *__restrict__ img, int64_t pitch, int x, int y) {
    const uint8_t *p = img + (int64_t)y * pitch + (int64_t)x * C;
    if (C == 1) return p[0];
    return gray_px<3>((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
}

// K1: grid (ceil(w/256), h), block 256
template <int C>
__global__ void __launch_bounds__(256) orb_gray_hblur_kernel(const uint8_t *__restrict__ img, int h, int w, int64_t pitch,
                                                             uint8_t *__restrict__ g, uint16_t *__restrict__ hs,
                                                             uint64_t *__restrict__ zb, int zw) {
    __shared__ int sm[256 + 6];
    const int y = blockIdx.y, x0 = blockIdx.x * 256, tid = threadIdx.x, x = x0 + tid;
    const int gv = orb_gray_at<C>(img, pitch, min(x, w - 1), y);
    sm[tid + 3] = gv;
    if (tid < 3) sm[tid] = orb_gray_at<C>(img, pitch, min(max(x0 + tid - 3, 0), w - 1), y);
    if (tid >= 253) sm[tid + 6] = orb_gray_at<C>(img, pitch, min(x0 + tid + 3, w - 1), y);
    __syncthreads();
    const int s = 18 * (sm[tid] + sm[tid + 6]) + 34 * (sm[tid + 1] + sm[tid + 5]) + 49 * (sm[tid + 2] + sm[tid + 4]) + 54 * sm[tid + 3];
    if (x < w) {
        g[(int64_t)y * w + x] = (uint8_t)gv;
        hs[(int64_t)y * w + x] = (uint16_t)s;
    }
    const uint64_t zbits = __ballot(x < w && gv == 0);
    if ((tid & 63) == 0 && x < w) zb[(int64_t)y * zw + (x >> 6)] = zbits;
}

// K2: grid (ceil(w/256), h), block 256
__global__ void __launch_bounds__(256) orb_vblur_kernel(const uint16_t *__restrict__ hs, int h, int w, uint8_t *__restrict__ bl) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int r[7];
#pragma unroll
    for (int j = 0; j < 7; j++) r[j] = hs[(int64_t)min(max(y + j - 3, 0), h - 1) * w + x];
    const int s = 18 * (r[0] + r[6]) + 34 * (r[1] + r[5]) + 49 * (r[2] + r[4]) + 54 * r[3];
    bl[(int64_t)y * w + x] = (uint8_t)((s + 32768) >> 16);
}

__device__ __forceinline__ bool orb_run9(uint32_t m) {
    uint32_t x = m | (m << 16);
    x &= x >> 1;
    x &= x >> 2;
    x &= x >> 4;
    x &= x >> 1;
    return x != 0;
}

// any bit set in columns [x0, x1] (inclusive, clipped to the row) of one bit row
__device__ __forceinline__ bool orb_row_has_zero(const uint64_t *__restrict__ row, int zw, int x0, int x1) {
    const int w0 = x0 >> 6, w1 = x1 >> 6;
    bool any = false;
    for (int k = w0; k <= w1; k++) {
        uint64_t v = row[k];
        if (k == w0) v &= ~0ull << (x0 & 63);
        if (k == w1 && (x1 & 63) != 63) v &= (1ull << ((x1 & 63) + 1)) - 1ull;
        any = any || v != 0ull;
    }
    return any;
}

// K3: grid (ceil(w/64), ceil(h/16)), block 256.  Writes EVERY pixel of the key map (0 = not a keypoint candidate).
__device__ __forceinline__ int orb_row_sum(int v) { // sum over each row of 16 lanes, in every lane of the row
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    return v;
}
__device__ __forceinline__ int orb_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); // row_mirror: every lane holds its row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__global__ void __launch_bounds__(256) orb_fast_harris_kernel(const uint8_t *__restrict__ g, int h, int w, int thr, int clean,
                                                              const uint64_t *__restrict__ zb, int zw,
                                                              uint32_t *__restrict__ key) {
    __shared__ __align__(16) uint8_t t[ORB_TH + 2 * ORB_HALO][ORB_TW + 2 * ORB_HALO];
    __shared__ int corners[ORB_MAX_CORNERS], cands[ORB_MAX_CORNERS];
    __shared__ int n_corners, n_cand;
    // the bit rows (gray == 0) every `clean` window of this tile can touch: rows Y0-16 .. Y0+31, three 64-column words
    // around the tile's own (X0 is a multiple of 64) -- staged once, so scoring a corner waits on no global load
    __shared__ uint64_t zt[ORB_TH + 2 * ORB_E][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = blockIdx.x * ORB_TW, Y0 = blockIdx.y * ORB_TH;
    if (tid == 0) n_corners = n_cand = 0;
    if (tid < (ORB_TH + 2 * ORB_E) * 3) {
        const int r = tid / 3, k = tid % 3, yy = Y0 - ORB_E + r, wd = (X0 >> 6) - 1 + k;
        zt[r][k] = (yy >= 0 && yy < h && wd >= 0 && wd < zw) ? zb[(int64_t)yy * zw + wd] : 0ull;
    }
    constexpr int TWH = ORB_TW + 2 * ORB_HALO, THH = ORB_TH + 2 * ORB_HALO;
    if (X0 >= ORB_HALO && X0 + ORB_TW + ORB_HALO <= w && Y0 >= ORB_HALO && Y0 + ORB_TH + ORB_HALO <= h) {
        // interior tile: 18 unaligned dword loads per row, both rounds in flight together (the byte-wise loop below
        // serialises seven memory round trips per workgroup and set the kernel's time)
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int q = tid + 256 * it;
            if (q < THH * (TWH / 4)) {
                const int ly = q / (TWH / 4), c4 = q % (TWH / 4);
                *(uint32_t *)&t[ly][4 * c4] = *(const u32u *)(g + (int64_t)(Y0 - ORB_HALO + ly) * w + (X0 - ORB_HALO + 4 * c4));
            }
        }
    } else {
        for (int p = tid; p < THH * TWH; p += 256) {
            const int ly = p / TWH, lx = p % TWH;
            const int yy = min(max(Y0 + ly - ORB_HALO, 0), h - 1), xx = min(max(X0 + lx - ORB_HALO, 0), w - 1);
            t[ly][lx] = g[(int64_t)yy * w + xx];
        }
    }
    __syncthreads();
    // phase 1a -- quick reject on the four compass points of the circle (indices 0, 4, 8, 12): any 9 contiguous points
    // contain at least two of them, so a corner has >= 2 compass points brighter (or darker).  The survivors go to a
    // list; the full 16-point test then runs on the list only, spread evenly over the threads (the per-pixel full test
    // is 4-5x the vector work and set this kernel's time)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + 256 * i, ly = p >> 6, lx = p & 63;
        const int X = X0 + lx, Y = Y0 + ly;
        if (X < w && Y < h) key[(int64_t)Y * w + X] = 0u;
        bool cand = false;
        if (X >= ORB_E && X < w - ORB_E && Y >= ORB_E && Y < h - ORB_E) {
            const int c = t[ly + ORB_HALO][lx + ORB_HALO];
            const int q0 = t[ly + ORB_HALO - 3][lx + ORB_HALO], q4 = t[ly + ORB_HALO][lx + ORB_HALO + 3];
            const int q8 = t[ly + ORB_HALO + 3][lx + ORB_HALO], q12 = t[ly + ORB_HALO][lx + ORB_HALO - 3];
            const int hi = c + thr, lo = c - thr;
            const int nb = (q0 > hi) + (q4 > hi) + (q8 > hi) + (q12 > hi), nd = (q0 < lo) + (q4 < lo) + (q8 < lo) + (q12 < lo);
            cand = nb >= 2 || nd >= 2;
        }
        const uint64_t m = __ballot(cand);
        if (m) { // one LDS atomic per wave; the order of the list does not matter
            const int first = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&n_cand, __popcll(m));
            base = __shfl(base, first);
            if (cand) cands[base + __popcll(m & ((1ull << lane) - 1ull))] = p; // <= 1024 pixels per tile: cannot overflow
        }
    }
    __syncthreads();
    // phase 1b -- FAST-9 on the candidates
    const int ncand = n_cand;
    for (int ci = tid; ci < ncand; ci += 256) {
        const int p = cands[ci], ly = p >> 6, lx = p & 63;
        const int c = t[ly + ORB_HALO][lx + ORB_HALO];
        uint32_t br = 0, dk = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int q = t[ly + ORB_HALO + c_orb_circle[k][1]][lx + ORB_HALO + c_orb_circle[k][0]];
            br |= (uint32_t)(q > c + thr) << k;
            dk |= (uint32_t)(q < c - thr) << k;
        }
        if (orb_run9(br) || orb_run9(dk)) {
            const int slot = atomicAdd(&n_corners, 1); // (LDS atomic, corners are few)
            corners[slot] = p;
        }
    }
    __syncthreads();
    const int nc = n_corners;
    // one wave per corner: lanes 0..48 take one position of the 7x7 block each, lanes 0..2*clean take one bit row each
    for (int ci = wave; ci < nc; ci += 4) {
        const int p = corners[ci], ly = p >> 6, lx = p & 63;
        const int X = X0 + lx, Y = Y0 + ly;
        long long a = 0, b = 0, c = 0;
        if (lane < 49) {
            const int by = lane / 7 - 3, bx = lane % 7 - 3;
            const int cy = ly + ORB_HALO + by, cx = lx + ORB_HALO + bx;
            const int ix = (t[cy - 1][cx + 1] + 2 * t[cy][cx + 1] + t[cy + 1][cx + 1]) - (t[cy - 1][cx - 1] + 2 * t[cy][cx - 1] + t[cy + 1][cx - 1]);
            const int iy = (t[cy + 1][cx - 1] + 2 * t[cy + 1][cx] + t[cy + 1][cx + 1]) - (t[cy - 1][cx - 1] + 2 * t[cy - 1][cx] + t[cy - 1][cx + 1]);
            a = ix * ix, b = ix * iy, c = iy * iy;
        }
        bool dirty = false;
        if (clean > 0 && lane <= 2 * clean) {
            // ORB_E >= clean: the window lies inside the image, and inside the staged rows / words
            dirty = orb_row_has_zero(zt[ly + ORB_E - clean + lane], 3, lx + 64 - clean, lx + 64 + clean);
        }
        // (products up to 1020^2: the three sums fit 32 bits.)  Row sums by four DPP steps, the four rows added on the
        // scalar unit: no LDS-crossbar shuffles (18 of them per corner made this loop the longer half of the kernel)
        a = orb_wave_sum((int)a), b = orb_wave_sum((int)b), c = orb_wave_sum((int)c);
        const bool any_dirty = __any(dirty);
        if (lane == 0) {
            const long long R = 25 * (a * c - b * b) - (a + c) * (a + c);
            uint32_t k32 = 0;
            if (R > 0 && !any_dirty) k32 = __float_as_uint((float)R);
            key[(int64_t)Y * w + X] = k32;
        }
    }
}

// K4: grid (ceil(w/32), ceil(h/32)), block 256 (4 pixels per thread)
__global__ void __launch_bounds__(256) orb_nms_kernel(const uint32_t *__restrict__ key, int h, int w, int32_t *__restrict__ tile_count,
                                                      int32_t *__restrict__ tile_pos, uint32_t *__restrict__ tile_key) {
    constexpr int R = 3, S = ORB_TILE + 2 * R;
    __shared__ uint32_t t[S][S];
    __shared__ int cnt;
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * ORB_TILE, Y0 = blockIdx.y * ORB_TILE;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) cnt = 0;
    for (int p = tid; p < S * S; p += 256) {
        const int ly = p / S, lx = p % S, yy = Y0 + ly - R, xx = X0 + lx - R;
        t[ly][lx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? key[(int64_t)yy * w + xx] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + 256 * i, ly = p >> 5, lx = p & 31;
        const uint32_t k = t[ly + R][lx + R];
        if (k == 0u) continue;
        bool keep = true;
        for (int dy = -R; dy <= R && keep; dy++)
            for (int dx = -R; dx <= R; dx++) {
                const uint32_t q = t[ly + R + dy][lx + R + dx];
                // a neighbour beats this pixel if its key is larger, or equal with a lower linear index
                const bool earlier = dy < 0 || (dy == 0 && dx < 0);
                if (q > k || (q == k && earlier)) keep = false;
            }
        if (keep) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < ORB_SLOTS) { // (always true: survivors are >= 4 px apart)
                tile_pos[tile * ORB_SLOTS + slot] = ((Y0 + ly) << 16) | (X0 + lx);
                tile_key[tile * ORB_SLOTS + slot] = k;
            }
        }
    }
    __syncthreads();
    if (tid == 0) tile_count[tile] = min(cnt, ORB_SLOTS);
}

// K5: grid n, block 64
__global__ void __launch_bounds__(64) orb_describe_kernel(const uint8_t *__restrict__ g, const uint8_t *__restrict__ bl, int h, int w,
                                                          const int32_t *__restrict__ pos, const int32_t *__restrict__ n_dev,
                                                          const int8_t *__restrict__ pattern,
                                                          uint8_t *__restrict__ des, int32_t *__restrict__ ang) {
    const int kp = blockIdx.x, lane = threadIdx.x;
    if (n_dev && kp >= n_dev[0]) return; // (the count of a device-side selection: the grid was sized for nfeatures)
    const int pk = pos[kp], X = pk & 0xFFFF, Y = pk >> 16;
    if (X < ORB_E || X >= w - ORB_E || Y < ORB_E || Y >= h - ORB_E) { // not a position K4 can have produced: touch nothing outside
        if (lane < 32) des[(int64_t)kp * 32 + lane] = 0;
        if (lane == 0) ang[kp] = -1;
        return;
    }
    // moments over the disc: lane walks rows v = lane - 15 .. (31 rows) in parallel, each row serially (<= 31 pixels)
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        const int v = lane - 15, um = c_orb_umax[v < 0 ? -v : v];
        const uint8_t *row = g + (int64_t)(Y + v) * w + X;
        int sv = 0;
        for (int u = -um; u <= um; u++) {
            const int val = row[u];
            m10 += u * val;
            sv += val;
        }
        m01 = v * sv;
    }
    m10 = orb_wave_sum(m10), m01 = orb_wave_sum(m01);
    // orientation bin: argmax over 32 integer dot products, lowest k on ties
    long long dot = LLONG_MIN;
    int best = lane & 31;
    if (lane < 32) dot = (long long)m10 * c_orb_dirs[lane][0] + (long long)m01 * c_orb_dirs[lane][1];
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) {
        const int olo = __shfl_xor((int)(unsigned)(dot & 0xFFFFFFFFll), m), ohi = __shfl_xor((int)(dot >> 32), m);
        const long long od = ((long long)ohi << 32) | (unsigned)olo;
        const int ob = __shfl_xor(best, m);
        if (od > dot || (od == dot && ob < best)) dot = od, best = ob;
    }
    best = __shfl(best, 0);
    // 256 bits: lane l tests pairs 4l .. 4l+3; np.packbits order (pair 8j is bit 7 of byte j)
    const int8_t *pt = pattern + ((int64_t)best * 256 + 4 * lane) * 4;
    uint32_t nib = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // (clamped: a caller's table can never take a read outside the patch, hence outside the image)
        const int x1 = min(max((int)pt[4 * i], -15), 15), y1 = min(max((int)pt[4 * i + 1], -15), 15);
        const int x2 = min(max((int)pt[4 * i + 2], -15), 15), y2 = min(max((int)pt[4 * i + 3], -15), 15);
        const int a = bl[(int64_t)(Y + y1) * w + X + x1], b = bl[(int64_t)(Y + y2) * w + X + x2];
        nib |= (uint32_t)(a < b) << (3 - i);
    }
    const uint32_t nxt = (uint32_t)__shfl_down((int)nib, 1);
    if ((lane & 1) == 0) des[(int64_t)kp * 32 + (lane >> 1)] = (uint8_t)((nib << 4) | nxt);
    if (lane == 0) ang[kp] = best;
}

extern "C" int isx_orb_candidates(const uint8_t *img_dev, int h, int w, int c, int64_t pitch, int fast_threshold, int clean,
                                  void *workspace_dev, int32_t *tile_count_dev, int32_t *tile_pos_dev,
                                  uint32_t *tile_key_dev, void *stream) {
    ISX_REQUIRE(h > 0 && w > 0 && h <= 32767 && w <= 32767, "image size %dx%d out of range", w, h);
    ISX_REQUIRE(c == 1 || c == 3, "channels must be 1 or 3 (got %d)", c);
    ISX_REQUIRE(pitch >= (int64_t)w * c, "pitch smaller than a row");
    ISX_REQUIRE(fast_threshold >= 1 && fast_threshold <= 254, "fast_threshold out of range");
    ISX_REQUIRE(clean >= 0 && clean <= ORB_E, "clean must be in [0, %d]", ORB_E);
    ISX_REQUIRE(img_dev && workspace_dev && tile_count_dev && tile_pos_dev && tile_key_dev, "null device pointer");
    ISX_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, "workspace must be 256-byte aligned");
    const OrbWs s = orb_ws(workspace_dev, h, w);
    hipStream_t st = (hipStream_t)stream;
    const dim3 rows((w + 255) / 256, h);
    if (c == 3)
        hipLaunchKernelGGL((orb_gray_hblur_kernel<3>), rows, dim3(256), 0, st, img_dev, h, w, pitch, s.g, s.hs, s.zb, s.zw);
    else
        hipLaunchKernelGGL((orb_gray_hblur_kernel<1>), rows, dim3(256), 0, st, img_dev, h, w, pitch, s.g, s.hs, s.zb, s.zw);
    hipLaunchKernelGGL(orb_vblur_kernel, rows, dim3(256), 0, st, s.hs, h, w, s.bl);
    hipLaunchKernelGGL(orb_fast_harris_kernel, dim3((w + ORB_TW - 1) / ORB_TW, (h + ORB_TH - 1) / ORB_TH), dim3(256), 0, st, s.g, h, w,
                       fast_threshold, clean, s.zb, s.zw, s.key);
    hipLaunchKernelGGL(orb_nms_kernel, dim3((w + ORB_TILE - 1) / ORB_TILE, (h + ORB_TILE - 1) / ORB_TILE), dim3(256), 0, st, s.key, h, w,
                       tile_count_dev, tile_pos_dev, tile_key_dev);
    return isx_launch_status("orb_candidates");
}

extern "C" int isx_orb_describe(const void *workspace_dev, int h, int w, const int32_t *pos_dev, int n,
                                const int8_t *pattern_dev, uint8_t *des_out_dev, int32_t *angle_out_dev, void *stream) {
    ISX_REQUIRE(h > 0 && w > 0 && h <= 32767 && w <= 32767, "image size %dx%d out of range", w, h);
    ISX_REQUIRE(n >= 0, "negative n");
    if (n == 0) return ISX_OK;
    ISX_REQUIRE(workspace_dev && pos_dev && pattern_dev && des_out_dev && angle_out_dev, "null device pointer");
    const OrbWs s = orb_ws((void *)workspace_dev, h, w);
    hipLaunchKernelGGL(orb_describe_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, s.g, s.bl, h, w, pos_dev,
                       (const int32_t *)nullptr, pattern_dev, des_out_dev, angle_out_dev);
    return isx_launch_status("orb_describe_kernel");
}

// ------------------------------------------------------------------------------------------------
// Selection on the device: the nfeatures best NMS survivors (largest key, ties to the lower linear index) in order.
//   S1 orb_gather_kernel  prefix sum of the tile counts (a workgroup per 64 tiles; each sums the counts before its own),
//                         the sparse tile slots packed into a dense list of 64-bit ranks (~key << 32 | y*w + x), all
//                         distinct (the list lives in the workspace's h-blur buffer, which is dead by now)
//   S2 orb_rank_kernel    16 threads per survivor: its position in the sorted order = the number of ranks below its own
//                         (the list streams through LDS in blocks of 1024); positions < nfeatures are written.  n^2
//                         compares -- 4 x 10^7 for the ~6500 survivors of a 1080p frame -- spread over every CU; measured
//                         alternatives: a one-workgroup bitonic sort in LDS 126 us (91 LDS-bound passes), one thread per
//                         survivor 103 us (26 workgroups busy)
// sel[0] = number selected (or -1: more than ORB_SELECT_CAP survivors, the caller selects on the host),
// sel[1] = survivors in total, sel[2 ..] = positions ((y << 16) | x), then nfeatures keys.
// ------------------------------------------------------------------------------------------------
#define ORB_SELECT_CAP 65536
// S1: grid = ceil(n_tiles / 64) workgroups of 256: workgroup b packs tiles [64b, 64b + 64)
__global__ void __launch_bounds__(256) orb_gather_kernel(const int32_t *__restrict__ tile_count, const int32_t *__restrict__ tile_pos,
                                                         const uint32_t *__restrict__ tile_key, int n_tiles, int w, int nfeatures,
                                                         uint64_t *__restrict__ dense, int32_t *__restrict__ sel) {
    __shared__ int wsum[4];
    __shared__ int offs[64], cnts[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t_first = blockIdx.x * 64;
    // survivors in the tiles before this workgroup's (every workgroup sums them itself: a few thousand ints from L2)
    int before = 0;
    for (int t = tid; t < t_first; t += 256) before += max(0, min(tile_count[t], ORB_SLOTS));
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
    if (lane == 0) wsum[wave] = before;
    __syncthreads();
    before = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (wave == 0) { // exclusive prefix over this workgroup's 64 tiles
        const int t = t_first + lane;
        const int c = t < n_tiles ? max(0, min(tile_count[t], ORB_SLOTS)) : 0;
        int incl = c;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int o = __shfl_up(incl, m);
            if (lane >= m) incl += o;
        }
        offs[lane] = before + incl - c;
        cnts[lane] = c;
        if (lane == 63 && t_first + 64 >= n_tiles) { // the last workgroup knows the total
            const int total = before + incl;
            sel[1] = total;
            sel[0] = total > ORB_SELECT_CAP ? -1 : min(total, nfeatures);
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int sl = tid; sl < 64 * ORB_SLOTS; sl += 256) {
        const int tl = sl >> 6, i = sl & 63;
        if (i < cnts[tl] && offs[tl] + i < ORB_SELECT_CAP) {
            const int t = t_first + tl;
            const int p = tile_pos[t * ORB_SLOTS + i];
            const uint32_t k = tile_key[t * ORB_SLOTS + i];
            dense[offs[tl] + i] = ((uint64_t)(~k) << 32) | (uint32_t)((p >> 16) * w + (p & 0xFFFF));
        }
    }
}

// S2: grid = ceil(cap / 16) workgroups of 256: 16 survivors per workgroup, 16 threads per survivor (each compares
// its survivor with every 16th entry of the list, which streams through LDS in blocks of 1024)
__global__ void __launch_bounds__(256) orb_rank_kernel(const uint64_t *__restrict__ dense, int w, int nfeatures,
                                                       int32_t *__restrict__ sel) {
    __shared__ uint64_t blk[1024];
    const int total = sel[1];
    if (sel[0] < 0 || (int)blockIdx.x * 16 >= total) return; // (block-uniform)
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
    const uint64_t mine = i < total ? dense[i] : 0ull;
    int below = 0;
    for (int b0 = 0; b0 < total; b0 += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024; k += 256) blk[k] = b0 + k < total ? dense[b0 + k] : ~0ull;
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 64; k++) below += blk[16 * k + part] < mine;
    }
    below = orb_row_sum(below); // (the 16 threads of a survivor are one DPP row)
    if (part == 0 && i < total && below < nfeatures) {
        const uint32_t idx = (uint32_t)mine;
        sel[2 + below] = (int32_t)(((idx / (uint32_t)w) << 16) | (idx % (uint32_t)w));
        sel[2 + nfeatures + below] = (int32_t)~(uint32_t)(mine >> 32);
    }
}

extern "C" int isx_orb_select_describe(const void *workspace_dev, int h, int w, const int32_t *tile_count_dev,
                                       const int32_t *tile_pos_dev, const uint32_t *tile_key_dev, int nfeatures,
                                       const int8_t *pattern_dev, int32_t *sel_out_dev, uint8_t *des_out_dev,
                                       int32_t *angle_out_dev, void *stream) {
    ISX_REQUIRE(h > 0 && w > 0 && h <= 32767 && w <= 32767, "image size %dx%d out of range", w, h);
    ISX_REQUIRE(nfeatures >= 1 && nfeatures <= ORB_SELECT_CAP, "nfeatures must be in [1, %d]", ORB_SELECT_CAP);
    ISX_REQUIRE(h >= 2 * ORB_E + 1 && w >= 2 * ORB_E + 1, "image smaller than one patch (%dx%d)", w, h);
    ISX_REQUIRE(workspace_dev && tile_count_dev && tile_pos_dev && tile_key_dev && pattern_dev && sel_out_dev && des_out_dev &&
                    angle_out_dev,
                "null device pointer");
    const int n_tiles = ((h + ORB_TILE - 1) / ORB_TILE) * ((w + ORB_TILE - 1) / ORB_TILE);
    const OrbWs s = orb_ws((void *)workspace_dev, h, w);
    hipStream_t st = (hipStream_t)stream;
    uint64_t *dense = (uint64_t *)s.hs; // h*w*2 bytes >= n_tiles*64*8 for every admitted size; dead after the v-blur
    const int64_t cap = (int64_t)n_tiles * ORB_SLOTS < ORB_SELECT_CAP ? (int64_t)n_tiles * ORB_SLOTS : ORB_SELECT_CAP;
    ISX_REQUIRE((int64_t)h * w * 2 >= cap * 8, "image too small for the selection scratch"); // (h, w >= 33 always pass)
    hipLaunchKernelGGL(orb_gather_kernel, dim3((n_tiles + 63) / 64), dim3(256), 0, st, tile_count_dev, tile_pos_dev, tile_key_dev,
                       n_tiles, w, nfeatures, dense, sel_out_dev);
    hipLaunchKernelGGL(orb_rank_kernel, dim3((unsigned)((cap + 15) / 16)), dim3(256), 0, st, (const uint64_t *)dense, w, nfeatures,
                       sel_out_dev);
    hipLaunchKernelGGL(orb_describe_kernel, dim3(nfeatures), dim3(64), 0, st, s.g, s.bl, h, w, (const int32_t *)(sel_out_dev + 2),
                       (const int32_t *)sel_out_dev, pattern_dev, des_out_dev, angle_out_dev);
    return isx_launch_status("orb_select_describe");
}
