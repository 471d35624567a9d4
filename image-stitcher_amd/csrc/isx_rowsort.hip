// Rows of every image sorted by norm -- the row order isx_knn2_l2_count_pairs likes (round 3).
//
// ArbitraryStitcher.get_collections (reference stitcher.py:292-296) uses only len(good_matches) of an image pair, which does
// not depend on the order of either image's descriptor rows.  The pair kernel (isx_knn2.hip, knn2_pair_filter_kernel) bounds
// the row direction per CLASS of 128 consecutive query rows by the class's smallest norm, so it settles more rows without an
// exact recount when the rows of a class have similar norms: rows sorted by norm inside each image, rotated by 64 so that
// the 64 smallest norms follow the 64 largest (one wide class per image -- the last -- instead of two).
//
// One workgroup of 1024 per image: (norm << 32 | row) keys into LDS, bitonic sort (ties by row index: a deterministic
// order), then the workgroup moves its image's rows -- descriptor bytes and the three per-row words -- to their sorted
// places, 16 bytes per lane.  HBM-bound byte work: (d_pad + 12) bytes read and written per row (73 MB + 73 MB for 512 images x
// 4096 rows x 128 bytes), the sort itself is 78 LDS stages for 4096 rows.
#include "isx_common.h"

template <int NSORT>
__global__ void __launch_bounds__(1024)
rows_sort_gather_kernel(const int8_t *__restrict__ desc_in, const uint32_t *__restrict__ cbase_in, const int32_t *__restrict__ nrm_in,
                        const int32_t *__restrict__ cin_in, int dp, const int32_t *__restrict__ starts, int uniform_count,
                        int8_t *__restrict__ desc_out, uint32_t *__restrict__ cbase_out, int32_t *__restrict__ nrm_out,
                        int32_t *__restrict__ cin_out) {
    extern __shared__ __align__(16) unsigned long long key[];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int64_t start = starts ? (int64_t)starts[img] : (int64_t)img * uniform_count;
    const int n = starts ? starts[img + 1] - starts[img] : uniform_count;
    if (n <= 0 || n > NSORT) return; // (uniform over the workgroup; n > NSORT: the caller's max_count was wrong -- touch nothing)
    for (int i = tid; i < NSORT; i += 1024)
        key[i] = i < n ? ((unsigned long long)(uint32_t)nrm_in[start + i] << 32) | (uint32_t)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= NSORT; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NSORT / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned long long a = key[i], b = key[i + j];
                if ((a > b) == ((i & k) == 0)) key[i] = b, key[i + j] = a;
            }
            __syncthreads();
        }
    const int rot = n > 256 ? 64 : 0;
    const int lpr = dp >> 4; // lanes per row: 16 bytes each (dp = 32, 64 or 128)
    const int piece = tid % lpr;
    for (int j = tid / lpr; j < n; j += 1024 / lpr) {
        int p = j + rot;
        p = p >= n ? p - n : p;
        const int64_t src = start + (int64_t)(uint32_t)key[p], dst = start + j;
        ((uint4 *)(desc_out + dst * dp))[piece] = ((const uint4 *)(desc_in + src * dp))[piece];
        if (piece == 0) {
            cbase_out[dst] = cbase_in[src];
            nrm_out[dst] = nrm_in[src];
            cin_out[dst] = cin_in[src];
        }
    }
}

extern "C" int isx_sort_rows_by_norm(const int8_t *desc_in_dev, const uint32_t *cbase_in_dev, const int32_t *nrm_in_dev,
                                     const int32_t *cin_in_dev, int d_pad, const int32_t *starts_dev, int n_images,
                                     int uniform_count, int max_count, int8_t *desc_out_dev, uint32_t *cbase_out_dev,
                                     int32_t *nrm_out_dev, int32_t *cin_out_dev, void *stream) {
    ISX_REQUIRE(d_pad == 32 || d_pad == 64 || d_pad == 128, "isx_sort_rows_by_norm: d_pad must be 32, 64 or 128 (got %d)", d_pad);
    ISX_REQUIRE(n_images >= 0 && max_count >= 0 && uniform_count >= 0, "negative size");
    ISX_REQUIRE(starts_dev || uniform_count > 0 || n_images == 0, "either starts_dev (n_images + 1 row offsets) or uniform_count");
    ISX_REQUIRE(!starts_dev || uniform_count == 0, "starts_dev and uniform_count exclude each other");
    ISX_REQUIRE(starts_dev || uniform_count <= max_count, "max_count is smaller than uniform_count");
    if (n_images == 0 || max_count == 0) return ISX_OK;
    ISX_REQUIRE(desc_in_dev && cbase_in_dev && nrm_in_dev && cin_in_dev && desc_out_dev && cbase_out_dev && nrm_out_dev && cin_out_dev,
                "null device pointer");
    ISX_REQUIRE(desc_in_dev != desc_out_dev && cbase_in_dev != cbase_out_dev && nrm_in_dev != nrm_out_dev && cin_in_dev != cin_out_dev,
                "isx_sort_rows_by_norm does not work in place");
    if (max_count > ISX_SORT_MAX_ROWS)
        return isx_fail(ISX_E_UNSUPPORTED, "isx_sort_rows_by_norm: an image of %d rows (limit %d: the keys of one image live in LDS)",
                        max_count, ISX_SORT_MAX_ROWS);
    hipStream_t st = (hipStream_t)stream;
    typedef void (*kern_t)(const int8_t *, const uint32_t *, const int32_t *, const int32_t *, int, const int32_t *, int, int8_t *,
                           uint32_t *, int32_t *, int32_t *);
    const int nsort = max_count <= 1024 ? 1024 : max_count <= 4096 ? 4096 : 16384;
    const kern_t fn = nsort == 1024 ? rows_sort_gather_kernel<1024> : nsort == 4096 ? rows_sort_gather_kernel<4096> : rows_sort_gather_kernel<16384>;
    const int lds = nsort * 8;
    ISX_HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)n_images), dim3(1024), lds, st, desc_in_dev, cbase_in_dev, nrm_in_dev, cin_in_dev, d_pad,
                       starts_dev, uniform_count, desc_out_dev, cbase_out_dev, nrm_out_dev, cin_out_dev);
    return isx_launch_status("rows_sort_gather_kernel");
}
