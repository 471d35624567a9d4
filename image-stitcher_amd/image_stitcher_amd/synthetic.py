"""Seeded synthetic inputs (cv2-free) shaped like the reference's workloads -- SURVEY.md section 8d.

Used by bench.py and the tests; generated identically here and on the GPU box (numpy PCG64).
"""
import numpy as np


def sift_like_pool(rng, n_features, d=128):
    """Integer-valued SIFT-like descriptors: Exp(1) -> L2 normalise -> clip 0.2 -> renormalise ->
    x512 -> round -> clip [0,255]   (mimics cv2.SIFT's post-processing; ||d|| ~ 512)."""
    x = rng.standard_exponential((n_features, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    np.minimum(x, 0.2, out=x)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.clip(np.rint(x * 512.0), 0, 255).astype(np.uint8)


def overlapping_crops_descriptors(seed=1234, n_scenes=8, imgs_per_scene=64, kp=4096, step=448, d=128,
                                  noise_sigma=6.0, shuffle_images=True):
    """`n_scenes*imgs_per_scene` images x `kp` descriptors.  Image k of a scene takes `kp` consecutive
    pool features starting at k*step, each perturbed by round(N(0, sigma)) and re-clipped; rows are
    shuffled per image and images are shuffled globally.  Returns (des uint8 (n,kp,d), scene_of (n,))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pool_n = (imgs_per_scene - 1) * step + kp
    n = n_scenes * imgs_per_scene
    des = np.empty((n, kp, d), np.uint8)
    scene_of = np.empty((n,), np.int64)
    for s in range(n_scenes):
        pool = sift_like_pool(rng, pool_n, d).astype(np.int16)
        for k in range(imgs_per_scene):
            base = pool[k * step:k * step + kp]
            noise = np.rint(rng.normal(0.0, noise_sigma, base.shape)).astype(np.int16)
            img = np.clip(base + noise, 0, 255).astype(np.uint8)
            img = img[rng.permutation(kp)]
            des[s * imgs_per_scene + k] = img
            scene_of[s * imgs_per_scene + k] = s
    if shuffle_images:
        p = rng.permutation(n)
        des, scene_of = des[p], scene_of[p]
    return des, scene_of


def orb_like_descriptors(seed=1235, n_images=8, kp=500, d=32, flips=8, overlap=0.5):
    """Uniform random byte descriptors; consecutive images share `overlap` of their rows up to
    `flips` random bit flips (planted neighbours)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    prev = None
    for _ in range(n_images):
        cur = rng.integers(0, 256, (kp, d), dtype=np.uint8)
        if prev is not None:
            m = int(kp * overlap)
            src = prev[rng.permutation(kp)[:m]].copy()
            for r in range(m):
                for _f in range(flips):
                    b = rng.integers(0, d * 8)
                    src[r, b >> 3] ^= np.uint8(1 << (b & 7))
            cur[:m] = src
            cur = cur[rng.permutation(kp)]
        out.append(cur)
        prev = cur
    return out
