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


def overlapping_crops_descriptors_device(device, seed=77, n_scenes=64, imgs_per_scene=64, kp=4096, step=448, d=128,
                                         noise_sigma=6.0):
    """The same construction as overlapping_crops_descriptors, generated ON THE DEVICE with torch's generator (BASELINE
    configs[4] is 4096 images x 4096 kp = 2 GiB of descriptors: minutes of numpy on the host, under a second here).
    The values differ from the numpy generator's (another random stream); whoever needs an image on the host
    downloads it.  Returns (des uint8 tensor (n, kp, d) on `device`, scene_of int64 ndarray (n,))."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pool_n = (imgs_per_scene - 1) * step + kp
    n = n_scenes * imgs_per_scene
    des = torch.empty((n, kp, d), dtype=torch.uint8, device=device)
    order = torch.randperm(n, generator=g, device=device).cpu().numpy()   # global image shuffle: slot -> (scene, k)
    scene_of = np.empty((n,), np.int64)
    slot_of = np.empty((n,), np.int64)
    slot_of[order] = np.arange(n)
    for s in range(n_scenes):
        x = torch.empty((pool_n, d), dtype=torch.float32, device=device).exponential_(1.0, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        x.clamp_(max=0.2)
        x /= x.norm(dim=1, keepdim=True)
        pool = torch.clamp(torch.round(x * 512.0), 0, 255)
        for k in range(imgs_per_scene):
            base = pool[k * step:k * step + kp]
            noise = torch.round(torch.randn(base.shape, generator=g, device=device) * noise_sigma)
            img = torch.clamp(base + noise, 0, 255).to(torch.uint8)
            slot = int(slot_of[s * imgs_per_scene + k])
            des[slot] = img[torch.randperm(kp, generator=g, device=device)]
            scene_of[slot] = s
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


# ---------------------------------------------------------------------------------------------
# image-level streams
# ---------------------------------------------------------------------------------------------
def procedural_world(seed=42, h=3072, w=6144, channels=3, speckle=0):
    """Seeded texture: multi-octave value noise + random rectangles, uint8 BGR, every pixel >= 12 so
    content is never "black" under the reference's gray > 10 / gray != 0 tests.  speckle = small high-contrast
    blobs per 1000 pixels (0 = none): the smooth texture alone has few FAST corners, photographs have thousands."""
    from scipy import ndimage
    rng = np.random.Generator(np.random.PCG64(seed))
    acc = np.zeros((h, w, channels), np.float32)
    amp = 1.0
    for cell in (256, 96, 32, 12, 5):
        gh, gw = h // cell + 3, w // cell + 3
        g = rng.random((gh, gw, channels), dtype=np.float32)
        up = ndimage.zoom(g, (cell, cell, 1), order=1)[:h, :w]
        acc += amp * up
        amp *= 0.62
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    img = 30 + 200 * acc
    for _ in range(int(h * w / 40000)):
        y, x = rng.integers(0, h - 8), rng.integers(0, w - 8)
        hh, ww = rng.integers(4, 60, 2)
        img[y:y + hh, x:x + ww] = rng.integers(12, 256, channels)
    if speckle:
        rs = np.random.Generator(np.random.PCG64(seed + 1000))  # (its own stream: speckle=0 worlds are unchanged)
        n = int(h * w * speckle / 1000)
        ys, xs = rs.integers(0, h - 5, n), rs.integers(0, w - 5, n)
        sz = rs.integers(2, 5, (n, 2))
        delta = rs.choice([-1.0, 1.0], n) * rs.integers(40, 100, n)
        for y, x, (hh, ww), dv in zip(ys, xs, sz, delta):
            img[y:y + hh, x:x + ww] += dv
    img = np.clip(img, 12, 255).astype(np.uint8)
    return img if channels > 1 else img[..., 0]


def crop_path(world, n_frames, frame_hw=(480, 640), step_frac=0.15, max_rot_deg=1.0, seed=43, start=None):
    """Frames along a mostly-horizontal path: `step_frac` of the frame width per frame plus a small
    vertical drift and a rotation in [-max_rot, max_rot] degrees.  Returns (frames list of uint8
    arrays, poses list of (cx, cy, theta_rad))."""
    from scipy import ndimage
    rng = np.random.Generator(np.random.PCG64(seed))
    fh, fw = frame_hw
    H, W = world.shape[:2]
    cx = fw * 0.75 if start is None else start[0]
    cy = H / 2 if start is None else start[1]
    frames, poses = [], []
    ys, xs = np.mgrid[0:fh, 0:fw].astype(np.float32)
    xs -= (fw - 1) / 2
    ys -= (fh - 1) / 2
    for k in range(n_frames):
        th = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg)) if max_rot_deg > 0 else 0.0
        c, s = np.cos(th), np.sin(th)
        wx = cx + c * xs - s * ys
        wy = cy + s * xs + c * ys
        if max_rot_deg == 0 and float(cx).is_integer() and float(cy).is_integer() and fw % 2 == 0 and fh % 2 == 0:
            pass
        chans = []
        wv = world.reshape(H, W, -1)
        for ch in range(wv.shape[2]):
            chans.append(ndimage.map_coordinates(wv[..., ch], [wy, wx], order=1, mode="nearest"))
        f = np.stack(chans, -1).astype(np.uint8)
        frames.append(np.ascontiguousarray(f if world.ndim == 3 else f[..., 0]))
        poses.append((cx, cy, th))
        cx += step_frac * fw
        cy += rng.uniform(-0.03, 0.03) * fh
        cy = min(max(cy, fh * 0.8), H - fh * 0.8)
        if cx > W - fw * 0.75:
            break
    return frames, poses


# ---------------------------------------------------------------------------------------------
# the same kind of stream generated on the device (bench.py's long 1080p stream: a 4k x 10k world and 130 rotated
# crops take minutes with scipy on the host, about a second here).  Another random stream than the numpy functions
# above: whoever compares two paths feeds both the same frames.
# ---------------------------------------------------------------------------------------------
def procedural_world_torch(device, seed=42, h=3072, w=6144, speckle=0):
    """procedural_world on `device`: multi-octave value noise + random rectangles (+ 3x3 high-contrast blobs, `speckle`
    per 1000 pixels), uint8 (h, w, 3) tensor, every pixel >= 12."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    acc = torch.zeros((1, 3, h, w), dtype=torch.float32, device=device)
    amp = 1.0
    for cell in (256, 96, 32, 12, 5):
        grid = torch.rand((1, 3, h // cell + 3, w // cell + 3), generator=g, device=device)
        acc += amp * F.interpolate(grid, scale_factor=cell, mode="bilinear", align_corners=False)[..., :h, :w]
        amp *= 0.62
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    img = 30 + 200 * acc[0]
    n = int(h * w / 40000)
    ys = torch.randint(0, h - 8, (n,), generator=g, device=device).tolist()
    xs = torch.randint(0, w - 8, (n,), generator=g, device=device).tolist()
    sz = torch.randint(4, 60, (n, 2), generator=g, device=device).tolist()
    col = torch.randint(12, 256, (n, 3), generator=g, device=device).to(torch.float32)
    for k in range(n):
        img[:, ys[k]:ys[k] + sz[k][0], xs[k]:xs[k] + sz[k][1]] = col[k][:, None, None]
    if speckle:
        m = int(h * w * speckle / 1000)
        by = torch.randint(1, h - 2, (m,), generator=g, device=device)
        bx = torch.randint(1, w - 2, (m,), generator=g, device=device)
        dv = (torch.randint(0, 2, (m,), generator=g, device=device) * 2 - 1) * torch.randint(40, 100, (m,), generator=g, device=device)
        delta = torch.zeros((1, 1, h, w), dtype=torch.float32, device=device)
        delta[0, 0].index_put_((by, bx), dv.to(torch.float32), accumulate=True)
        # 3x3 box sum (avg_pool: a plain kernel -- conv2d would send MIOpen on a multi-second search for an algorithm)
        img = img + 9.0 * F.avg_pool2d(delta, 3, stride=1, padding=1, count_include_pad=True)[0]
    return img.clamp_(12, 255).to(torch.uint8).permute(1, 2, 0).contiguous()


def serpentine_poses(n_frames, frame_hw=(1080, 1920), step_frac=0.15, per_row=28, down_steps=3, max_rot_deg=1.0, seed=43):
    """Camera poses (cx, cy, theta) of a mapping flight: `per_row` frames to the right (step_frac of the frame width
    each), `down_steps` frames downwards (step_frac of the frame HEIGHT... of the width, so that rows overlap by about
    half a frame), then back to the left, and so on -- the mosaic grows in both axes.  Returns (poses, (h, w) of the
    world that contains every frame)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fh, fw = frame_hw
    sx, sy = step_frac * fw, step_frac * fw * 0.6
    margin = 0.75 * max(fw, fh)
    cx, cy, direction = margin, margin, 1
    poses, k = [], 0
    while len(poses) < n_frames:
        for _ in range(per_row):
            poses.append((cx, cy, np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg))))
            cx += direction * sx
            cy += rng.uniform(-0.01, 0.01) * fh
        cx -= direction * sx
        for _ in range(down_steps):
            cy += sy
            poses.append((cx, cy, np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg))))
        cy += sy
        direction = -direction
    poses = poses[:n_frames]
    W = int(max(p[0] for p in poses) + margin) + 1
    H = int(max(p[1] for p in poses) + margin) + 1
    return poses, (H, W)


def crop_poses_torch(world_t, poses, frame_hw=(1080, 1920)):
    """Frames (uint8 numpy arrays, host) cut from a device world along `poses`: bilinear samples of the rotated frame
    rectangle, as crop_path does with scipy."""
    import torch
    import torch.nn.functional as F
    fh, fw = frame_hw
    H, W = world_t.shape[:2]
    dev = world_t.device
    src = world_t.permute(2, 0, 1)[None].to(torch.float32)
    ys, xs = torch.meshgrid(torch.arange(fh, device=dev, dtype=torch.float32) - (fh - 1) / 2,
                            torch.arange(fw, device=dev, dtype=torch.float32) - (fw - 1) / 2, indexing="ij")
    frames = []
    for cx, cy, th in poses:
        c, s = float(np.cos(th)), float(np.sin(th))
        gx = (cx + c * xs - s * ys) * (2.0 / (W - 1)) - 1.0
        gy = (cy + s * xs + c * ys) * (2.0 / (H - 1)) - 1.0
        f = F.grid_sample(src, torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="border", align_corners=True)
        frames.append(f[0].round_().clamp_(12, 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy())
    return frames
