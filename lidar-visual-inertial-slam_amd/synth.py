"""Seeded synthetic inputs for the hot path (there is no dataset and no network).

Lidar: a MID360-like scan of an indoor hall (walls, floor, ceiling, 12 square pillars,
6 partition walls).  The reference's feature extraction (featureExtraction.cpp:87-148)
treats consecutive points of one Livox line as spatial neighbours, so the scan pattern
here is a continuous rosette per line (azimuth sweeps n_rev turns per frame while the
elevation oscillates), not a random/low-discrepancy pattern — with the latter every
point would look like an edge and the reference algorithm would be meaningless.

Images: band-limited noise texture plus rectangles, and a homography warp of it.
"""
import numpy as np

from . import _abi as A

ROOM = np.array([[-30.0, 30.0], [-22.5, 22.5], [0.0, 8.0]])      # hall 60 x 45 x 8 m
SENSOR_Z = 1.2
LOOP_RADIUS = 9.5                                                 # ~60 m loop


def scene_boxes():
    """axis-aligned boxes (xmin,xmax,ymin,ymax,zmin,zmax): 12 pillars + 6 partition walls"""
    boxes = []
    for px in (-18.0, -6.0, 6.0, 18.0):
        for py in (-13.0, 0.0, 13.0):
            boxes.append([px - 0.3, px + 0.3, py - 0.3, py + 0.3, 0.0, 8.0])
    walls = [(-27.0, -19.0, -17.1, -16.9), (19.0, 27.0, 16.9, 17.1), (-24.1, -23.9, 4.0, 12.0),
             (23.9, 24.1, -12.0, -4.0), (-4.0, 4.0, 18.9, 19.1), (-4.0, 4.0, -19.1, -18.9)]
    for (x0, x1, y0, y1) in walls:
        boxes.append([x0, x1, y0, y1, 0.0, 3.5])
    return np.array(boxes, np.float64)


def rot_zyx(roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll) — same convention as pcl::getTransformation"""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, sy * sr + cy * sp * cr],
                     [sy * cp, cy * cr + sy * sp * sr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def scan_directions(n, n_rev=10.0, f_el=37.3):
    """unit ray directions in the sensor frame, MID360-like: 4 lines, 360 deg, elevation -7..52 deg"""
    i = np.arange(n, dtype=np.float64)
    line = (np.arange(n) % 4).astype(np.uint8)
    tau = i / n
    az = 2 * np.pi * (n_rev * tau) + line * (np.pi / 2)
    mid, amp = np.deg2rad(22.5), np.deg2rad(29.5)
    el = mid + amp * np.sin(2 * np.pi * f_el * tau + line * 1.7)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    return d, line


def raycast(origin, dirs_world, boxes=None):
    """distance along each ray to the first surface of the hall"""
    if boxes is None:
        boxes = scene_boxes()
    o = np.asarray(origin, np.float64)
    d = np.where(np.abs(dirs_world) < 1e-12, 1e-12, dirs_world)
    inv = 1.0 / d
    # inside the room: exit distance through the bounding planes
    t_far = np.where(d > 0, (ROOM[:, 1] - o) * inv, (ROOM[:, 0] - o) * inv)
    t = t_far.min(axis=1)
    for b in boxes:
        lo = (b[0::2] - o) * inv
        hi = (b[1::2] - o) * inv
        tmin = np.minimum(lo, hi).max(axis=1)
        tmax = np.maximum(lo, hi).min(axis=1)
        hit = (tmax >= tmin) & (tmin > 0)
        t = np.where(hit & (tmin < t), tmin, t)
    return t


def raycast_torch(origin, dirs_world, device, boxes=None):
    """same ray casting on a torch device (bench set-up only: 250 keyframe scans in seconds)"""
    import torch
    if boxes is None:
        boxes = scene_boxes()
    o = torch.as_tensor(np.asarray(origin, np.float64), device=device)
    d = torch.as_tensor(dirs_world, device=device)
    d = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    inv = 1.0 / d
    room = torch.as_tensor(ROOM, device=device)
    t = torch.where(d > 0, (room[:, 1] - o) * inv, (room[:, 0] - o) * inv).min(dim=1).values
    for b in boxes:
        bt = torch.as_tensor(b, device=device)
        lo = (bt[0::2] - o) * inv
        hi = (bt[1::2] - o) * inv
        tmin = torch.minimum(lo, hi).max(dim=1).values
        tmax = torch.maximum(lo, hi).min(dim=1).values
        hit = (tmax >= tmin) & (tmin > 0) & (tmin < t)
        t = torch.where(hit, tmin, t)
    return t.cpu().numpy()


def make_scan(n_raw, pose, seed, noise=0.02, torch_device=None):
    """one Livox CustomMsg worth of points (LIVOX_DTYPE, n_raw entries) seen from `pose`
    = (roll, pitch, yaw, x, y, z) of the sensor in the map frame"""
    rng = np.random.default_rng(seed)
    dirs, line = scan_directions(n_raw)
    R = rot_zyx(pose[0], pose[1], pose[2])
    if torch_device is not None:
        t = raycast_torch(np.array(pose[3:6], np.float64), dirs @ R.T, torch_device)
    else:
        t = raycast(np.array(pose[3:6], np.float64), dirs @ R.T)
    r = t + rng.normal(0.0, noise, n_raw)
    p = dirs * r[:, None]
    out = np.zeros(n_raw, A.LIVOX_DTYPE)
    out["x"], out["y"], out["z"] = p[:, 0].astype(np.float32), p[:, 1].astype(np.float32), p[:, 2].astype(np.float32)
    out["reflectivity"] = rng.integers(0, 256, n_raw, dtype=np.uint8)
    out["line"] = line
    out["offset_time"] = (np.arange(n_raw, dtype=np.float64) * (1e8 / n_raw)).astype(np.uint32)
    return out


def loop_pose(alpha, roll=0.0, pitch=0.0):
    return np.array([roll, pitch, alpha + np.pi / 2, LOOP_RADIUS * np.cos(alpha), LOOP_RADIUS * np.sin(alpha), SENSOR_Z], np.float64)


def transform_points(pts, pose):
    """float64 rigid transform of a PT_DTYPE cloud (generator side, not the path's a-5)"""
    x = A.pts_xyzi(pts).astype(np.float64)
    R = rot_zyx(pose[0], pose[1], pose[2])
    w = x[:, :3] @ R.T + np.asarray(pose[3:6])
    out = np.concatenate([w, x[:, 3:4]], axis=1).astype(np.float32)
    return out.view(A.PT_DTYPE).reshape(-1)


def make_map(extractor, n_keyframes, n_raw, seed=4711, pose_sigma=(0.01, np.deg2rad(0.1)), torch_device=None, target_surf=None, keyframes_out=None):
    """frozen local map: laserCloud{Corner,Surf}FromMap before downsampling.

    `extractor` is any LidarHotpath (its organize+extract stages turn each synthetic
    keyframe scan into corner/surf clouds, as the reference's own pipeline would); the
    clouds are moved to the map frame with the ground-truth pose plus a small error."""
    rng = np.random.default_rng(seed)
    corners, surfs = [], []
    n_surf = 0
    for k in range(n_keyframes):
        if target_surf is not None and n_surf >= target_surf:
            break
        alpha = 2 * np.pi * k / n_keyframes
        pose = loop_pose(alpha, roll=rng.normal(0, 0.01), pitch=rng.normal(0, 0.01))
        scan = make_scan(n_raw, pose, seed * 1000 + k, torch_device=torch_device)
        extractor.scan_upload(scan)
        extractor.scan_organize()
        extractor.scan_extract()
        c, s = extractor.get_features()
        noisy = pose.copy()
        noisy[:3] += rng.normal(0, pose_sigma[1], 3)
        noisy[3:] += rng.normal(0, pose_sigma[0], 3)
        if keyframes_out is not None:
            keyframes_out.append((c.copy(), s.copy(), noisy.astype(np.float32)))      # sensor-frame clouds + the pose they were mapped with
        corners.append(transform_points(c, noisy))
        surfs.append(transform_points(s, noisy))
        n_surf += len(s)
    mc, ms = np.concatenate(corners), np.concatenate(surfs)
    if target_surf is not None:
        ms = ms[:target_surf]
    return mc, ms


def perturbed_guess(pose, scan_id, rot_deg=1.5, trans=0.15):
    rng = np.random.default_rng(777 + scan_id)
    g = np.array(pose, np.float64)
    g[:3] += np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3))
    g[3:] += rng.uniform(-trans, trans, 3)
    return g.astype(np.float32)


# ----------------------------------------------------------------------------- images
def _bilinear_up(a, h, w):
    sh, sw = a.shape
    ys = np.linspace(0, sh - 1, h)
    xs = np.linspace(0, sw - 1, w)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, sh - 1); x1 = np.minimum(x0 + 1, sw - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    return (a[np.ix_(y0, x0)] * (1 - fy) * (1 - fx) + a[np.ix_(y0, x1)] * (1 - fy) * fx
            + a[np.ix_(y1, x0)] * fy * (1 - fx) + a[np.ix_(y1, x1)] * fy * fx)


def make_texture(w, h, seed=4242):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w))
    for o in range(6):
        sh, sw = max(2, h >> (6 - o)), max(2, w >> (6 - o))
        img += _bilinear_up(rng.uniform(-1, 1, (sh, sw)), h, w) / (1.6 ** o)
    for _ in range(40):
        x0 = rng.integers(0, w - 20); y0 = rng.integers(0, h - 20)
        ww = rng.integers(10, max(11, w // 8)); hh = rng.integers(10, max(11, h // 8))
        img[y0:y0 + hh, x0:x0 + ww] += rng.choice([-1.0, 1.0]) * rng.uniform(0.4, 1.0)
    lo, hi = np.percentile(img, [1, 99])
    img = np.clip((img - lo) / (hi - lo), 0, 1)
    return (img * 255 + 0.5).astype(np.uint8)


def warp_homography(img, Hm):
    """frame k = frame 0 seen through homography Hm (maps frame-0 pixels to frame-k pixels)"""
    h, w = img.shape
    Hi = np.linalg.inv(Hm)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    sx = np.clip(sx, 0, w - 1.001); sy = np.clip(sy, 0, h - 1.001)
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int)
    fx = sx - x0; fy = sy - y0
    f = img.astype(np.float64)
    out = f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy) + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy
    return (out + 0.5).astype(np.uint8)


def small_motion_homography(w, h, seed, max_px=6.0):
    rng = np.random.default_rng(seed)
    ang = rng.uniform(-0.004, 0.004)
    s = 1.0 + rng.uniform(-0.003, 0.003)
    tx, ty = rng.uniform(-max_px * 0.6, max_px * 0.6, 2)
    cx, cy = w / 2, h / 2
    c, sn = np.cos(ang) * s, np.sin(ang) * s
    Hm = np.array([[c, -sn, cx - c * cx + sn * cy + tx], [sn, c, cy - sn * cx - c * cy + ty], [0, 0, 1.0]])
    Hm[2, 0] = rng.uniform(-1e-6, 1e-6); Hm[2, 1] = rng.uniform(-1e-6, 1e-6)
    return Hm


def apply_homography(Hm, xy):
    xy = np.asarray(xy, np.float64)
    den = Hm[2, 0] * xy[:, 0] + Hm[2, 1] * xy[:, 1] + Hm[2, 2]
    return np.stack([(Hm[0, 0] * xy[:, 0] + Hm[0, 1] * xy[:, 1] + Hm[0, 2]) / den,
                     (Hm[1, 0] * xy[:, 0] + Hm[1, 1] * xy[:, 1] + Hm[1, 2]) / den], axis=1)
