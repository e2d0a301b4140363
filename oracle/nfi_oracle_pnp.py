"""TEST INFRASTRUCTURE ONLY (never imported by nerf_from_image_amd): an independent float64 solver for the pose problem of
lib/pose_estimation.py:30-131, used to check the HIP kernel (csrc/nfi_pnp.inc).

PARITY UNPINNED: the reference delegates to OpenCV (cv2.solvePnPGeneric: SQPNP / EPNP / iterative), which is absent
from this image and from /root/reference, so neither this file nor the kernel can be pinned to the reference's numbers.
What IS pinned by construction is the problem statement - screen grid (col / W - 0.5, row / H - 0.5), intrinsics
diag(f, f, 1), RMS reprojection error sqrt(sum |r|^2 / 2N) (OpenCV's definition), best focal proposal with t_z > 0, the
flip @ [R | t] output and the dummy pose - which follows pose_estimation.py line by line.  The minimiser here is
scipy.optimize.least_squares (trust-region, rotation-vector parametrisation) started from an SVD-based DLT, i.e. a
different algorithm from the kernel's, so agreement between the two is evidence about both.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def screen_grid(height, width):
    ii, jj = np.meshgrid(np.arange(width) / width, np.arange(height) / height, indexing='xy')    # pose_estimation.py:32-36
    return (np.stack((ii, jj), axis=-1) - 0.5).reshape(-1, 2)


def project(R, t, f, P):
    q = P @ R.T + t
    return f * q[:, :2] / q[:, 2:3], q[:, 2]


def dlt(P, xy, f):
    u = xy / f
    c3, c2 = P.mean(0), u.mean(0)
    s3 = np.sqrt(3) / np.linalg.norm(P - c3, axis=1).mean()
    s2 = np.sqrt(2) / np.linalg.norm(u - c2, axis=1).mean()
    Pn, un = (P - c3) * s3, (u - c2) * s2
    ph = np.concatenate([Pn, np.ones((len(P), 1))], axis=1)
    z = np.zeros_like(ph)
    A = np.concatenate([np.concatenate([ph, z, -un[:, :1] * ph], axis=1), np.concatenate([z, ph, -un[:, 1:] * ph], axis=1)])
    Mn = np.linalg.svd(A, full_matrices=False)[2][-1].reshape(3, 4)
    T2i = np.array([[1 / s2, 0, c2[0]], [0, 1 / s2, c2[1]], [0, 0, 1]])
    T3 = np.eye(4) * s3
    T3[3, 3] = 1
    T3[:3, 3] = -s3 * c3
    M = T2i @ Mn @ T3
    if np.linalg.det(M[:, :3]) < 0:
        M = -M
    U, S, Vt = np.linalg.svd(M[:, :3])
    return U @ Vt, M[:, 3] / S.mean()


def dlt_is_degenerate(P, xy, f):
    """Coplanar (or otherwise degenerate) points: the constraint matrix has a null space of more than one dimension
    (second-smallest singular value ~ 0) and the linear start is an arbitrary member of it."""
    u = xy / f
    c3, c2 = P.mean(0), u.mean(0)
    s3 = np.sqrt(3) / np.linalg.norm(P - c3, axis=1).mean()
    s2 = np.sqrt(2) / np.linalg.norm(u - c2, axis=1).mean()
    Pn, un = (P - c3) * s3, (u - c2) * s2
    ph = np.concatenate([Pn, np.ones((len(P), 1))], axis=1)
    z = np.zeros_like(ph)
    A = np.concatenate([np.concatenate([ph, z, -un[:, :1] * ph], axis=1), np.concatenate([z, ph, -un[:, 1:] * ph], axis=1)])
    sv = np.linalg.svd(A, compute_uv=False)
    return not (sv[-2] ** 2 > 1e-8 * sv[0] ** 2)


def axis_aligned_rotations():
    """The 24 rotations that map coordinate axes onto coordinate axes."""
    out = []
    for perm, parity in (((0, 1, 2), 1), ((1, 2, 0), 1), ((2, 0, 1), 1), ((0, 2, 1), -1), ((1, 0, 2), -1), ((2, 1, 0), -1)):
        for s0 in (1, -1):
            for s1 in (1, -1):
                R = np.zeros((3, 3))
                R[0, perm[0]], R[1, perm[1]], R[2, perm[2]] = s0, s1, parity * s0 * s1
                out.append(R)
    return out


def solve_one(P, xy, f, refine=True):
    """Returns (R, t, rmse) or None.  >= 6 points in general position: the linear start + refinement.  4 or 5 points
    (pose_estimation.py:58 solves from 4 on), coplanar points, or a refinement that ends behind the camera (OpenCV falls
    back to EPNP there): refinement from the 24 axis-aligned rotations at the weak-perspective depth, best valid result."""
    if len(P) < 4:
        return None

    def res(p):
        pr, z = project(Rotation.from_rotvec(p[:3]).as_matrix(), p[3:], f, P)
        r = pr - xy
        r[z <= 1e-9] = 1.0
        return r.ravel()

    def polish(R, t, do=True):
        if do and len(P) >= 3:
            sol = least_squares(res, np.concatenate([Rotation.from_matrix(R).as_rotvec(), t]), method='lm' if 2 * len(P) >= 6 else 'trf',
                                xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=400)
            R, t = Rotation.from_rotvec(sol.x[:3]).as_matrix(), sol.x[3:]
        pr, z = project(R, t, f, P)
        r = pr - xy
        r[z <= 1e-9] = 1.0
        return R, t, float(np.sqrt((r ** 2).sum() / (2 * len(P))))

    if len(P) >= 6 and not dlt_is_degenerate(P, xy, f):
        R, t = dlt(P, xy, f)
        if np.isfinite(R).all() and np.isfinite(t).all():
            first = polish(R, t, refine)
            if first[1][2] > 0 and np.isfinite(first[2]):
                return first
    u = xy / f
    c3, c2 = P.mean(0), u.mean(0)
    d0 = np.linalg.norm(P - c3, axis=1).mean() / np.sqrt(3) * np.sqrt(2) / max(np.linalg.norm(u - c2, axis=1).mean(), 1e-300)
    best = None
    for R0 in axis_aligned_rotations():
        t0 = d0 * np.array([c2[0], c2[1], 1.0]) - R0 @ c3
        cand = polish(R0, t0)
        if cand[1][2] > 0 and np.isfinite(cand[2]) and (best is None or cand[2] < best[2]):
            best = cand
    return best


def compute_pose_pnp(coords, masks, focal_proposals, refine=True):
    """numpy in, numpy out, with pose_estimation.py's conventions: (world2cam [B,4,4], focal [B], errors [B])."""
    bs, height, width, _ = coords.shape
    grid = screen_grid(height, width)
    coords = coords.astype(np.float64)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    mats, focals, errors = [], [], []
    for b in range(bs):
        idx, = np.where(masks[b].flatten())
        P, xy = coords[b].reshape(-1, 3)[idx], grid[idx]
        best = None
        for f in focal_proposals:
            s = solve_one(P, xy, float(f), refine)
            if s is not None and s[1][2] > 0 and (best is None or s[2] < best[2]):
                best = (s[0], s[1], s[2], float(f))
        m = np.eye(4)
        if best is None:
            m[:3, 3] = (0.0, 0.0, -10.0)                 # pose_estimation.py:110-117
            focal, err = 1.0, 10.0
        else:
            m[:3, :3], m[:3, 3], err, focal = best
        mats.append(flip @ m)
        focals.append(focal)
        errors.append(err)
    return np.stack(mats), np.array(focals), np.array(errors)


def synthetic_correspondences(bs, res, seed, noise=0.0, focal=1.2, distance=2.5):
    """Exact correspondences for tests: a camera at distance ~`distance` looking at an ellipsoid whose surface points (in object
    coordinates) are what a perfect coordinate regressor would output per pixel.  Returns coords [B,res,res,3] float32,
    masks [B,res,res] bool, ground-truth R [B,3,3], t [B,3] (OpenCV convention: P_cam = R P + t, z forward)."""
    rng = np.random.default_rng(seed)
    grid = screen_grid(res, res)
    coords = np.zeros((bs, res * res, 3), np.float32)
    masks = np.zeros((bs, res * res), bool)
    Rs, ts = [], []
    radii = np.array([0.9, 0.6, 0.75])
    for b in range(bs):
        R = Rotation.from_rotvec(rng.normal(size=3) * 1.2).as_matrix()
        tz = distance * (1.0 + rng.normal() * 0.08)
        t = np.array([rng.normal() * 0.07 * tz / focal, rng.normal() * 0.07 * tz / focal, tz])    # the object stays on screen
        # rays in camera space through every pixel, intersected with the ellipsoid |(R^T (q - t)) / radii| = 1
        d_cam = np.concatenate([grid / focal, np.ones((len(grid), 1))], axis=1)
        o_obj, d_obj = -R.T @ t, d_cam @ R            # origin / direction in object coordinates
        oo, dd = o_obj / radii, d_obj / radii
        a, bq, c = (dd * dd).sum(1), 2 * (dd @ oo), oo @ oo - 1
        disc = bq * bq - 4 * a * c
        hit = disc > 0
        s = (-bq - np.sqrt(np.where(hit, disc, 0))) / (2 * a)
        P = o_obj + d_obj * s[:, None]
        coords[b] = np.where(hit[:, None], P, 0) + (rng.normal(size=P.shape) * noise if noise else 0)
        masks[b] = hit
        Rs.append(R)
        ts.append(t)
    return coords.reshape(bs, res, res, 3), masks.reshape(bs, res, res), np.stack(Rs), np.stack(ts)
