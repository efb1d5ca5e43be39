"""Synthetic ray batches generated ON THE DEVICE (torch tensors as buffers): the generators of scene.py —
pinhole camera rays with per-sample jitter, one cosine-weighted diffuse bounce per hit, shadow rays towards
area-light quads or a light box — as a few vectorised passes over device arrays, so that `bench.py` (one rank per
GPU) does not spend tens of seconds of host time per rank before its first step.  Same recipes as scene.py
(float64 arithmetic, float32 records), their own seeded random streams; rays are 32-B nnbvh_ray records
(o.xyz, tMax, d.xyz, time) in a float32 [n, 8] tensor, hits 32-B nnbvh_hit records viewed as int32 / float32."""
import numpy as np
import torch

from . import scene


def _gen(device, *seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(np.random.SeedSequence([int(s) for s in seed]).generate_state(1, np.uint64)[0] >> 1))
    return g


class DeviceScene:
    """Vertex / index arrays and camera pixel lists resident on the device."""

    def __init__(self, verts, tris, device):
        self.device = device
        self.verts = torch.from_numpy(np.ascontiguousarray(verts, np.float32)).to(device)
        self.tris = torch.from_numpy(np.ascontiguousarray(tris, np.int64)).to(device)
        self.scale = float(np.abs(verts).max())

    def camera_rays(self, cam, px, py, seed, sample):
        """px, py: float64 device tensors of the pixels to cover (any order); one jittered ray per pixel.  The jitter
        of a pixel depends on (seed, sample) and on its position in the list."""
        eye, look, up, fov, xres, yres = scene.CAMERAS[cam] if isinstance(cam, str) else cam
        eye, look, up = (np.asarray(a, np.float64) for a in (eye, look, up))
        w = look - eye
        w /= np.linalg.norm(w)
        right = np.cross(w, up)
        right /= np.linalg.norm(right)
        upv = np.cross(right, w)
        half = np.tan(np.radians(fov) / 2)
        s = min(xres, yres)
        j = torch.rand((len(px), 2), generator=_gen(self.device, 11, seed, sample), device=self.device, dtype=torch.float64)
        sx = ((px + j[:, 0]) - xres / 2) / (s / 2) * half
        sy = (yres / 2 - (py + j[:, 1])) / (s / 2) * half
        t = lambda v: torch.tensor(v, dtype=torch.float64, device=self.device)  # noqa: E731
        d = t(w)[None] + sx[:, None] * t(right)[None] + sy[:, None] * t(upv)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rays = torch.zeros((len(px), 8), dtype=torch.float32, device=self.device)
        rays[:, 0:3] = t(eye).to(torch.float32)
        rays[:, 3] = float("inf")
        rays[:, 4:7] = d.to(torch.float32)
        return rays

    def _hit_points(self, rays, hits):
        """(indices of the hit rays, hit point, unit geometric normal facing the ray origin), float64"""
        prim = hits.view(torch.int32).view(-1, 8)[:, 0]
        idx = torch.nonzero(prim >= 0).squeeze(1)
        h = hits.view(torch.float32).view(-1, 8)[idx]
        tri = self.tris[prim[idx].long()]
        p0, p1, p2 = (self.verts[tri[:, k]] for k in range(3))
        p = (h[:, 2:3] * p0 + h[:, 3:4] * p1 + h[:, 4:5] * p2).double()
        n = torch.cross(p1 - p0, p2 - p0, dim=1).double()
        ln = n.norm(dim=1, keepdim=True)
        n = n / torch.where(ln > 0, ln, torch.ones_like(ln))
        flip = (n * rays[idx, 4:7].double()).sum(1) > 0
        n = torch.where(flip[:, None], -n, n)
        return idx, p, n

    def bounce_rays(self, rays, hits, seed, eps_scale=1e-4):
        """one cosine-weighted diffuse bounce per hit ray; returns (rays [m, 8], index of the parent ray [m])"""
        idx, p, n = self._hit_points(rays, hits)
        u = torch.rand((len(idx), 2), generator=_gen(self.device, 12, *np.atleast_1d(seed)), device=self.device, dtype=torch.float64)
        rr, phi = u[:, 0].sqrt(), 2 * np.pi * u[:, 1]
        lx, ly, lz = rr * phi.cos(), rr * phi.sin(), (1 - u[:, 0]).clamp(min=0).sqrt()
        a = torch.zeros_like(n)
        steep = n[:, 0].abs() > 0.9
        a[:, 0] = (~steep).double()
        a[:, 1] = steep.double()
        t = torch.cross(n, a, dim=1)
        t = t / t.norm(dim=1, keepdim=True)
        b = torch.cross(n, t, dim=1)
        d = lx[:, None] * t + ly[:, None] * b + lz[:, None] * n
        out = torch.zeros((len(idx), 8), dtype=torch.float32, device=self.device)
        out[:, 0:3] = (p + n * (eps_scale * self.scale)).to(torch.float32)
        out[:, 3] = float("inf")
        out[:, 4:7] = d.to(torch.float32)
        return out, idx

    def shadow_rays(self, rays, hits, seed, quads=None, box=None, eps_scale=1e-4):
        """shadow rays from the hit points to uniformly sampled points on area-light quads (corners p0 p1 p2 p3) or in an
        axis-aligned box (lo, hi): un-normalised d = pLight - p, tMax = 1 - ShadowEpsilon (cpu/integrators.h:52-54)"""
        idx, p, n = self._hit_points(rays, hits)
        g = _gen(self.device, 13, *np.atleast_1d(seed))
        if quads is not None:
            q = torch.tensor(np.asarray(quads, np.float64), device=self.device)
            which = torch.randint(0, len(q), (len(idx),), generator=g, device=self.device)
            uv = torch.rand((len(idx), 2), generator=g, device=self.device, dtype=torch.float64)
            qq = q[which]
            pl = qq[:, 0] + uv[:, 0:1] * (qq[:, 1] - qq[:, 0]) + uv[:, 1:2] * (qq[:, 3] - qq[:, 0])
        else:
            lo, hi = (torch.tensor(np.asarray(v, np.float64), device=self.device) for v in box)
            pl = lo + torch.rand((len(idx), 3), generator=g, device=self.device, dtype=torch.float64) * (hi - lo)
        o = (p + n * (eps_scale * self.scale)).to(torch.float32)
        out = torch.zeros((len(idx), 8), dtype=torch.float32, device=self.device)
        out[:, 0:3] = o
        out[:, 3] = float(np.float32(1 - 1e-4))
        out[:, 4:7] = pl.to(torch.float32) - o
        return out, idx


def as_records(t):
    """float32 [n, 8] device tensor -> numpy RAY_DTYPE records (host copy)"""
    from ._lib import RAY_DTYPE
    return t.cpu().numpy().view(RAY_DTYPE).reshape(-1)
