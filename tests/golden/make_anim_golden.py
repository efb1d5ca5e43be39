#!/usr/bin/env python3
"""Generate tests/golden/anim_interpolate.npz from the REFERENCE's own compiled AnimatedTransform
(oracle/_ref/ref_anim: constructor with Decompose, and Interpolate, util/transform.cpp:375-470,
1062-1081).  Build container only.  The file holds seeded inputs (start / end matrices, times) and
the reference's raw outputs: the decomposed members T, R, S the C ABI carries, the start / end inverse
matrices, and Interpolate(time)'s m / mInv."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_anim")
OUT = os.path.join(ROOT, "tests", "golden")


def rot(axis, ang):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    c, s = np.cos(ang), np.sin(ang)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) * c + s * K + (1 - c) * np.outer(a, a)


def affine(rng, rotate=True, scale=True):
    m = np.eye(4)
    R = rot(rng.normal(size=3), rng.uniform(-3, 3)) if rotate else np.eye(3)
    S = np.diag(rng.uniform(0.3, 3, 3)) if scale else np.eye(3)
    if scale and rng.random() < 0.3:  # a shear-bearing stretch
        S = S + 0.2 * rng.normal(size=(3, 3))
        S = (S + S.T) / 2 + 2 * np.eye(3)
    m[:3, :3] = R @ S
    m[:3, 3] = rng.uniform(-20, 20, 3)
    return m


def cases(n, seed):
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 35), np.float32)
    for i in range(n):
        kind = i % 6
        a = affine(rng)
        if kind == 0:      # general
            b = affine(rng)
        elif kind == 1:    # translation only: no rotation between the two
            b = a.copy()
            b[:3, 3] += rng.uniform(-5, 5, 3)
        elif kind == 2:    # small rotation (Dot(R0, R1) > 0.9995: SinXOverX near 1)
            b = a.copy()
            b[:3, :3] = rot(rng.normal(size=3), rng.uniform(-0.01, 0.01)) @ a[:3, :3]
        elif kind == 3:    # rotation by more than 90 degrees about the other way (R[1] flipped)
            b = a.copy()
            b[:3, :3] = rot(rng.normal(size=3), rng.uniform(2.5, 3.1)) @ a[:3, :3]
        elif kind == 4:    # not animated
            b = a.copy()
        else:              # rigid motion
            a, b = affine(rng, scale=False), affine(rng, scale=False)
        t0 = rng.uniform(-1, 1)
        t1 = t0 + rng.uniform(0.1, 2)
        time = rng.choice([t0 - 0.1, t0, t1, t1 + 0.3, rng.uniform(t0, t1), rng.uniform(t0, t1), rng.uniform(t0, t1)])
        rec[i, :16], rec[i, 16:32], rec[i, 32:] = a.ravel(), b.ravel(), [t0, t1, time]
    return rec


def run_ref(rec):
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "i.bin"), os.path.join(td, "o.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(np.ascontiguousarray(rec, np.float32).tobytes())
        subprocess.run([REF, fi, fo], check=True)
        return np.fromfile(fo, np.uint32).reshape(len(rec), 116)


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_anim missing: run `make -C oracle ref` in the build container")
    rec = cases(3000, 20241010)
    out = run_ref(rec)
    np.savez_compressed(os.path.join(OUT, "anim_interpolate.npz"), inputs=rec, outputs=out)
    print(f"anim: {len(rec)} cases, {int(out[:, 0].view(np.float32).sum())} animated")
