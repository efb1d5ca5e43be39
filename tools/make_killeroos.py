#!/usr/bin/env python3
"""Build data/killeroos.npz (BASELINE.json configs 0/1) from the reference's scene text files
(build container only): the two loop-subdivided killeroos + the two ground quads of
scenes/killeroos/killeroo-simple.pbrt (the light sphere, a quadric, is omitted — SURVEY.md §6
probe E did the same).

Scene ingestion, not hot path: a float32 restatement of
  * LoopSubdivide          /root/reference/src/pbrt/util/loopsubdiv.cpp:135-432
    (even/odd vertex rules, boundary rules, limit-surface push; one-ring walk order kept so the
    float32 sums associate identically)
  * the CTM of killeroo-simple.pbrt:48-61 built like BasicSceneBuilder does (scene.cpp:121-124,
    589-597: ctm = ctm * T), Scale / Rotate / Translate matrices (util/transform.h:220-247,
    util/transform.cpp:21-49), FMA-chain matrix product (util/math.h:1499-1509) and the point
    transform applied by the TriangleMesh constructor (util/transform.h:310-319, util/mesh.cpp:36-39).
Check (tests/test_oracle.py): the SAH tree over this geometry must have the node count the
reference's own build produced (129 771, depth 24 — SURVEY.md §6).
"""
import math
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nn_bvh_amd import scene  # noqa: E402

REF = os.environ.get("NNBVH_REFERENCE", "/root/reference")
F = np.float32


def NEXT(i):
    return (i + 1) % 3


def PREV(i):
    return (i + 2) % 3


class SDVertex:
    __slots__ = ("p", "startFace", "child", "regular", "boundary")

    def __init__(self, p=None):
        self.p = np.zeros(3, F) if p is None else np.asarray(p, F)
        self.startFace = None
        self.child = None
        self.regular = False
        self.boundary = False

    def valence(self):
        f = self.startFace
        if not self.boundary:
            nf = 1
            f = f.nextFace(self)
            while f is not self.startFace:
                nf += 1
                f = f.nextFace(self)
            return nf
        nf = 1
        f = f.nextFace(self)
        while f is not None:
            nf += 1
            f = f.nextFace(self)
        f = self.startFace.prevFace(self)
        while f is not None:
            nf += 1
            f = f.prevFace(self)
        return nf + 1

    def oneRing(self):
        ring = []
        if not self.boundary:
            face = self.startFace
            while True:
                ring.append(face.nextVert(self).p)
                face = face.nextFace(self)
                if face is self.startFace:
                    break
        else:
            face = self.startFace
            f2 = face.nextFace(self)
            while f2 is not None:
                face = f2
                f2 = face.nextFace(self)
            ring.append(face.nextVert(self).p)
            while True:
                ring.append(face.prevVert(self).p)
                face = face.prevFace(self)
                if face is None:
                    break
        return ring


class SDFace:
    __slots__ = ("v", "f", "children")

    def __init__(self):
        self.v = [None] * 3
        self.f = [None] * 3
        self.children = [None] * 4

    def vnum(self, vert):
        for i in range(3):
            if self.v[i] is vert:
                return i
        raise RuntimeError("vnum")

    def nextFace(self, vert):
        return self.f[self.vnum(vert)]

    def prevFace(self, vert):
        return self.f[PREV(self.vnum(vert))]

    def nextVert(self, vert):
        return self.v[NEXT(self.vnum(vert))]

    def prevVert(self, vert):
        return self.v[PREV(self.vnum(vert))]

    def otherVert(self, v0, v1):
        for i in range(3):
            if self.v[i] is not v0 and self.v[i] is not v1:
                return self.v[i]
        raise RuntimeError("otherVert")


def beta(valence):
    return F(3) / F(16) if valence == 3 else F(3) / (F(8) * F(valence))


def loop_gamma(valence):
    return F(1) / (F(valence) + F(3) / (F(8) * beta(valence)))


def weight_one_ring(vert, b):
    valence = vert.valence()
    ring = vert.oneRing()
    p = (F(1) - F(valence) * b) * vert.p
    for i in range(valence):
        p = p + b * ring[i]
    return p.astype(F)


def weight_boundary(vert, b):
    valence = vert.valence()
    ring = vert.oneRing()
    p = (F(1) - F(2) * b) * vert.p
    p = p + b * ring[0]
    p = p + b * ring[valence - 1]
    return p.astype(F)


def loop_subdivide(n_levels, indices, P):
    order = {}  # creation order stands in for the pointer order SDEdge sorts by
    vertices = []
    for p in P:
        v = SDVertex(p)
        order[id(v)] = len(order)
        vertices.append(v)
    faces = [SDFace() for _ in range(len(indices) // 3)]
    for i, f in enumerate(faces):
        for j in range(3):
            v = vertices[indices[3 * i + j]]
            f.v[j] = v
            v.startFace = f

    def edge_key(a, b):
        ia, ib = order[id(a)], order[id(b)]
        return (ia, ib) if ia < ib else (ib, ia)

    edges = {}
    for f in faces:
        for e in range(3):
            k = edge_key(f.v[e], f.v[NEXT(e)])
            if k not in edges:
                edges[k] = (f, e)
            else:
                f0, e0 = edges.pop(k)
                f0.f[e0] = f
                f.f[e] = f0
    for v in vertices:
        f = v.startFace
        while True:
            f = f.nextFace(v)
            if f is None or f is v.startFace:
                break
        v.boundary = f is None
        val = v.valence()
        v.regular = (not v.boundary and val == 6) or (v.boundary and val == 4)

    f_cur, v_cur = faces, vertices
    for _ in range(n_levels):
        new_faces, new_vertices = [], []
        for vertex in v_cur:
            c = SDVertex()
            order[id(c)] = len(order)
            c.regular, c.boundary = vertex.regular, vertex.boundary
            vertex.child = c
            new_vertices.append(c)
        for face in f_cur:
            for k in range(4):
                face.children[k] = SDFace()
                new_faces.append(face.children[k])
        for vertex in v_cur:
            if not vertex.boundary:
                if vertex.regular:
                    vertex.child.p = weight_one_ring(vertex, F(1) / F(16))
                else:
                    vertex.child.p = weight_one_ring(vertex, beta(vertex.valence()))
            else:
                vertex.child.p = weight_boundary(vertex, F(1) / F(8))
        edge_verts = {}
        for face in f_cur:
            for k in range(3):
                a, b = face.v[k], face.v[NEXT(k)]
                key = edge_key(a, b)
                if key in edge_verts:
                    continue
                e0, e1 = (a, b) if order[id(a)] < order[id(b)] else (b, a)
                vert = SDVertex()
                order[id(vert)] = len(order)
                new_vertices.append(vert)
                vert.regular = True
                vert.boundary = face.f[k] is None
                vert.startFace = face.children[3]
                if vert.boundary:
                    p = F(0.5) * e0.p
                    p = p + F(0.5) * e1.p
                else:
                    p = (F(3) / F(8)) * e0.p
                    p = p + (F(3) / F(8)) * e1.p
                    p = p + (F(1) / F(8)) * face.otherVert(e0, e1).p
                    p = p + (F(1) / F(8)) * face.f[k].otherVert(e0, e1).p
                vert.p = p.astype(F)
                edge_verts[key] = vert
        for vertex in v_cur:
            vertex.child.startFace = vertex.startFace.children[vertex.startFace.vnum(vertex)]
        for face in f_cur:
            for j in range(3):
                face.children[3].f[j] = face.children[NEXT(j)]
                face.children[j].f[NEXT(j)] = face.children[3]
                f2 = face.f[j]
                face.children[j].f[j] = f2.children[f2.vnum(face.v[j])] if f2 is not None else None
                f2 = face.f[PREV(j)]
                face.children[j].f[PREV(j)] = (f2.children[f2.vnum(face.v[j])]
                                               if f2 is not None else None)
        for face in f_cur:
            for j in range(3):
                face.children[j].v[j] = face.v[j].child
                vert = edge_verts[edge_key(face.v[j], face.v[NEXT(j)])]
                face.children[j].v[NEXT(j)] = vert
                face.children[NEXT(j)].v[j] = vert
                face.children[3].v[j] = vert
        f_cur, v_cur = new_faces, new_vertices

    p_limit = []
    for v in v_cur:
        if v.boundary:
            p_limit.append(weight_boundary(v, F(1) / F(5)))
        else:
            p_limit.append(weight_one_ring(v, loop_gamma(v.valence())))
    index_of = {id(v): i for i, v in enumerate(v_cur)}
    tris = np.array([[index_of[id(f.v[j])] for j in range(3)] for f in f_cur], np.int32)
    return np.array(p_limit, F), tris


# ---- Transform restatement (float32) ---------------------------------------------------
def fma32(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))


def matmul(m1, m2):
    r = np.zeros((4, 4), F)
    for i in range(4):
        for j in range(4):
            acc = F(0)
            for k in range(4):
                acc = fma32(m1[i, k], m2[k, j], acc)
            r[i, j] = acc
    return r


def translate(x, y, z):
    m = np.eye(4, dtype=F)
    m[:3, 3] = [x, y, z]
    return m


def scale(x, y, z):
    return np.diag(np.array([x, y, z, 1], F))


def rotate(theta, axis):
    rad = F(F(math.pi) / F(180)) * F(theta)
    s, c = F(math.sin(float(rad))), F(math.cos(float(rad)))
    a = np.asarray(axis, F)
    a = a / F(math.sqrt(float(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])))
    m = np.eye(4, dtype=F)
    one = F(1)
    m[0, 0] = a[0] * a[0] + (one - a[0] * a[0]) * c
    m[0, 1] = a[0] * a[1] * (one - c) - a[2] * s
    m[0, 2] = a[0] * a[2] * (one - c) + a[1] * s
    m[1, 0] = a[0] * a[1] * (one - c) + a[2] * s
    m[1, 1] = a[1] * a[1] + (one - a[1] * a[1]) * c
    m[1, 2] = a[1] * a[2] * (one - c) - a[0] * s
    m[2, 0] = a[0] * a[2] * (one - c) - a[1] * s
    m[2, 1] = a[1] * a[2] * (one - c) + a[0] * s
    m[2, 2] = a[2] * a[2] + (one - a[2] * a[2]) * c
    return m


def apply(m, P):
    P = np.asarray(P, F)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    out = np.empty_like(P)
    for r in range(3):
        out[:, r] = m[r, 0] * x + m[r, 1] * y + m[r, 2] * z + m[r, 3]
    return out  # w == 1 for these affine matrices


def floats(text, key):
    m = re.search(r'"%s"\s*\[([^\]]*)\]' % re.escape(key), text)
    return m.group(1).split()


def main():
    base = os.path.join(REF, "scenes", "killeroos")
    if not os.path.isdir(base):
        sys.exit(f"{base} not present: killeroos can only be made in the build container")
    text = open(os.path.join(base, "geometry", "killeroo.pbrt")).read()
    levels = int(floats(text, "integer levels")[0])
    P = np.array(floats(text, "point3 P"), F).reshape(-1, 3)
    idx = np.array(floats(text, "integer indices"), np.int64)
    p_limit, tris = loop_subdivide(levels, idx, P)
    print(f"control mesh {len(idx) // 3} faces / {len(P)} verts -> level {levels}: "
          f"{len(tris)} tris / {len(p_limit)} verts")
    # killeroo-simple.pbrt:48-61
    ctm = scale(0.5, 0.5, 0.5)
    ctm = matmul(ctm, rotate(-60, (0, 0, 1)))
    ctm = matmul(ctm, translate(100, 200, -140))
    k1 = apply(ctm, p_limit)
    ctm2 = matmul(ctm, translate(-200, 0, 0))
    k2 = apply(ctm2, p_limit)
    # :32-46 ground quads under Translate 0 0 -140
    g = translate(0, 0, -140)
    q1 = apply(g, [[-1000, -1000, 0], [1000, -1000, 0], [1000, 1000, 0], [-1000, 1000, 0]])
    q2 = apply(g, [[-400, -1000, -1000], [-400, 1000, -1000], [-400, 1000, 1000], [-400, -1000, 1000]])
    quad = np.array([[0, 1, 2], [2, 3, 0]], np.int32)
    # creation order in the file: the two ground meshes, then killeroo 1, killeroo 2
    verts = np.concatenate([q1, q2, k1, k2])
    t = np.concatenate([quad, quad + 4, tris + 8, tris + 8 + len(k1)])
    scene.save_blob("killeroos", verts, t)
    print(f"killeroos: {len(verts)} verts, {len(t)} tris -> data/killeroos.npz")


if __name__ == "__main__":
    main()
