"""TEST INFRASTRUCTURE ONLY (imported by tests/ — never by the product).

Quadric edge-collapse decimation (Garland & Heckbert 1997, in the form described in
drawingspinup_amd/csrc/mesh_decimate.hip) restated naively: no priority queue, no lazy lists — at
every step the cost of EVERY remaining edge is recomputed from the current quadrics and the
cheapest admissible one is collapsed.  With distinct costs (vertices in general position) this
takes the same collapses in the same order as the library's queue, so the two meshes must agree
vertex for vertex: a check of the queue / version / adjacency bookkeeping, not of the method.
trimesh / Open3D are absent: parity with them is unpinned."""
import numpy as np


def _plane_quadric(n, d, w):
    q = np.zeros((4, 4))
    p = np.append(n, d)
    return w * np.outer(p, p)


def _target(Q, v0, v1):
    A, b = Q[:3, :3], Q[:3, 3]
    ev = lambda x: float(x @ A @ x + 2 * b @ x + Q[3, 3])
    det, tr = np.linalg.det(A), np.trace(A)
    if tr > 0 and abs(det) > 1e-9 * tr ** 3:
        x = -np.linalg.solve(A, b)
        if np.linalg.norm(x - 0.5 * (v0 + v1)) <= 4.0 * np.linalg.norm(v1 - v0):
            return ev(x), x
    cands = [v0, v1, 0.5 * (v0 + v1)]
    costs = [ev(c) for c in cands]
    k = int(np.argmin(costs))                   # first minimum, as the library's strict '<'
    return costs[k], cands[k]


def decimate(verts, faces, target_faces, boundary_weight=1.0, keep_manifold=True):
    v = [np.array(p, np.float64) for p in verts]
    f = [list(map(int, t)) for t in faces if len(set(map(int, t))) == 3]
    Q = [np.zeros((4, 4)) for _ in v]
    cnt = {}
    for t in f:
        cr = np.cross(v[t[1]] - v[t[0]], v[t[2]] - v[t[0]])
        l = np.linalg.norm(cr)
        if l > 0:
            n = cr / l
            for i in t:
                Q[i] += _plane_quadric(n, -n @ v[t[0]], 0.5 * l)
        for k in range(3):
            e = tuple(sorted((t[k], t[(k + 1) % 3])))
            cnt.setdefault(e, []).append(t)
    if boundary_weight > 0:
        for (a, b), ts in cnt.items():
            if len(ts) != 1:
                continue
            t = ts[0]
            cr = np.cross(v[t[1]] - v[t[0]], v[t[2]] - v[t[0]])
            l = np.linalg.norm(cr)
            if not l > 0:
                continue
            en = np.cross(v[b] - v[a], cr / l)
            el = np.linalg.norm(en)
            if not el > 0:
                continue
            en /= el
            q = _plane_quadric(en, -en @ v[a], boundary_weight * 0.5 * l)
            Q[a] += q
            Q[b] += q
    ver = [0] * len(v)
    alive_v = [True] * len(v)
    orient = {}                                  # edge -> the end point that was pushed first (survivor)
    blocked = set()                              # (a, b, ver_a, ver_b) rejected in this state
    while len(f) > target_faces:
        edges = sorted({tuple(sorted((t[k], t[(k + 1) % 3]))) for t in f for k in range(3)})
        best = None
        for a, b in edges:
            if (a, b, ver[a], ver[b]) in blocked:
                continue
            c, x = _target(Q[a] + Q[b], v[a], v[b])
            if best is None or c < best[0]:
                best = (c, a, b, x)
        if best is None:
            break
        _, a, b, x = best
        # the library keeps the pushed orientation (v0, v1): v1 is removed.  Edges are pushed as
        # (low, high) initially and as (collapsed vertex, neighbour) afterwards: the end point whose
        # version changed last is v0
        v0, v1 = (a, b)
        if orient.get((a, b)) == b:
            v0, v1 = b, a
        ok = True
        shared = [t for t in f if v0 in t and v1 in t]
        for mv, other in ((v1, v0), (v0, v1)):
            for t in f:
                if mv in t and other not in t:
                    p = [v[i] for i in t]
                    before = np.cross(p[1] - p[0], p[2] - p[0])
                    p2 = [x if i == mv else v[i] for i in t]
                    after = np.cross(p2[1] - p2[0], p2[2] - p2[0])
                    if before @ after < 0:
                        ok = False
        if ok and keep_manifold:
            n0 = {i for t in f if v0 in t for i in t} - {v0, v1}
            n1 = {i for t in f if v1 in t for i in t} - {v0, v1}
            if len(n0 & n1) != len(shared):
                ok = False
            else:
                t0 = [set(t) - {v0} for t in f if v0 in t and v1 not in t]
                t1 = [set(t) - {v1} for t in f if v1 in t and v0 not in t]
                if any(s in t0 for s in t1):
                    ok = False
        if not ok or not shared:
            blocked.add((a, b, ver[a], ver[b]))
            continue
        f = [[v0 if i == v1 else i for i in t] for t in f if not (v0 in t and v1 in t)]
        v[v0] = x
        Q[v0] = Q[v0] + Q[v1]
        alive_v[v1] = False
        ver[v0] += 1
        for t in f:
            if v0 in t:
                for i in t:
                    if i != v0:
                        orient[tuple(sorted((v0, i)))] = v0
    used = sorted({i for t in f for i in t})
    return np.array([v[i] for i in used]), f, used

