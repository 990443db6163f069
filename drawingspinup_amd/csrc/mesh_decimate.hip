// Quadric edge-collapse decimation of a triangle mesh — `remesh()` of the reference's export
// (2_charactor_reconstructor/instant_nsr/utils/mesh_utils.py:10-22, called from
// models/geometry.py:63-64 with face_count = 50000 on the fine marching-cubes mesh):
// trimesh's `simplify_quadratic_decimation`, which hands the mesh to Open3D's
// `TriangleMesh::simplify_quadric_decimation(target_number_of_triangles)`.
//
// HOST code, as in the reference: every collapse is chosen by a priority-queue pop and changes the
// costs around it; at ~1 M input faces it costs about a second of one core.  It lives in
// libdsu_hip.so so that the export has one native implementation behind the C ABI.
//
// Restated from the published method (Garland & Heckbert, "Surface Simplification Using Quadric
// Error Metrics", SIGGRAPH 97) in the form Open3D documents for that call — neither trimesh nor
// Open3D is installed in this image: PARITY UNPINNED (the collapse order of equal-cost edges and
// the conditioning threshold below are this file's; the tests check the contract instead: face
// count, topology, distance to the input surface).  What is kept:
//   * per-vertex quadrics = sum over incident triangles of (area x plane quadric);
//   * boundary edges add (boundary_weight x area x quadric of the plane through the edge,
//     perpendicular to its triangle) to both end points;
//   * an edge's target = minimiser of the summed quadric when its 3x3 block is well conditioned,
//     otherwise the best of {v0, v1, midpoint};
//   * a collapse that flips the normal of a surviving triangle is rejected;
//   * collapses are taken in order of increasing cost until the face target is met.
// Added: the link condition (the end points' common neighbours are exactly the apexes of the
// triangles on the edge), so that a closed 2-manifold stays one — Open3D does not test it; the
// host steps behind this one (cotangent Laplacian of the thinning deformation, z-ray caster)
// want a manifold.  `flags & 1` switches it off.
#include "common.h"

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <queue>
#include <unordered_map>
#include <vector>

namespace {

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(V3 a) { return sqrt(dot(a, a)); }

// symmetric 4x4 [A b; b^T c]: error(v) = v^T A v + 2 b^T v + c
struct Quadric {
  double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0, c = 0;
  void add_plane(V3 n, double d, double w) {
    a00 += w * n.x * n.x; a01 += w * n.x * n.y; a02 += w * n.x * n.z;
    a11 += w * n.y * n.y; a12 += w * n.y * n.z; a22 += w * n.z * n.z;
    b0 += w * n.x * d; b1 += w * n.y * d; b2 += w * n.z * d;
    c += w * d * d;
  }
  void add(const Quadric& q) {
    a00 += q.a00; a01 += q.a01; a02 += q.a02; a11 += q.a11; a12 += q.a12; a22 += q.a22;
    b0 += q.b0; b1 += q.b1; b2 += q.b2; c += q.c;
  }
  double eval(V3 v) const {
    return v.x * (a00 * v.x + 2 * a01 * v.y + 2 * a02 * v.z) + v.y * (a11 * v.y + 2 * a12 * v.z) +
           a22 * v.z * v.z + 2 * (b0 * v.x + b1 * v.y + b2 * v.z) + c;
  }
  // minimiser -A^-1 b when A is well conditioned relative to its own scale
  bool minimum(V3& out) const {
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    const double tr = a00 + a11 + a22;
    if (!(tr > 0.0) || !(fabs(det) > 1e-9 * tr * tr * tr)) return false;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double inv = -1.0 / det;
    out.x = inv * (c00 * b0 + c01 * b1 + c02 * b2);
    out.y = inv * (c01 * b0 + c11 * b1 + c12 * b2);
    out.z = inv * (c02 * b0 + c12 * b1 + c22 * b2);
    return true;
  }
};

struct Entry {
  double cost;
  int32_t v0, v1;
  uint32_t ver0, ver1;
  uint64_t order;
};
struct EntryCmp {
  bool operator()(const Entry& a, const Entry& b) const {
    return a.cost > b.cost || (a.cost == b.cost && a.order > b.order);
  }
};

struct Mesh {
  std::vector<V3> v;
  std::vector<int32_t> f;                  // 3 per triangle
  std::vector<uint8_t> f_alive, v_alive;
  std::vector<std::vector<int32_t>> vt;    // vertex -> triangles (lazily cleaned)
  std::vector<Quadric> q;
  std::vector<uint32_t> ver;
  std::vector<uint32_t> stamp;
  uint32_t cur_stamp = 0;

  bool has(int32_t t, int32_t vi) const {
    return f[3 * t] == vi || f[3 * t + 1] == vi || f[3 * t + 2] == vi;
  }
  V3 tri_cross(int32_t t) const {
    return cross(v[f[3 * t + 1]] - v[f[3 * t]], v[f[3 * t + 2]] - v[f[3 * t]]);
  }
  // drop dead / foreign / repeated triangles from a vertex's list
  void clean(int32_t vi) {
    auto& l = vt[vi];
    size_t k = 0;
    for (size_t i = 0; i < l.size(); ++i)
      if (f_alive[l[i]] && has(l[i], vi)) l[k++] = l[i];
    l.resize(k);
    std::sort(l.begin(), l.end());
    l.erase(std::unique(l.begin(), l.end()), l.end());
  }
};

void edge_target(const Mesh& m, int32_t v0, int32_t v1, double& cost, V3& vbar) {
  Quadric q = m.q[v0];
  q.add(m.q[v1]);
  if (q.minimum(vbar)) {
    // a minimiser far outside the edge's neighbourhood means the conditioning test was too kind
    const double len = norm(m.v[v1] - m.v[v0]);
    const V3 mid = (m.v[v0] + m.v[v1]) * 0.5;
    if (norm(vbar - mid) <= 4.0 * len) {
      cost = q.eval(vbar);
      return;
    }
  }
  const V3 cand[3] = {m.v[v0], m.v[v1], (m.v[v0] + m.v[v1]) * 0.5};
  cost = q.eval(cand[0]);
  vbar = cand[0];
  for (int i = 1; i < 3; ++i) {
    const double c = q.eval(cand[i]);
    if (c < cost) { cost = c; vbar = cand[i]; }
  }
}

}  // namespace

static int decimate_impl(const double* verts, int64_t n_verts, const int32_t* faces,
                         int64_t n_faces, int64_t target_faces, double boundary_weight,
                         int32_t flags, const double* init_quadrics, double* out_verts,
                         int64_t* out_n_verts, int32_t* out_faces, int64_t* out_n_faces) {
  if (n_verts < 0 || n_faces < 0 || target_faces < 0 || (n_verts && !verts) || (n_faces && !faces) ||
      !out_n_verts || !out_n_faces || (n_verts && !out_verts) || (n_faces && !out_faces) ||
      n_verts > INT32_MAX || n_faces > INT32_MAX || !(boundary_weight >= 0.0))
    return DSU_EINVAL;
  for (int64_t i = 0; i < 3 * n_faces; ++i)
    if (faces[i] < 0 || faces[i] >= n_verts) return DSU_EINVAL;
  const bool link_test = !(flags & 1);
  Mesh m;
  m.v.resize(n_verts);
  for (int64_t i = 0; i < n_verts; ++i) m.v[i] = {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
  m.f.assign(faces, faces + 3 * n_faces);
  m.f_alive.assign(n_faces, 1);
  m.v_alive.assign(n_verts, 1);
  m.vt.resize(n_verts);
  m.q.resize(n_verts);
  m.ver.assign(n_verts, 0);
  m.stamp.assign(n_verts, 0);
  int64_t alive = 0;
  // triangles with a repeated vertex carry no surface: dropped up front
  for (int32_t t = 0; t < n_faces; ++t) {
    const int32_t a = m.f[3 * t], b = m.f[3 * t + 1], c = m.f[3 * t + 2];
    if (a == b || b == c || a == c) { m.f_alive[t] = 0; continue; }
    ++alive;
    m.vt[a].push_back(t); m.vt[b].push_back(t); m.vt[c].push_back(t);
  }
  // ---- vertex quadrics
  std::unordered_map<uint64_t, int32_t> edge_tri;      // edge -> one of its triangles
  edge_tri.reserve((size_t)alive * 2);
  std::unordered_map<uint64_t, int32_t> edge_count;
  edge_count.reserve((size_t)alive * 2);
  auto ekey = [](int32_t a, int32_t b) {
    if (a > b) std::swap(a, b);
    return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
  };
  for (int32_t t = 0; t < n_faces; ++t) {
    if (!m.f_alive[t]) continue;
    const V3 cr = m.tri_cross(t);
    const double l = norm(cr);
    if (l > 0.0 && !init_quadrics) {
      const V3 n = cr * (1.0 / l);
      const double area = 0.5 * l, d = -dot(n, m.v[m.f[3 * t]]);
      for (int k = 0; k < 3; ++k) m.q[m.f[3 * t + k]].add_plane(n, d, area);
    }
    for (int k = 0; k < 3; ++k) {
      const uint64_t key = ekey(m.f[3 * t + k], m.f[3 * t + (k + 1) % 3]);
      ++edge_count[key];
      edge_tri[key] = t;
    }
  }
  if (init_quadrics) {
    // quadrics accumulated by earlier collapses (dsu_mesh_decimate_parallel): (n_verts,10) in this
    // file's member order a00 a01 a02 a11 a12 a22 b0 b1 b2 c
    for (int64_t i = 0; i < n_verts; ++i) {
      const double* p = init_quadrics + 10 * i;
      Quadric& q = m.q[i];
      q.a00 = p[0]; q.a01 = p[1]; q.a02 = p[2]; q.a11 = p[3]; q.a12 = p[4]; q.a22 = p[5];
      q.b0 = p[6]; q.b1 = p[7]; q.b2 = p[8]; q.c = p[9];
    }
  }
  if (boundary_weight > 0.0 && !init_quadrics) {
    for (const auto& kv : edge_count) {
      if (kv.second != 1) continue;
      const int32_t t = edge_tri[kv.first];
      const int32_t a = (int32_t)(kv.first >> 32), b = (int32_t)(kv.first & 0xffffffffu);
      const V3 cr = m.tri_cross(t);
      const double l = norm(cr);
      if (!(l > 0.0)) continue;
      V3 en = cross(m.v[b] - m.v[a], cr * (1.0 / l));
      const double el = norm(en);
      if (!(el > 0.0)) continue;
      en = en * (1.0 / el);
      const double w = boundary_weight * 0.5 * l, d = -dot(en, m.v[a]);
      m.q[a].add_plane(en, d, w);
      m.q[b].add_plane(en, d, w);
    }
  }
  // ---- queue of all edges
  std::priority_queue<Entry, std::vector<Entry>, EntryCmp> pq;
  uint64_t order = 0;
  auto push_edge = [&](int32_t a, int32_t b) {
    double cost; V3 vb;
    edge_target(m, a, b, cost, vb);
    pq.push(Entry{cost, a, b, m.ver[a], m.ver[b], order++});
  };
  {
    std::vector<uint64_t> keys;
    keys.reserve(edge_count.size());
    for (const auto& kv : edge_count) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());               // deterministic push order
    for (uint64_t k : keys) push_edge((int32_t)(k >> 32), (int32_t)(k & 0xffffffffu));
  }
  edge_count.clear();
  edge_tri.clear();

  while (alive > target_faces && !pq.empty()) {
    const Entry e = pq.top();
    pq.pop();
    const int32_t v0 = e.v0, v1 = e.v1;
    if (!m.v_alive[v0] || !m.v_alive[v1] || m.ver[v0] != e.ver0 || m.ver[v1] != e.ver1) continue;
    m.clean(v0);
    m.clean(v1);
    double cost; V3 vbar;
    edge_target(m, v0, v1, cost, vbar);
    // triangles on the edge, flip test on the others
    int shared = 0;
    bool bad = false;
    for (int pass = 0; pass < 2 && !bad; ++pass) {
      const int32_t mv = pass ? v0 : v1, other = pass ? v1 : v0;
      for (int32_t t : m.vt[mv]) {
        if (m.has(t, other)) { if (!pass) ++shared; continue; }
        const V3 before = m.tri_cross(t);
        V3 p[3];
        for (int k = 0; k < 3; ++k) p[k] = m.f[3 * t + k] == mv ? vbar : m.v[m.f[3 * t + k]];
        const V3 after = cross(p[1] - p[0], p[2] - p[0]);
        if (dot(before, after) < 0.0) { bad = true; break; }
      }
    }
    if (bad || shared == 0) continue;
    if (link_test) {
      // common neighbours of v0 and v1 must be exactly the apexes of the shared triangles
      ++m.cur_stamp;
      for (int32_t t : m.vt[v0])
        for (int k = 0; k < 3; ++k) m.stamp[m.f[3 * t + k]] = m.cur_stamp;
      m.stamp[v0] = m.stamp[v1] = 0;
      int common = 0;
      for (int32_t t : m.vt[v1])
        for (int k = 0; k < 3; ++k) {
          const int32_t u = m.f[3 * t + k];
          if (m.stamp[u] == m.cur_stamp) { ++common; m.stamp[u] = 0; }
        }
      if (common != shared) continue;
      // ... and no edge in both links: a triangle (v1, x, y) next to a triangle (v0, x, y) would
      // land on it (the last step of a tetrahedron folding flat)
      bool twin = false;
      for (int32_t t1 : m.vt[v1]) {
        if (m.has(t1, v0)) continue;
        int32_t xy[2], k2 = 0;
        for (int k = 0; k < 3; ++k)
          if (m.f[3 * t1 + k] != v1) xy[k2++] = m.f[3 * t1 + k];
        for (int32_t t0 : m.vt[v0])
          if (!m.has(t0, v1) && m.has(t0, xy[0]) && m.has(t0, xy[1])) { twin = true; break; }
        if (twin) break;
      }
      if (twin) continue;
    }
    // ---- collapse v1 into v0
    for (int32_t t : m.vt[v1]) {
      if (m.has(t, v0)) {
        m.f_alive[t] = 0;
        --alive;
      } else {
        for (int k = 0; k < 3; ++k)
          if (m.f[3 * t + k] == v1) m.f[3 * t + k] = v0;
        m.vt[v0].push_back(t);
      }
    }
    m.vt[v1].clear();
    m.vt[v1].shrink_to_fit();
    m.v_alive[v1] = 0;
    m.v[v0] = vbar;
    m.q[v0].add(m.q[v1]);
    ++m.ver[v0];
    m.clean(v0);
    // new costs of the edges around v0
    ++m.cur_stamp;
    for (int32_t t : m.vt[v0])
      for (int k = 0; k < 3; ++k) {
        const int32_t u = m.f[3 * t + k];
        if (u != v0 && m.stamp[u] != m.cur_stamp) {
          m.stamp[u] = m.cur_stamp;
          push_edge(v0, u);
        }
      }
  }
  // ---- compact
  std::vector<int32_t> remap(n_verts, -1);
  int64_t nv = 0, nf = 0;
  for (int32_t t = 0; t < n_faces; ++t) {
    if (!m.f_alive[t]) continue;
    for (int k = 0; k < 3; ++k) {
      const int32_t u = m.f[3 * t + k];
      if (remap[u] < 0) {
        remap[u] = (int32_t)nv;
        out_verts[3 * nv] = m.v[u].x; out_verts[3 * nv + 1] = m.v[u].y; out_verts[3 * nv + 2] = m.v[u].z;
        ++nv;
      }
      out_faces[3 * nf + k] = remap[u];
    }
    ++nf;
  }
  *out_n_verts = nv;
  *out_n_faces = nf;
  return DSU_OK;
}

extern "C" {

int dsu_mesh_decimate_quadric(const double* verts, int64_t n_verts, const int32_t* faces,
                              int64_t n_faces, int64_t target_faces, double boundary_weight,
                              int32_t flags, double* out_verts, int64_t* out_n_verts,
                              int32_t* out_faces, int64_t* out_n_faces) {
  return decimate_impl(verts, n_verts, faces, n_faces, target_faces, boundary_weight, flags, nullptr,
                       out_verts, out_n_verts, out_faces, out_n_faces);
}

int dsu_mesh_decimate_quadric_q(const double* verts, int64_t n_verts, const int32_t* faces,
                                int64_t n_faces, int64_t target_faces, double boundary_weight,
                                int32_t flags, const double* vertex_quadrics, double* out_verts,
                                int64_t* out_n_verts, int32_t* out_faces, int64_t* out_n_faces) {
  return decimate_impl(verts, n_verts, faces, n_faces, target_faces, boundary_weight, flags,
                       vertex_quadrics, out_verts, out_n_verts, out_faces, out_n_faces);
}

}  // extern "C"
