// Parallel quadric edge-collapse decimation on the device: the bulk of `remesh()` of the
// reference's export (instant_nsr/utils/mesh_utils.py:10-22, geometry.py:63-64: the 512^3
// marching-cubes mesh, 1-3 M triangles, down to face_count = 50 000).
//
// The reference hands this to Open3D's serial priority-queue algorithm (host).  The serial form
// (csrc/mesh_decimate.hip, same file header: Garland & Heckbert quadrics, boundary planes,
// minimiser-or-endpoints target, normal-flip rejection, link condition) costs ~13 s of one host
// core at that size — three times the whole NSR optimisation.  Here the same admissible-collapse
// rules run as ROUNDS of independent collapses:
//   1. vertex -> triangle lists (CSR) of the live mesh; one "owner" half-edge per edge;
//   2. cost + target of every edge (identical arithmetic to the serial code: edge_target);
//   3. the cheapest `budget` edges are candidates, ranked by (cost, edge order);
//   4. every candidate writes a (hashed, unique) priority into all vertices of its closed
//      neighbourhood (v0, v1 and their one-rings) with atomicMin; a candidate whose two END POINTS
//      still carry its priority is selected: no selected collapse has an end point equal or
//      adjacent to another's, so their flip tests, link conditions and updates cannot see each
//      other (see collapse_kernel);
//   5. selected + admissible collapses are applied in place; dead triangles are compacted away.
// A collapse rejected by the flip / link tests is remembered (direct-mapped table keyed by the
// edge, valid while neither end point's neighbourhood changed) so that it does not keep winning
// its neighbourhood round after round.
// The rounds stop above the target (`stop_faces`) and never go below `floor_faces`; the caller
// finishes the last stretch with the serial code seeded with the accumulated quadrics
// (dsu_mesh_decimate_quadric_q), which also lands on the exact face count.
//
// Everything is float64 (positions, quadrics, costs): near-planar regions have costs ~1e-20 that
// float32 quadrics cannot order.  Deterministic: vertex lists are sorted, ranks are unique, sums
// run in list order; atomics only take minima of unique integers.
// Scans / compaction / radix sort: hipCUB device primitives (header-only), on the caller's stream
// with the caller's workspace.
#include "common.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

namespace {

struct D3 {
  double x, y, z;
};
__device__ inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline D3 cross3(D3 a, D3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline double norm3(D3 a) { return sqrt(dot3(a, a)); }

// [A b; b^T c], 10 doubles: a00 a01 a02 a11 a12 a22 b0 b1 b2 c  (the serial code's Quadric)
struct Quad {
  double q[10];
};
__device__ inline void quad_zero(Quad& a) {
  for (int i = 0; i < 10; ++i) a.q[i] = 0.0;
}
__device__ inline void quad_add_plane(Quad& a, D3 n, double d, double w) {
  a.q[0] += w * n.x * n.x; a.q[1] += w * n.x * n.y; a.q[2] += w * n.x * n.z;
  a.q[3] += w * n.y * n.y; a.q[4] += w * n.y * n.z; a.q[5] += w * n.z * n.z;
  a.q[6] += w * n.x * d; a.q[7] += w * n.y * d; a.q[8] += w * n.z * d;
  a.q[9] += w * d * d;
}
__device__ inline double quad_eval(const Quad& a, D3 v) {
  return v.x * (a.q[0] * v.x + 2 * a.q[1] * v.y + 2 * a.q[2] * v.z) +
         v.y * (a.q[3] * v.y + 2 * a.q[4] * v.z) + a.q[5] * v.z * v.z +
         2 * (a.q[6] * v.x + a.q[7] * v.y + a.q[8] * v.z) + a.q[9];
}
__device__ inline bool quad_minimum(const Quad& a, D3& out) {
  const double a00 = a.q[0], a01 = a.q[1], a02 = a.q[2], a11 = a.q[3], a12 = a.q[4], a22 = a.q[5];
  const double b0 = a.q[6], b1 = a.q[7], b2 = a.q[8];
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

struct Mesh {
  double* P;        // (nv,3)
  double* Q;        // (nv,10)
  int32_t* F;       // (nf,3) live triangles
  int32_t* voff;    // (nv+1) CSR offsets
  int32_t* vfaces;  // (3 nf)
  int64_t nv, nf;
};

__device__ inline D3 ldP(const Mesh& m, int32_t v) { return {m.P[3 * v], m.P[3 * v + 1], m.P[3 * v + 2]}; }
__device__ inline Quad ldQ(const Mesh& m, int32_t v) {
  Quad a;
  for (int i = 0; i < 10; ++i) a.q[i] = m.Q[10 * (int64_t)v + i];
  return a;
}
__device__ inline bool tri_has(const Mesh& m, int32_t t, int32_t v) {
  return m.F[3 * t] == v || m.F[3 * t + 1] == v || m.F[3 * t + 2] == v;
}

// mesh_decimate.hip: edge_target
__device__ inline void edge_target(const Mesh& m, int32_t v0, int32_t v1, double& cost, D3& vbar) {
  Quad q = ldQ(m, v0);
  const Quad q1 = ldQ(m, v1);
  for (int i = 0; i < 10; ++i) q.q[i] += q1.q[i];
  const D3 p0 = ldP(m, v0), p1 = ldP(m, v1);
  if (quad_minimum(q, vbar)) {
    const double len = norm3(p1 - p0);
    const D3 mid = (p0 + p1) * 0.5;
    if (norm3(vbar - mid) <= 4.0 * len) {
      cost = quad_eval(q, vbar);
      return;
    }
  }
  const D3 cand[3] = {p0, p1, (p0 + p1) * 0.5};
  cost = quad_eval(q, cand[0]);
  vbar = cand[0];
  for (int i = 1; i < 3; ++i) {
    const double c = quad_eval(q, cand[i]);
    if (c < cost) { cost = c; vbar = cand[i]; }
  }
}

// ------------------------------------------------------------------------------------ CSR
__global__ void flag_nondegenerate_kernel(const int32_t* F, int64_t nf, uint8_t* flag) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nf; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t a = F[3 * t], b = F[3 * t + 1], c = F[3 * t + 2];
    flag[t] = (a != b && b != c && a != c) ? 1 : 0;
  }
}

__global__ void degree_kernel(const int32_t* F, int64_t nf, int32_t* deg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 3 * nf; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&deg[F[i]], 1);
}

__global__ void fill_kernel(const int32_t* F, int64_t nf, const int32_t* voff, int32_t* cursor, int32_t* vfaces) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 3 * nf; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = F[i];
    vfaces[voff[v] + atomicAdd(&cursor[v], 1)] = (int32_t)(i / 3);
  }
}

// insertion sort of every vertex's triangle list: list order (hence every later sum and scan) does
// not depend on the order the atomics above happened to run in
__global__ void sort_lists_kernel(const int32_t* voff, int32_t* vfaces, int64_t nv) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nv; v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = voff[v], e = voff[v + 1];
    for (int32_t i = b + 1; i < e; ++i) {
      const int32_t x = vfaces[i];
      int32_t j = i - 1;
      while (j >= b && vfaces[j] > x) { vfaces[j + 1] = vfaces[j]; --j; }
      vfaces[j + 1] = x;
    }
  }
}

// does some live triangle hold the half-edge a -> b ?
__device__ inline bool has_half_edge(const Mesh& m, int32_t a, int32_t b) {
  for (int32_t i = m.voff[a]; i < m.voff[a + 1]; ++i) {
    const int32_t t = m.vfaces[i];
    for (int k = 0; k < 3; ++k)
      if (m.F[3 * t + k] == a && m.F[3 * t + (k + 1) % 3] == b) return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------ quadrics
__global__ void init_quadrics_kernel(Mesh m, double boundary_weight) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < m.nv; v += (int64_t)gridDim.x * blockDim.x) {
    Quad q;
    quad_zero(q);
    for (int32_t i = m.voff[v]; i < m.voff[v + 1]; ++i) {
      const int32_t t = m.vfaces[i];
      const int32_t a = m.F[3 * t], b = m.F[3 * t + 1], c = m.F[3 * t + 2];
      const D3 pa = ldP(m, a), pb = ldP(m, b), pc = ldP(m, c);
      const D3 cr = cross3(pb - pa, pc - pa);
      const double l = norm3(cr);
      if (!(l > 0.0)) continue;
      const D3 n = cr * (1.0 / l);
      quad_add_plane(q, n, -dot3(n, pa), 0.5 * l);
      if (boundary_weight > 0.0) {
        // the two edges of t at v; an edge with no opposite half-edge is a boundary edge
        for (int k = 0; k < 3; ++k) {
          const int32_t e0 = m.F[3 * t + k], e1 = m.F[3 * t + (k + 1) % 3];
          if (e0 != (int32_t)v && e1 != (int32_t)v) continue;
          if (has_half_edge(m, e1, e0)) continue;
          // a second triangle with the SAME half-edge (non-manifold fan): not a boundary
          int same = 0;
          for (int32_t j = m.voff[e0]; j < m.voff[e0 + 1]; ++j) {
            const int32_t t2 = m.vfaces[j];
            for (int kk = 0; kk < 3; ++kk)
              if (m.F[3 * t2 + kk] == e0 && m.F[3 * t2 + (kk + 1) % 3] == e1) ++same;
          }
          if (same != 1) continue;
          const int32_t lo = e0 < e1 ? e0 : e1, hi = e0 < e1 ? e1 : e0;
          D3 en = cross3(ldP(m, hi) - ldP(m, lo), n);
          const double el = norm3(en);
          if (!(el > 0.0)) continue;
          en = en * (1.0 / el);
          quad_add_plane(q, en, -dot3(en, ldP(m, lo)), boundary_weight * 0.5 * l);
        }
      }
    }
    for (int i = 0; i < 10; ++i) m.Q[10 * v + i] = q.q[i];
  }
}

// ------------------------------------------------------------------------------------ edges
// half-edge i = 3 t + k (a -> b) owns its edge when a < b, or when a > b and no triangle holds b -> a
__global__ void owner_flags_kernel(Mesh m, uint8_t* flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 3 * m.nf; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / 3;
    const int k = (int)(i % 3);
    const int32_t a = m.F[3 * t + k], b = m.F[3 * t + (k + 1) % 3];
    flag[i] = (a < b || (a > b && !has_half_edge(m, b, a))) ? 1 : 0;
  }
}

// Direct-mapped memory of rejected collapses, one 64-bit word per slot:
//   [63:48] round of the rejection | [47:0] hash of (edge, version of v0, version of v1)
// written with atomicMax: a later round replaces an earlier one, and when several rejections of one
// round fall into one slot the larger hash stays — the same one on every run (a plain store would
// keep whichever writer came last).  A look-up compares the low 48 bits with the hash of the
// edge's CURRENT versions.
typedef unsigned long long Reject;

__device__ inline unsigned long long ekey(int32_t a, int32_t b) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  return ((unsigned long long)lo << 32) | hi;
}
__device__ inline unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
__device__ inline uint32_t slot_of(unsigned long long key, uint32_t n_slots) {
  return (uint32_t)(mix64(key) % n_slots);
}
__device__ inline unsigned long long reject_tag(unsigned long long key, uint32_t ver0, uint32_t ver1) {
  return mix64(key ^ mix64(((unsigned long long)ver0 << 32) | ver1)) & 0xffffffffffffull;
}

// cost of every owner half-edge, as the float32 bit pattern of max(cost, 0) (monotone as an
// unsigned integer); remembered rejections get +inf
__global__ void edge_cost_kernel(Mesh m, const int32_t* edges, int64_t ne, const Reject* rej, uint32_t n_slots,
                                 const uint32_t* vver, uint32_t* cost_bits, uint32_t* edge_idx) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < ne; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t h = edges[e];
    const int32_t t = h / 3, k = h % 3;
    const int32_t a = m.F[3 * t + k], b = m.F[3 * t + (k + 1) % 3];
    const int32_t v0 = a < b ? a : b, v1 = a < b ? b : a;
    double cost; D3 vb;
    edge_target(m, v0, v1, cost, vb);
    float c = (float)(cost > 0.0 ? cost : 0.0);
    const unsigned long long key = ekey(v0, v1);
    if ((rej[slot_of(key, n_slots)] & 0xffffffffffffull) == reject_tag(key, vver[v0], vver[v1])) c = INFINITY;
    cost_bits[e] = __float_as_uint(c);
    edge_idx[e] = (uint32_t)e;
  }
}

__device__ inline void edge_ends(const Mesh& m, int32_t h, int32_t& v0, int32_t& v1) {
  const int32_t t = h / 3, k = h % 3;
  const int32_t a = m.F[3 * t + k], b = m.F[3 * t + (k + 1) % 3];
  v0 = a < b ? a : b;
  v1 = a < b ? b : a;
}

// Priority of a candidate inside the independent-set selection: a hash of (edge, round) — NOT its
// cost rank.  Costs vary smoothly over the surface (and tie on regular meshes), so "the cheapest
// edge of its neighbourhood wins" leaves one winner per monotone chain: a few dozen collapses per
// round on a 150 k-triangle sphere.  The cost decides who is a candidate (the cheapest `ncand`
// edges); a locally random, globally unique 64-bit priority decides which non-conflicting subset
// of them goes first.
__device__ inline unsigned long long priority_of(int32_t v0, int32_t v1, uint32_t round, uint32_t r) {
  unsigned long long k = ekey(v0, v1) ^ ((unsigned long long)round * 0x9e3779b97f4a7c15ull);
  k ^= k >> 31; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 29; k *= 0x94d049bb133111ebull; k ^= k >> 32;
  return (k << 32) | r;                                      // unique: r is the candidate's index
}

// every candidate stamps its priority on all vertices of its closed neighbourhood (v0, v1, rings)
// Several selection passes share one round's lists, edge list and cost order.  After a pass, the
// lists of the end points of applied collapses are stale (v1 is gone, v0 gained triangles) and so
// is every triangle that held a v1: a candidate is `dirty` — skipped until the next round — when
// any vertex of its closed neighbourhood was an end point of a collapse applied earlier in this
// round (`touched[v] == round`).  Everything a clean candidate reads is as it was at round start.
__device__ inline bool is_dirty(const Mesh& m, int32_t v0, int32_t v1, const uint32_t* touched, uint32_t round) {
  for (int pass = 0; pass < 2; ++pass) {
    const int32_t v = pass ? v1 : v0;
    for (int32_t i = m.voff[v]; i < m.voff[v + 1]; ++i) {
      const int32_t t = m.vfaces[i];
      for (int k = 0; k < 3; ++k)
        if (touched[m.F[3 * t + k]] == round) return true;
    }
  }
  return touched[v0] == round || touched[v1] == round;
}

__global__ void claim_kernel(Mesh m, const int32_t* edges, const uint32_t* sorted_idx, const uint32_t* sorted_cost,
                             int64_t ncand, uint32_t round, uint32_t seed, const uint32_t* touched,
                             const Reject* rej, uint32_t n_slots, const uint32_t* vver,
                             unsigned long long* claim) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < ncand; r += (int64_t)gridDim.x * blockDim.x) {
    if (sorted_cost[r] == 0x7f800000u) continue;            // +inf: remembered rejection
    int32_t v0, v1;
    edge_ends(m, edges[sorted_idx[r]], v0, v1);
    if (is_dirty(m, v0, v1, touched, round)) continue;
    {                                                       // rejected in an earlier pass of this round
      const unsigned long long key = ekey(v0, v1);
      if ((rej[slot_of(key, n_slots)] & 0xffffffffffffull) == reject_tag(key, vver[v0], vver[v1])) continue;
    }
    const unsigned long long pr = priority_of(v0, v1, seed, (uint32_t)r);
    for (int pass = 0; pass < 2; ++pass) {
      const int32_t v = pass ? v1 : v0;
      for (int32_t i = m.voff[v]; i < m.voff[v + 1]; ++i) {
        const int32_t t = m.vfaces[i];
        for (int k = 0; k < 3; ++k) atomicMin(&claim[m.F[3 * t + k]], pr);
      }
    }
  }
}

// Selected = both END POINTS still carry this candidate's priority.  If an end point of another
// candidate B lies in the closed neighbourhood of A, A stamped it and B stamped one of A's end
// points (adjacency is symmetric), so at most one of the two keeps both of its end points: selected
// collapses never have an end point equal or adjacent to another's.  That is all they need: a
// collapse moves v0, deletes v1 and rewrites only triangles around v1; its tests read the
// positions of ring vertices (not moved: not end points of a selected collapse) and the triangles
// around v0 / v1 (not rewritten: they would have to contain another collapse's v1, which would
// then be a ring vertex).  Rings may overlap.
// selected -> admissibility tests of the serial code -> apply
__global__ void collapse_kernel(Mesh m, const int32_t* edges, const uint32_t* sorted_idx, const uint32_t* sorted_cost,
                                int64_t ncand, uint32_t round, uint32_t seed, const unsigned long long* claim,
                                int link_test, uint8_t* f_alive, Reject* rej, uint32_t n_slots, uint32_t* vver,
                                uint32_t* touched_next, const uint32_t* touched, int32_t* counters) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < ncand; r += (int64_t)gridDim.x * blockDim.x) {
    if (sorted_cost[r] == 0x7f800000u) continue;
    int32_t v0, v1;
    edge_ends(m, edges[sorted_idx[r]], v0, v1);
    const unsigned long long pr = priority_of(v0, v1, seed, (uint32_t)r);
    if (claim[v0] != pr || claim[v1] != pr) continue;
    if (is_dirty(m, v0, v1, touched, round)) continue;      // (its claim was never written: belt and braces)
    double cost; D3 vbar;
    edge_target(m, v0, v1, cost, vbar);
    const int32_t b0 = m.voff[v0], e0 = m.voff[v0 + 1], b1 = m.voff[v1], e1 = m.voff[v1 + 1];
    int shared = 0;
    bool bad = false;
    for (int pass = 0; pass < 2 && !bad; ++pass) {
      const int32_t mv = pass ? v0 : v1, other = pass ? v1 : v0;
      for (int32_t i = m.voff[mv]; i < m.voff[mv + 1]; ++i) {
        const int32_t t = m.vfaces[i];
        if (tri_has(m, t, other)) { if (!pass) ++shared; continue; }
        D3 p[3], q[3];
        for (int k = 0; k < 3; ++k) {
          p[k] = ldP(m, m.F[3 * t + k]);
          q[k] = m.F[3 * t + k] == mv ? vbar : p[k];
        }
        const D3 before = cross3(p[1] - p[0], p[2] - p[0]);
        const D3 after = cross3(q[1] - q[0], q[2] - q[0]);
        if (dot3(before, after) < 0.0) { bad = true; break; }
      }
    }
    if (!bad && shared == 0) bad = true;
    if (!bad && link_test) {
      // |N(v0) ∩ N(v1)| must equal the number of triangles on the edge
      int common = 0;
      for (int32_t i = b0; i < e0; ++i) {
        const int32_t t = m.vfaces[i];
        for (int k = 0; k < 3; ++k) {
          const int32_t u = m.F[3 * t + k];
          if (u == v0 || u == v1) continue;
          // first occurrence of u in the ring of v0 ?
          bool first = true;
          for (int32_t j = b0; j < i && first; ++j)
            if (tri_has(m, m.vfaces[j], u)) first = false;
          if (first)
            for (int kk = 0; kk < k; ++kk)
              if (m.F[3 * t + kk] == u) first = false;
          if (!first) continue;
          bool in1 = false;
          for (int32_t j = b1; j < e1 && !in1; ++j)
            if (tri_has(m, m.vfaces[j], u)) in1 = true;
          if (in1) ++common;
        }
      }
      if (common != shared) bad = true;
      // no triangle (v1, x, y) next to a triangle (v0, x, y)
      for (int32_t j = b1; j < e1 && !bad; ++j) {
        const int32_t t1 = m.vfaces[j];
        if (tri_has(m, t1, v0)) continue;
        int32_t xy[2], k2 = 0;
        for (int k = 0; k < 3; ++k)
          if (m.F[3 * t1 + k] != v1 && k2 < 2) xy[k2++] = m.F[3 * t1 + k];
        if (k2 < 2) continue;
        for (int32_t i = b0; i < e0; ++i) {
          const int32_t t0 = m.vfaces[i];
          if (!tri_has(m, t0, v1) && tri_has(m, t0, xy[0]) && tri_has(m, t0, xy[1])) { bad = true; break; }
        }
      }
    }
    if (bad) {
      const unsigned long long key = ekey(v0, v1);
      atomicMax(&rej[slot_of(key, n_slots)],
                ((unsigned long long)(round & 0xffffu) << 48) | reject_tag(key, vver[v0], vver[v1]));
      atomicAdd(&counters[1], 1);
      continue;
    }
    // ---- collapse v1 into v0 (this thread owns every vertex and triangle it touches)
    for (int pass = 0; pass < 2; ++pass) {                  // the neighbourhood changed: versions
      const int32_t v = pass ? v1 : v0;
      for (int32_t i = m.voff[v]; i < m.voff[v + 1]; ++i) {
        const int32_t t = m.vfaces[i];
        for (int k = 0; k < 3; ++k) atomicAdd(&vver[m.F[3 * t + k]], 1u);   // rings may be shared
      }
    }
    int removed = 0;
    for (int32_t j = b1; j < e1; ++j) {
      const int32_t t = m.vfaces[j];
      if (tri_has(m, t, v0)) {
        f_alive[t] = 0;
        ++removed;
      } else {
        for (int k = 0; k < 3; ++k)
          if (m.F[3 * t + k] == v1) m.F[3 * t + k] = v0;
      }
    }
    m.P[3 * v0] = vbar.x; m.P[3 * v0 + 1] = vbar.y; m.P[3 * v0 + 2] = vbar.z;
    for (int i = 0; i < 10; ++i) m.Q[10 * (int64_t)v0 + i] += m.Q[10 * (int64_t)v1 + i];
    touched_next[v0] = round;                                // visible to the NEXT pass (separate array:
    touched_next[v1] = round;                                // this pass's dirty tests read `touched`)
    atomicAdd(&counters[0], 1);
    atomicAdd(&counters[2], removed);
  }
}

struct Tri {
  int32_t a, b, c;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t off_Q, off_F2, off_voff, off_cursor, off_vfaces, off_flag, off_edges, off_iota, off_cost, off_cost2,
      off_idx, off_idx2, off_claim, off_vver, off_touched, off_touched2, off_rej, off_counters, off_nsel, off_cub, cub_bytes, total;
  uint32_t n_slots;
};

Layout make_layout(int64_t nv, int64_t nf) {
  Layout L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align256(bytes); return r; };
  L.off_Q = take((size_t)nv * 10 * sizeof(double));
  L.off_F2 = take((size_t)nf * 3 * sizeof(int32_t));
  L.off_voff = take((size_t)(nv + 1) * sizeof(int32_t));
  L.off_cursor = take((size_t)(nv + 1) * sizeof(int32_t));
  L.off_vfaces = take((size_t)nf * 3 * sizeof(int32_t));
  L.off_flag = take((size_t)nf * 3);
  L.off_edges = take((size_t)nf * 3 * sizeof(int32_t));
  L.off_iota = take((size_t)nf * 3 * sizeof(int32_t));
  L.off_cost = take((size_t)nf * 3 * sizeof(uint32_t));
  L.off_cost2 = take((size_t)nf * 3 * sizeof(uint32_t));
  L.off_idx = take((size_t)nf * 3 * sizeof(uint32_t));
  L.off_idx2 = take((size_t)nf * 3 * sizeof(uint32_t));
  L.off_claim = take((size_t)nv * sizeof(unsigned long long));
  L.off_vver = take((size_t)nv * sizeof(uint32_t));
  L.off_touched = take((size_t)nv * sizeof(uint32_t));
  L.off_touched2 = take((size_t)nv * sizeof(uint32_t));
  L.n_slots = (uint32_t)(nv * 2 + 1024);
  L.off_rej = take((size_t)L.n_slots * sizeof(Reject));
  L.off_counters = take(64);
  L.off_nsel = take(64);
  // hipCUB temporary storage: the largest of the calls below at the largest sizes
  size_t b_scan = 0, b_sel_i = 0, b_sel_t = 0, b_sort = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, b_scan, (int32_t*)nullptr, (int32_t*)nullptr, (int)(nv + 1));
  hipcub::DeviceSelect::Flagged(nullptr, b_sel_i, (int32_t*)nullptr, (uint8_t*)nullptr, (int32_t*)nullptr,
                                (int32_t*)nullptr, (int)(3 * nf));
  hipcub::DeviceSelect::Flagged(nullptr, b_sel_t, (Tri*)nullptr, (uint8_t*)nullptr, (Tri*)nullptr, (int32_t*)nullptr,
                                (int)nf);
  hipcub::DeviceRadixSort::SortPairs(nullptr, b_sort, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)(3 * nf));
  L.cub_bytes = std::max(std::max(b_scan, b_sel_i), std::max(b_sel_t, b_sort)) + 256;
  L.off_cub = take(L.cub_bytes);
  L.total = o;
  return L;
}

__global__ void iota_kernel(int32_t* p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int32_t)i;
}

}  // namespace

#define DSU_HIP_TRY(expr)                       \
  do {                                          \
    if ((expr) != hipSuccess) return DSU_ELAUNCH; \
  } while (0)

extern "C" {

int64_t dsu_mesh_decimate_parallel_workspace_bytes(int64_t n_verts, int64_t n_faces) {
  if (n_verts < 0 || n_faces < 0 || n_verts > (1 << 30) || n_faces > (1 << 29)) return DSU_EINVAL;
  return (int64_t)make_layout(n_verts, n_faces).total;
}

int dsu_mesh_decimate_parallel(double* verts, int64_t n_verts, int32_t* faces, int64_t n_faces,
                               int64_t stop_faces, int64_t floor_faces, double boundary_weight, int32_t flags,
                               int32_t max_rounds, double* out_quadrics, int64_t* out_n_faces,
                               int32_t* out_stats, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n_verts < 0 || n_faces < 0 || stop_faces < floor_faces || floor_faces < 0 || !out_n_faces ||
      (n_verts && !verts) || (n_faces && !faces) || !(boundary_weight >= 0.0) || max_rounds < 0 ||
      n_verts > (1 << 30) || n_faces > (1 << 29))
    return DSU_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (out_stats) out_stats[0] = out_stats[1] = out_stats[2] = 0;
  *out_n_faces = n_faces;
  if (n_faces == 0 || n_verts == 0) return DSU_OK;
  const Layout L = make_layout(n_verts, n_faces);
  if (!workspace || workspace_bytes < (int64_t)L.total) return DSU_EINVAL;
  char* w = (char*)workspace;
  double* Q = (double*)(w + L.off_Q);
  int32_t* F2 = (int32_t*)(w + L.off_F2);
  int32_t* voff = (int32_t*)(w + L.off_voff);
  int32_t* cursor = (int32_t*)(w + L.off_cursor);
  int32_t* vfaces = (int32_t*)(w + L.off_vfaces);
  uint8_t* flag = (uint8_t*)(w + L.off_flag);
  int32_t* edges = (int32_t*)(w + L.off_edges);
  int32_t* iota = (int32_t*)(w + L.off_iota);
  uint32_t* cost = (uint32_t*)(w + L.off_cost);
  uint32_t* cost2 = (uint32_t*)(w + L.off_cost2);
  uint32_t* idx = (uint32_t*)(w + L.off_idx);
  uint32_t* idx2 = (uint32_t*)(w + L.off_idx2);
  unsigned long long* claim = (unsigned long long*)(w + L.off_claim);
  uint32_t* vver = (uint32_t*)(w + L.off_vver);
  uint32_t* touched = (uint32_t*)(w + L.off_touched);
  uint32_t* touched2 = (uint32_t*)(w + L.off_touched2);
  Reject* rej = (Reject*)(w + L.off_rej);
  int32_t* counters = (int32_t*)(w + L.off_counters);
  int32_t* nsel = (int32_t*)(w + L.off_nsel);
  void* cub = (void*)(w + L.off_cub);
  const int T = 256;
  const int PASSES = 4;                     // selection passes per round (see is_dirty)
  const int link_test = !(flags & 1);

  int32_t* F = faces;                       // live triangles, compacted in place (through F2)
  int64_t nf = n_faces;
  int32_t host_n = 0;
  auto compact_faces = [&]() -> int {       // flag[0..nf) marks the survivors
    size_t b = L.cub_bytes;
    if (hipcub::DeviceSelect::Flagged(cub, b, (Tri*)F, flag, (Tri*)F2, nsel, (int)nf, s) != hipSuccess)
      return DSU_ELAUNCH;
    if (hipMemcpyAsync(&host_n, nsel, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return DSU_ELAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return DSU_ELAUNCH;
    nf = host_n;
    if (nf && hipMemcpyAsync(F, F2, (size_t)nf * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return DSU_ELAUNCH;
    return DSU_OK;
  };
  auto build_csr = [&]() -> int {
    DSU_HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)(n_verts + 1) * sizeof(int32_t), s));
    degree_kernel<<<dsu_capped_blocks(3 * nf, T), T, 0, s>>>(F, nf, cursor);
    size_t b = L.cub_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(cub, b, cursor, voff, (int)(n_verts + 1), s) != hipSuccess)
      return DSU_ELAUNCH;
    DSU_HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)(n_verts + 1) * sizeof(int32_t), s));
    fill_kernel<<<dsu_capped_blocks(3 * nf, T), T, 0, s>>>(F, nf, voff, cursor, vfaces);
    sort_lists_kernel<<<dsu_capped_blocks(n_verts, T), T, 0, s>>>(voff, vfaces, n_verts);
    DSU_CHECK_LAUNCH();
    return DSU_OK;
  };

  // triangles with a repeated vertex carry no surface
  flag_nondegenerate_kernel<<<dsu_capped_blocks(nf, T), T, 0, s>>>(F, nf, flag);
  DSU_CHECK_LAUNCH();
  int rc = compact_faces();
  if (rc) return rc;
  DSU_HIP_TRY(hipMemsetAsync(vver, 0, (size_t)n_verts * sizeof(uint32_t), s));
  DSU_HIP_TRY(hipMemsetAsync(touched, 0xff, (size_t)n_verts * sizeof(uint32_t), s));
  DSU_HIP_TRY(hipMemsetAsync(touched2, 0xff, (size_t)n_verts * sizeof(uint32_t), s));
  DSU_HIP_TRY(hipMemsetAsync(rej, 0, (size_t)L.n_slots * sizeof(Reject), s));
  iota_kernel<<<dsu_capped_blocks(3 * nf, T), T, 0, s>>>(iota, 3 * nf);
  if ((rc = build_csr())) return rc;
  Mesh m{verts, Q, F, voff, vfaces, n_verts, nf};
  init_quadrics_kernel<<<dsu_capped_blocks(n_verts, T), T, 0, s>>>(m, boundary_weight);
  DSU_CHECK_LAUNCH();

  int rounds = 0, applied_total = 0, rejected_total = 0, stall = 0;
  while (nf > stop_faces && rounds < max_rounds) {
    const int64_t budget = (nf - floor_faces) / 2;
    if (budget < 1) break;
    m.nf = nf;
    // owner half-edges -> edge list
    owner_flags_kernel<<<dsu_capped_blocks(3 * nf, T), T, 0, s>>>(m, flag);
    DSU_CHECK_LAUNCH();
    size_t b = L.cub_bytes;
    if (hipcub::DeviceSelect::Flagged(cub, b, iota, flag, edges, nsel, (int)(3 * nf), s) != hipSuccess)
      return DSU_ELAUNCH;
    DSU_HIP_TRY(hipMemcpyAsync(&host_n, nsel, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DSU_HIP_TRY(hipStreamSynchronize(s));
    const int64_t ne = host_n;
    if (ne == 0) break;
    edge_cost_kernel<<<dsu_capped_blocks(ne, T), T, 0, s>>>(m, edges, ne, rej, L.n_slots, vver, cost, idx);
    DSU_CHECK_LAUNCH();
    b = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(cub, b, cost, cost2, idx, idx2, (int)ne, 0, 32, s) != hipSuccess)
      return DSU_ELAUNCH;
    int64_t ncand = budget < ne / 2 ? budget : ne / 2;
    if (ncand < 1) ncand = 1;
    DSU_HIP_TRY(hipMemsetAsync(counters, 0, 64, s));
    DSU_HIP_TRY(hipMemsetAsync(flag, 1, (size_t)nf, s));
    for (int pass = 0; pass < PASSES; ++pass) {
      const uint32_t seed = (uint32_t)rounds * 16u + (uint32_t)pass;
      DSU_HIP_TRY(hipMemsetAsync(claim, 0xff, (size_t)n_verts * sizeof(unsigned long long), s));
      claim_kernel<<<dsu_capped_blocks(ncand, T), T, 0, s>>>(m, edges, idx2, cost2, ncand, (uint32_t)rounds, seed,
                                                             touched, rej, L.n_slots, vver, claim);
      collapse_kernel<<<dsu_capped_blocks(ncand, T), T, 0, s>>>(m, edges, idx2, cost2, ncand, (uint32_t)rounds, seed,
                                                                claim, link_test, flag, rej, L.n_slots, vver,
                                                                touched2, touched, counters);
      DSU_CHECK_LAUNCH();
      DSU_HIP_TRY(hipMemcpyAsync(touched, touched2, (size_t)n_verts * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    int32_t hc[3];
    DSU_HIP_TRY(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, s));
    DSU_HIP_TRY(hipStreamSynchronize(s));
    ++rounds;
    if (dsu_ab_int("DSU_DECIMATE_DEBUG", 0))
      fprintf(stderr, "[decimate] round %d nf %lld ne %lld ncand %lld applied %d rejected %d removed %d\n", rounds,
              (long long)nf, (long long)ne, (long long)ncand, hc[0], hc[1], hc[2]);
    applied_total += hc[0];
    rejected_total += hc[1];
    if (hc[0] == 0) {
      // nothing admissible among the winners: their rejections are remembered, the next round sees
      // other winners; give up when that does not help either
      if (++stall >= 4) break;
      continue;
    }
    stall = 0;
    const int64_t before = nf;
    if ((rc = compact_faces())) return rc;
    // a round that removes under 0.5 % of a mesh already within 4x of the target (candidates
    // clustered in one place): the serial queue is the faster way from here
    if ((before - nf) * 200 < before && nf <= 4 * stop_faces) break;
    if (nf == 0) break;
    if ((rc = build_csr())) return rc;
  }
  if (out_quadrics)
    DSU_HIP_TRY(hipMemcpyAsync(out_quadrics, Q, (size_t)n_verts * 10 * sizeof(double), hipMemcpyDeviceToDevice, s));
  DSU_HIP_TRY(hipStreamSynchronize(s));
  *out_n_faces = nf;
  if (out_stats) { out_stats[0] = rounds; out_stats[1] = applied_total; out_stats[2] = rejected_total; }
  return DSU_OK;
}

}  // extern "C"
