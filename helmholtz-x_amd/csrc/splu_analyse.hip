// Host analysis of the sparse direct solver (hx_splu_analyse): fill-reducing ordering, supernodes, fronts, storage plan.
//
// replaces: the symbolic phase of PCLU / MUMPS behind ST sinvert (eigensolvers.py:49-50,102-103 of the reference; METIS / SCOTCH
// orderings inside MUMPS).  No device code in this file.
//
//  1. graph of the pattern of A + A^T (no diagonal);
//  2. nested dissection.  With vertex coordinates: recursive INERTIAL bisection (cut perpendicular to the principal axis of the
//     point cloud at its median), without: level-set bisection from a pseudo-peripheral vertex (George's automatic nested
//     dissection).  The vertex separator is the smaller one-sided boundary of the edge cut.  Parts of at most `leaf_size` vertices
//     are not cut further;
//  3. every leaf and every separator is ONE supernode (treated as dense; separators of a mesh are connected sets that fill in
//     anyway).  Border rows by the supernodal symbolic factorisation over the true supernodal elimination tree (parent = owner of
//     the first border row), children merged in ascending order;
//  4. the plan the device factorisation follows without further decisions: levels (height above the leaves), panel / inverse
//     offsets, a first-fit simulated arena for the contribution blocks (allocated when a level starts, released when the parent's
//     level has been assembled), the destination of every stored entry of the input CSR, extend-add positions.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <atomic>
#include <numeric>
#include <stdlib.h>
#include <thread>

#include "hx_common.h"
#include "splu_internal.h"

namespace {

struct Graph {
  int32_t n;
  std::vector<int64_t> xadj;
  std::vector<int32_t> adj;
};

void build_graph(Graph& g, int32_t n, const int32_t* rowptr, const int32_t* col, bool symmetric) {
  g.n = n;
  if (symmetric) {                          // the pattern is its own transpose: the rows are the adjacency lists
    g.xadj.assign((size_t)n + 1, 0);
    g.adj.resize((size_t)rowptr[n]);
    int64_t w = 0;
    for (int32_t i = 0; i < n; ++i) {
      for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (col[k] != i) g.adj[w++] = col[k];
      g.xadj[i + 1] = w;
    }
    g.adj.resize((size_t)w);
    return;
  }
  g.xadj.assign((size_t)n + 1, 0);
  for (int32_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t j = col[k];
      if (j != i) { g.xadj[i + 1]++; g.xadj[j + 1]++; }
    }
  for (int32_t i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
  std::vector<int32_t> raw((size_t)g.xadj[n]);
  std::vector<int64_t> pos(g.xadj.begin(), g.xadj.end() - 1);
  for (int32_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t j = col[k];
      if (j != i) { raw[pos[i]++] = j; raw[pos[j]++] = i; }
    }
  // sort + unique every list, compact
  std::vector<int64_t> nx((size_t)n + 1, 0);
  int64_t w = 0;
  for (int32_t i = 0; i < n; ++i) {
    int32_t* b = raw.data() + g.xadj[i];
    int32_t* e = raw.data() + g.xadj[i + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    nx[i] = w;
    for (int32_t* p = b; p < e; ++p) raw[w++] = *p;      // w never passes the read position
  }
  nx[n] = w;
  raw.resize((size_t)w);
  g.xadj.swap(nx);
  g.adj.swap(raw);
}

struct Dissector {
  const Graph& g;
  const double* xy;          // n x dim or null
  int dim;
  int leaf;
  int threads;
  std::vector<int32_t> verts;        // permuted in place: every call owns a contiguous slice; at the end: new -> old
  std::vector<int32_t> mark;         // which call / side a vertex currently belongs to (ids are unique across threads)
  std::vector<double> proj;
  std::vector<int32_t> queue, lvl, scratch;      // indexed by POSITION in verts: calls on disjoint slices never collide
  std::atomic<int32_t> stamp{0};

  Dissector(const Graph& g_, const double* xy_, int dim_, int leaf_, int threads_)
      : g(g_), xy(xy_), dim(dim_), leaf(leaf_), threads(threads_) {
    verts.resize(g.n);
    std::iota(verts.begin(), verts.end(), 0);
    mark.assign(g.n, -1);
    proj.resize(g.n);
    queue.resize(g.n);
    lvl.resize(g.n);
    scratch.resize(g.n);
  }

  // splits verts[lo, hi) into [lo, mid) and [mid, hi) by a median plane.  Candidate normals: the principal axes of the point
  // cloud whose extent is within a factor 8 of the largest (a mesh with 10 : 1 cells has FEWER vertices across a geometrically
  // longer direction: the longest axis alone picks 1248-vertex separators where 169 do, measured on the synthetic annulus);
  // the candidate with the smallest one-sided boundary of its cut wins.  false: degenerate (all points equal)
  bool split_inertial(int32_t lo, int32_t hi, int32_t& mid) {
    const int32_t cnt = hi - lo;
    double c[3] = {0, 0, 0}, M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int32_t i = lo; i < hi; ++i)
      for (int d = 0; d < dim; ++d) c[d] += xy[(size_t)verts[i] * dim + d];
    for (int d = 0; d < dim; ++d) c[d] /= cnt;
    for (int32_t i = lo; i < hi; ++i) {
      double r[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) r[d] = xy[(size_t)verts[i] * dim + d] - c[d];
      for (int a = 0; a < dim; ++a)
        for (int b = 0; b < dim; ++b) M[a][b] += r[a] * r[b];
    }
    // principal axes by power iteration with deflation (deterministic starts, not aligned with any mesh axis)
    double axes[3][3], lam[3] = {0, 0, 0};
    int naxes = 0;
    const double starts[3][3] = {{1.0, 0.913, 0.791}, {0.671, -1.0, 0.853}, {0.517, 0.771, -1.0}};
    for (int e = 0; e < dim; ++e) {
      double v[3] = {starts[e][0], starts[e][1], starts[e][2]};
      double ev = 0;
      bool ok = true;
      for (int it = 0; it < 60 && ok; ++it) {
        for (int p = 0; p < naxes; ++p) {               // stay orthogonal to the axes found so far
          double dp = 0;
          for (int a = 0; a < dim; ++a) dp += v[a] * axes[p][a];
          for (int a = 0; a < dim; ++a) v[a] -= dp * axes[p][a];
        }
        double w[3] = {0, 0, 0}, nrm = 0;
        for (int a = 0; a < dim; ++a)
          for (int b = 0; b < dim; ++b) w[a] += M[a][b] * v[b];
        for (int a = 0; a < dim; ++a) nrm += w[a] * w[a];
        nrm = std::sqrt(nrm);
        if (!(nrm > 0)) { ok = false; break; }
        ev = nrm;
        for (int a = 0; a < dim; ++a) v[a] = w[a] / nrm;
      }
      if (!ok) break;
      for (int p = 0; p < naxes; ++p) {
        double dp = 0;
        for (int a = 0; a < dim; ++a) dp += v[a] * axes[p][a];
        for (int a = 0; a < dim; ++a) v[a] -= dp * axes[p][a];
      }
      double nn = 0;
      for (int a = 0; a < dim; ++a) nn += v[a] * v[a];
      if (!(nn > 1e-20)) break;
      nn = std::sqrt(nn);
      for (int a = 0; a < dim; ++a) v[a] /= nn;
      // snap to a coordinate axis when within ~8 degrees of it: planar cuts on structured grids
      for (int a = 0; a < dim; ++a)
        if (std::fabs(v[a]) > 0.99) { for (int b = 0; b < dim; ++b) v[b] = (b == a) ? 1.0 : 0.0; break; }
      for (int a = 0; a < 3; ++a) axes[naxes][a] = a < dim ? v[a] : 0.0;
      lam[naxes] = ev;
      ++naxes;
    }
    if (naxes == 0 || !(lam[0] > 0)) return false;
    auto project = [&](const double* v) {
      for (int32_t i = lo; i < hi; ++i) {
        double p = 0;
        for (int d = 0; d < dim; ++d) p += v[d] * (xy[(size_t)verts[i] * dim + d] - c[d]);
        proj[verts[i]] = p;
      }
    };
    int32_t* sc = scratch.data() + lo;
    auto threshold = [&](double& t, double& eps) {
      for (int32_t i = 0; i < cnt; ++i) sc[i] = verts[lo + i];
      std::nth_element(sc, sc + cnt / 2, sc + cnt, [&](int32_t a, int32_t b) { return proj[a] < proj[b]; });
      t = proj[sc[cnt / 2]];
      double span = 0;
      for (int32_t i = 0; i < cnt; ++i) span = std::max(span, std::fabs(proj[sc[i]]));
      eps = 1e-9 * (span > 0 ? span : 1.0);
    };
    int best = 0;
    int ncand = 1;
    while (ncand < naxes && lam[ncand] * 64.0 >= lam[0]) ++ncand;
    if (ncand > 1 && cnt > 4 * leaf) {
      const int32_t id = ++stamp;
      for (int32_t i = lo; i < hi; ++i) mark[verts[i]] = id;
      // size of a candidate's separator, estimated without a pass over the edges: the vertices within half a typical
      // edge (its projection on the normal, from the edges of ~2000 sampled vertices) of the median plane
      double best_sep = -1;
      const int32_t stride = std::max<int32_t>(1, cnt / 2048);
      for (int e = 0; e < ncand; ++e) {
        project(axes[e]);
        double t, eps;
        threshold(t, eps);
        double hsum = 0;
        int64_t hcnt = 0;
        for (int32_t i = lo; i < hi; i += stride) {
          const int32_t v = verts[i];
          for (int64_t k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
            const int32_t u = g.adj[k];
            if (mark[u] == id) { hsum += std::fabs(proj[u] - proj[v]); ++hcnt; }
          }
        }
        if (hcnt == 0) continue;
        const double half = 0.5 * hsum / (double)hcnt + eps;
        int64_t nA = 0, slab = 0;
        for (int32_t i = lo; i < hi; ++i) {
          const double p = proj[verts[i]];
          nA += p < t - eps;
          slab += std::fabs(p - t) <= half;
        }
        if (nA == 0 || nA == cnt) continue;
        if (best_sep < 0 || (double)slab < best_sep) { best_sep = (double)slab; best = e; }
      }
    }
    project(axes[best]);
    double t, eps;
    threshold(t, eps);
    auto below = [&](int32_t u) { return proj[u] < t - eps; };
    int32_t* b = verts.data() + lo;
    int32_t* m = std::partition(b, b + cnt, below);
    if (m == b) {                                   // everything at or above the median value: cut above it instead
      m = std::partition(b, b + cnt, [&](int32_t u) { return proj[u] <= t + eps; });
      if (m == b + cnt) return false;
    }
    mid = lo + (int32_t)(m - b);
    return mid > lo && mid < hi;
  }

  // BFS over the slice (vertices with mark == id) from `root`; appends to queue[qlo..), returns new end; lvl[] by position
  int32_t bfs(int32_t root, int32_t id, int32_t visited_id, int32_t qlo, int32_t start_level) {
    int32_t head = qlo, tail = qlo;
    queue[tail] = root; lvl[tail] = start_level; ++tail;
    mark[root] = visited_id;
    while (head < tail) {
      const int32_t v = queue[head];
      const int32_t l = lvl[head];
      ++head;
      for (int64_t k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int32_t u = g.adj[k];
        if (mark[u] == id) { mark[u] = visited_id; queue[tail] = u; lvl[tail] = l + 1; ++tail; }
      }
    }
    return tail;
  }

  bool split_levelset(int32_t lo, int32_t hi, int32_t& mid) {
    const int32_t cnt = hi - lo;
    int32_t id = ++stamp;
    for (int32_t i = lo; i < hi; ++i) mark[verts[i]] = id;
    // two sweeps towards a pseudo-peripheral vertex of the first component
    int32_t root = verts[lo];
    for (int sweep = 0; sweep < 2; ++sweep) {
      const int32_t vis = ++stamp;
      const int32_t end = bfs(root, id, vis, lo, 0);
      root = queue[end - 1];
      for (int32_t i = lo; i < end; ++i) mark[queue[i]] = id;
    }
    const int32_t vis = ++stamp;
    int32_t end = bfs(root, id, vis, lo, 0);
    for (int32_t i = lo; i < hi && end < hi; ++i)          // further components, each starting a new level
      if (mark[verts[i]] == id) end = bfs(verts[i], id, vis, end, lvl[end - 1] + 1);
    // cut at the level boundary closest to the middle
    int32_t best = -1;
    for (int32_t i = 1; i < cnt; ++i)
      if (lvl[lo + i] != lvl[lo + i - 1] && (best < 0 || std::abs(i - cnt / 2) < std::abs(best - cnt / 2))) best = i;
    if (best < 0) best = cnt / 2;
    for (int32_t i = lo; i < hi; ++i) verts[i] = queue[i];
    mid = lo + best;
    return mid > lo && mid < hi;
  }

  // `ends`: end positions of the supernodes emitted by this call tree (every emitted slice is final in verts)
  void dissect(int32_t lo, int32_t hi, int depth, std::vector<int32_t>& ends) {
    const int32_t cnt = hi - lo;
    if (cnt <= 0) return;
    if (cnt <= leaf) { ends.push_back(hi); return; }
    int32_t mid = lo;
    bool ok = (xy != nullptr) && split_inertial(lo, hi, mid);
    if (!ok) ok = split_levelset(lo, hi, mid);
    if (!ok) { ends.push_back(hi); return; }
    const int32_t idA = ++stamp, idB = ++stamp;
    for (int32_t i = lo; i < mid; ++i) mark[verts[i]] = idA;
    for (int32_t i = mid; i < hi; ++i) mark[verts[i]] = idB;
    auto touches = [&](int32_t v, int32_t other) {
      for (int64_t k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
        if (mark[g.adj[k]] == other) return true;
      return false;
    };
    int32_t nbA = 0, nbB = 0;
    for (int32_t i = lo; i < mid; ++i) nbA += touches(verts[i], idB) ? 1 : 0;
    for (int32_t i = mid; i < hi; ++i) nbB += touches(verts[i], idA) ? 1 : 0;
    int32_t a_end, b_end;                        // [lo, a_end) = A', [a_end, b_end) = B', [b_end, hi) = separator
    if (nbA <= nbB) {                            // separator taken out of A
      int32_t* m = std::stable_partition(verts.data() + lo, verts.data() + mid, [&](int32_t v) { return !touches(v, idB); });
      const int32_t a1 = (int32_t)(m - verts.data());           // [a1, mid) = separator, then B follows: rotate it to the end
      std::rotate(verts.begin() + a1, verts.begin() + mid, verts.begin() + hi);
      a_end = a1; b_end = a1 + (hi - mid);
    } else {
      int32_t* m = std::stable_partition(verts.data() + mid, verts.data() + hi, [&](int32_t v) { return !touches(v, idA); });
      a_end = mid; b_end = (int32_t)(m - verts.data());
    }
    if (a_end - lo == 0 && b_end - a_end == 0) { ends.push_back(hi); return; }
    // the two parts share no edge: independent from here on (threads on the first levels of a large problem)
    if (threads > 1 && (1 << depth) < threads && a_end - lo > 8192 && b_end - a_end > 8192) {
      std::vector<int32_t> other;
      std::thread th([&]() { dissect(lo, a_end, depth + 1, other); });
      dissect(a_end, b_end, depth + 1, ends);
      th.join();
      ends.insert(ends.end(), other.begin(), other.end());
    } else {
      dissect(lo, a_end, depth + 1, ends);
      dissect(a_end, b_end, depth + 1, ends);
    }
    if (hi > b_end) ends.push_back(hi);
  }
};

}  // namespace

int splu_analyse_host(SpluSymbolic& S, int32_t n, const int32_t* rowptr, const int32_t* col, const double* coords, int dim,
                      int leaf_size, int flags) {
  const auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (getenv("HX_SPLU_TRACE")) fprintf(stderr, "  splu analyse: %-22s %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); };
  if (n <= 0) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "n <= 0");
  if (coords && (dim < 1 || dim > 3)) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "dim must be 1, 2 or 3");
  for (int32_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (col[k] < 0 || col[k] >= n) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "column index out of range");
  Graph g;
  build_graph(g, n, rowptr, col, (flags & 1) != 0);
  lap("graph");
  const int threads = (int)std::max<long>(1, std::min<long>(hx_option_int("host_threads", 8), (long)std::thread::hardware_concurrency()));
  Dissector D(g, coords, dim, leaf_size < 8 ? 8 : leaf_size, threads);
  std::vector<int32_t> ends;
  D.dissect(0, n, 0, ends);
  lap("dissection");
  std::sort(ends.begin(), ends.end());
  if (ends.empty() || ends.back() != n) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "internal: ordering lost vertices");
  S.n = n;
  S.perm.swap(D.verts);
  S.iperm.assign(n, 0);
  for (int32_t i = 0; i < n; ++i) S.iperm[S.perm[i]] = i;
  S.sfirst.assign(1, 0);
  S.sfirst.insert(S.sfirst.end(), ends.begin(), ends.end());
  const int32_t ns_ = S.nsuper = (int32_t)S.sfirst.size() - 1;
  std::vector<int32_t> super_of(n);
  for (int32_t s = 0; s < ns_; ++s)
    for (int32_t j = S.sfirst[s]; j < S.sfirst[s + 1]; ++j) super_of[j] = s;

  // ---- supernodal symbolic factorisation over the true elimination tree
  S.sparent.assign(ns_, -1);
  S.sptr.assign((size_t)ns_ + 1, 0);
  S.srow.clear();
  std::vector<int32_t> head(ns_, -1), tail(ns_, -1), next(ns_, -1), flag(n, -1), rows;
  for (int32_t s = 0; s < ns_; ++s) {
    const int32_t first = S.sfirst[s], last = S.sfirst[s + 1];
    rows.clear();
    for (int32_t j = first; j < last; ++j) {
      const int32_t v = S.perm[j];
      for (int64_t k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
        const int32_t r = S.iperm[g.adj[k]];
        if (r >= last && flag[r] != s) { flag[r] = s; rows.push_back(r); }
      }
    }
    for (int32_t c = head[s]; c >= 0; c = next[c])
      for (int64_t k = S.sptr[c]; k < S.sptr[c + 1]; ++k) {
        const int32_t r = S.srow[k];
        if (r >= last && flag[r] != s) { flag[r] = s; rows.push_back(r); }
      }
    std::sort(rows.begin(), rows.end());
    S.srow.insert(S.srow.end(), rows.begin(), rows.end());
    S.sptr[s + 1] = (int64_t)S.srow.size();
    if (!rows.empty()) {
      const int32_t p = super_of[rows[0]];
      S.sparent[s] = p;
      if (head[p] < 0) head[p] = s; else next[tail[p]] = s;
      tail[p] = s;
    }
  }
  lap("symbolic");
  S.child_ptr.assign((size_t)ns_ + 1, 0);
  S.child_list.clear();
  for (int32_t s = 0; s < ns_; ++s) {
    for (int32_t c = head[s]; c >= 0; c = next[c]) S.child_list.push_back(c);
    S.child_ptr[s + 1] = (int32_t)S.child_list.size();
  }
  // ---- extend-add positions
  S.rel.assign(S.srow.size(), 0);
  for (int32_t c = 0; c < ns_; ++c) {
    const int32_t p = S.sparent[c];
    if (p < 0) continue;
    const int32_t pf = S.sfirst[p], pl = S.sfirst[p + 1], pns = pl - pf;
    int64_t q = S.sptr[p];
    for (int64_t k = S.sptr[c]; k < S.sptr[c + 1]; ++k) {
      const int32_t r = S.srow[k];
      if (r < pl) { S.rel[k] = r - pf; continue; }
      while (q < S.sptr[p + 1] && S.srow[q] < r) ++q;
      if (q >= S.sptr[p + 1] || S.srow[q] != r) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "internal: border row missing in parent");
      S.rel[k] = pns + (int32_t)(q - S.sptr[p]);
    }
  }
  // ---- levels
  S.slevel.assign(ns_, 0);
  for (int32_t s = 0; s < ns_; ++s) {
    const int32_t p = S.sparent[s];
    if (p >= 0) S.slevel[p] = std::max(S.slevel[p], S.slevel[s] + 1);
  }
  S.nlevels = 0;
  for (int32_t s = 0; s < ns_; ++s) S.nlevels = std::max(S.nlevels, S.slevel[s] + 1);
  S.level_ptr.assign((size_t)S.nlevels + 1, 0);
  for (int32_t s = 0; s < ns_; ++s) S.level_ptr[S.slevel[s] + 1]++;
  for (int32_t l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
  S.level_list.assign(ns_, 0);
  {
    std::vector<int32_t> pos(S.level_ptr.begin(), S.level_ptr.end() - 1);
    for (int32_t s = 0; s < ns_; ++s) S.level_list[pos[S.slevel[s]]++] = s;
  }
  lap("children, rel, levels");
  // ---- storage plan
  S.loff.assign(ns_, 0); S.uoff.assign(ns_, 0); S.ioff.assign(ns_, 0); S.soff.assign(ns_, -1); S.woff.assign(ns_, 0);
  int64_t fcur = 0, icur = 0, wcur = 0;
  S.nnz_L = 0; S.flops = 0; S.max_front = 0; S.max_ns = 0;
  for (int32_t s = 0; s < ns_; ++s) {
    const int64_t ns = S.sfirst[s + 1] - S.sfirst[s], nr = S.sptr[s + 1] - S.sptr[s], m = ns + nr;
    S.loff[s] = fcur; fcur += m * ns;
    S.uoff[s] = fcur; fcur += m * ns;
    S.ioff[s] = icur; icur += (int64_t)SPLU_NB * SPLU_NB * ((ns + SPLU_NB - 1) / SPLU_NB);
    S.woff[s] = wcur; wcur += m + ns;            // m front entries + ns more (y / t of a large front's sweeps)
    S.nnz_L += ns * (ns + 1) / 2 + nr * ns;
    S.flops += 8.0 * ((double)ns * ns * ns / 3.0 + (double)nr * ns * ns + (double)nr * nr * ns);
    S.max_front = std::max<int32_t>(S.max_front, (int32_t)m);
    S.max_ns = std::max<int32_t>(S.max_ns, (int32_t)ns);
  }
  S.factor_elems = fcur; S.inverse_elems = icur; S.work_elems = wcur;
  {
    // first-fit arena simulation: contribution blocks of a level are allocated when the level starts and released when the
    // parents' level has absorbed them
    std::map<int64_t, int64_t> freeb;           // offset -> size
    int64_t top = 0;
    auto alloc = [&](int64_t sz) -> int64_t {
      for (auto it = freeb.begin(); it != freeb.end(); ++it)
        if (it->second >= sz) {
          const int64_t off = it->first, rest = it->second - sz;
          freeb.erase(it);
          if (rest > 0) freeb[off + sz] = rest;
          return off;
        }
      // grow: extend a free block that ends at the top, if any
      if (!freeb.empty()) {
        auto lastb = std::prev(freeb.end());
        if (lastb->first + lastb->second == top) {
          const int64_t off = lastb->first;
          freeb.erase(lastb);
          top = off + sz;
          return off;
        }
      }
      const int64_t off = top;
      top += sz;
      return off;
    };
    auto release = [&](int64_t off, int64_t sz) {
      auto it = freeb.emplace(off, sz).first;
      auto nx = std::next(it);
      if (nx != freeb.end() && it->first + it->second == nx->first) { it->second += nx->second; freeb.erase(nx); }
      if (it != freeb.begin()) {
        auto pv = std::prev(it);
        if (pv->first + pv->second == it->first) { pv->second += it->second; freeb.erase(it); }
      }
    };
    for (int32_t l = 0; l < S.nlevels; ++l) {
      for (int32_t k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
        const int32_t s = S.level_list[k];
        const int64_t nr = S.sptr[s + 1] - S.sptr[s];
        if (nr > 0) S.soff[s] = alloc(nr * nr);
      }
      for (int32_t k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
        const int32_t s = S.level_list[k];
        for (int32_t q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {
          const int32_t c = S.child_list[q];
          const int64_t nr = S.sptr[c + 1] - S.sptr[c];
          if (nr > 0) release(S.soff[c], nr * nr);
        }
      }
    }
    S.temp_elems = top;
  }
  lap("storage plan");
  // ---- destination of every stored entry of the input
  S.amap.assign((size_t)rowptr[n], -1);
  S.mirror.clear();
  if (flags & 1) S.mirror.assign((size_t)rowptr[n], -1);
  std::atomic<int> bad{0}, nomirror{0};
  auto map_rows = [&](int32_t i0, int32_t i1) {
  for (int32_t i = i0; i < i1; ++i) {
    const int32_t pi = S.iperm[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t pj = S.iperm[col[k]];
      const bool lower = pi >= pj;
      const int32_t piv = lower ? pj : pi, oth = lower ? pi : pj;
      const int32_t s = super_of[piv];
      const int64_t ns = S.sfirst[s + 1] - S.sfirst[s], nr = S.sptr[s + 1] - S.sptr[s], m = ns + nr;
      int64_t a;
      if (oth < S.sfirst[s + 1]) a = oth - S.sfirst[s];
      else {
        const int32_t* b = S.srow.data() + S.sptr[s];
        const int32_t* e = S.srow.data() + S.sptr[s + 1];
        const int32_t* p = std::lower_bound(b, e, oth);
        if (p == e || *p != oth) { bad = 1; continue; }
        a = ns + (p - b);
      }
      S.amap[k] = lower ? S.loff[s] + a + (int64_t)(piv - S.sfirst[s]) * m          // Lp[a, pivot]: m x ns, ld m
                        : -(S.uoff[s] + (int64_t)(piv - S.sfirst[s]) + a * ns) - 1;  // Up[pivot, a]: ns x m, ld ns; stored as -(dest + 1)
      if ((flags & 1) && !lower) {                 // position of the mirrored entry (col[k], i): values are compared on the device
        const int32_t j = col[k];
        const int32_t* b = col + rowptr[j];
        const int32_t* e = col + rowptr[j + 1];
        const int32_t* p = std::lower_bound(b, e, i);
        if (p == e || *p != i) nomirror = 1; else S.mirror[k] = (int32_t)(rowptr[j] + (p - b));
      }
    }
  }
  };
  {
    std::vector<std::thread> pool;
    const int nt = n > 50000 ? threads : 1;
    for (int t = 1; t < nt; ++t) pool.emplace_back(map_rows, (int32_t)((int64_t)n * t / nt), (int32_t)((int64_t)n * (t + 1) / nt));
    map_rows(0, (int32_t)((int64_t)n / nt));
    for (auto& th : pool) th.join();
  }
  if (bad) return hx_fail(HX_ERR_ARG, "hx_splu_analyse", "internal: entry outside its front");
  if (nomirror) S.mirror.clear();                  // (the pattern was not its own transpose after all: no symmetric mode)
  lap("entry map");
  S.analyse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return HX_OK;
}
