// nsk_partition.cpp -- multilevel k-way partitioner in front of the samplers' range partition.
//
//   nsk_graph_partition  <-  salt/src/messages.py:593-670  find_metis_parts (nxmetis.partition, objtype = vol)
//
// The reference hands the variable graph to METIS and stores variable -> part.  This is the same kind of
// partitioner, host C++ and O(edges) per level, with the two things the samplers here need on top:
//   * the parts have EXACTLY the sizes of the shard formula  [g n / G, (g + 1) n / G)  (inference.py:17-18), because
//     a partition is a variable ORDER in front of that formula (numbskull_amd/partition.py): order[new id] = old id;
//   * the objective of the last refinement is nsk_comm_volume itself -- (variable, foreign part) pairs read across
//     the cut, the values one boundary exchange moves -- not the edge cut the coarse levels work with.
// Steps: (1) variable graph: a factor's members pairwise (long factors: ring + star), edge weight = factors shared;
// (2) heavy-edge matching down to a few thousand vertices; (3) coarsest graph: a weighted maximum-adjacency walk
// smoothed by weighted-median placement (nsk_graph_order's method 3 with weights), cut by vertex weight; (4) on the
// way up, greedy k-way boundary refinement of the edge cut under a weight bound; (5) finest level: the parts are
// brought to their exact sizes by moving the best-gain boundary vertices along the chain of parts, then refined with
// the exact change in communication volume of every candidate move, sizes kept (a move into a full part is paired
// with the best move out of it).  Deterministic: every random choice comes from a generator seeded by `seed`.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"
#include "nsk_compile.h"          // nsk::parallel_for (host threads)

namespace nsk { void set_error(const std::string &m); }   // nsk_api.hip (thread-local message)

namespace {

struct Level {
    int64_t n = 0, totw = 0;
    std::vector<int64_t> xadj;       // n + 1
    std::vector<int32_t> adj, ew, vw;
    std::vector<int32_t> cmap;       // vertex -> vertex of the next coarser level
};

struct Rng {
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
};

static void random_permutation(std::vector<int32_t> &p, int64_t n, Rng &rng) {
    p.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) p[(size_t)i] = (int32_t)i;
    for (int64_t i = n - 1; i > 0; i--) std::swap(p[(size_t)i], p[(size_t)rng.below((uint64_t)i + 1)]);
}

// (1) the variable graph of the factor lists
static void build_fine(int64_t nvar, int64_t nfactor, const nsk_factor *factor, const nsk_ftv *fmap, Level &g) {
    const int64_t CLIQUE = 6;        // members pairwise up to this arity; beyond: neighbours in the list + the first
    g.n = nvar;
    g.totw = nvar;
    g.vw.assign((size_t)nvar, 1);
    std::vector<int64_t> deg((size_t)nvar + 1, 0);
    auto each_pair = [&](int64_t f, auto &&emit) {
        if (factor[f].factorFunction == -1) return;      // NOOP ties nothing together (remove_noop, messages.py:674)
        const int64_t s = factor[f].ftv_offset, a = factor[f].arity;
        if (a <= CLIQUE) {
            for (int64_t i = 0; i < a; i++)
                for (int64_t j = i + 1; j < a; j++) emit(fmap[s + i].vid, fmap[s + j].vid);
        } else {
            for (int64_t i = 0; i + 1 < a; i++) emit(fmap[s + i].vid, fmap[s + i + 1].vid);
            for (int64_t i = 2; i < a; i++) emit(fmap[s].vid, fmap[s + i].vid);
        }
    };
    for (int64_t f = 0; f < nfactor; f++)
        each_pair(f, [&](int64_t u, int64_t v) { if (u != v) { deg[(size_t)u + 1]++; deg[(size_t)v + 1]++; } });
    for (int64_t v = 0; v < nvar; v++) deg[(size_t)v + 1] += deg[(size_t)v];
    std::vector<int32_t> raw((size_t)deg[(size_t)nvar]);
    {
        std::vector<int64_t> fill(deg.begin(), deg.end() - 1);
        for (int64_t f = 0; f < nfactor; f++)
            each_pair(f, [&](int64_t u, int64_t v) {
                if (u != v) { raw[(size_t)fill[(size_t)u]++] = (int32_t)v; raw[(size_t)fill[(size_t)v]++] = (int32_t)u; }
            });
    }
    // per vertex: sort, count the distinct neighbours, then write (neighbour, multiplicity)
    g.xadj.assign((size_t)nvar + 1, 0);
    nsk::parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
        for (int64_t v = b0; v < b1; v++) {
            int32_t *lo = raw.data() + deg[(size_t)v], *hi = raw.data() + deg[(size_t)v + 1];
            std::sort(lo, hi);
            int64_t d = 0;
            for (int32_t *p = lo; p < hi; p++) d += (p == lo || p[0] != p[-1]);
            g.xadj[(size_t)v + 1] = d;
        }
    }, 4096);
    for (int64_t v = 0; v < nvar; v++) g.xadj[(size_t)v + 1] += g.xadj[(size_t)v];
    g.adj.resize((size_t)g.xadj[(size_t)nvar]);
    g.ew.resize((size_t)g.xadj[(size_t)nvar]);
    nsk::parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
        for (int64_t v = b0; v < b1; v++) {
            const int32_t *lo = raw.data() + deg[(size_t)v], *hi = raw.data() + deg[(size_t)v + 1];
            int64_t o = g.xadj[(size_t)v] - 1;
            for (const int32_t *p = lo; p < hi; p++) {
                if (p == lo || p[0] != p[-1]) { o++; g.adj[(size_t)o] = p[0]; g.ew[(size_t)o] = 1; }
                else g.ew[(size_t)o]++;
            }
        }
    }, 4096);
}

// (2) one level of heavy-edge matching: false when the graph no longer shrinks
static bool coarsen(Level &f, Level &c, int64_t maxvw, Rng &rng) {
    const int64_t n = f.n;
    std::vector<int32_t> perm, match((size_t)n, -1);
    random_permutation(perm, n, rng);
    for (int64_t i = 0; i < n; i++) {
        const int32_t v = perm[(size_t)i];
        if (match[(size_t)v] >= 0) continue;
        int32_t best = -1, bw = 0, bvw = 0;
        for (int64_t j = f.xadj[(size_t)v]; j < f.xadj[(size_t)v + 1]; j++) {
            const int32_t u = f.adj[(size_t)j];
            if (match[(size_t)u] >= 0 || (int64_t)f.vw[(size_t)v] + f.vw[(size_t)u] > maxvw) continue;
            if (best < 0 || f.ew[(size_t)j] > bw || (f.ew[(size_t)j] == bw && f.vw[(size_t)u] < bvw)) {
                best = u; bw = f.ew[(size_t)j]; bvw = f.vw[(size_t)u];
            }
        }
        if (best >= 0) { match[(size_t)v] = best; match[(size_t)best] = v; }
    }
    // what the matching left over: a vertex whose neighbours are all taken pairs up with another neighbour of one of
    // them (two hops: leaves of a star), a vertex without neighbours with the next such vertex -- or the graph would
    // stop shrinking at its isolated variables (a config-#5 variable that heads only unary factors)
    int32_t lone = -1;
    for (int64_t i = 0; i < n; i++) {
        const int32_t v = perm[(size_t)i];
        if (match[(size_t)v] >= 0) continue;
        const int64_t d = f.xadj[(size_t)v + 1] - f.xadj[(size_t)v];
        if (d == 0) {
            if (lone >= 0 && (int64_t)f.vw[(size_t)lone] + f.vw[(size_t)v] <= maxvw) { match[(size_t)v] = lone; match[(size_t)lone] = v; lone = -1; }
            else lone = v;
            continue;
        }
        for (int64_t j = f.xadj[(size_t)v]; j < std::min(f.xadj[(size_t)v] + 4, f.xadj[(size_t)v + 1]) && match[(size_t)v] < 0; j++) {
            const int32_t u = f.adj[(size_t)j];
            for (int64_t j2 = f.xadj[(size_t)u]; j2 < std::min(f.xadj[(size_t)u] + 32, f.xadj[(size_t)u + 1]); j2++) {
                const int32_t w = f.adj[(size_t)j2];
                if (w == v || match[(size_t)w] >= 0 || (int64_t)f.vw[(size_t)v] + f.vw[(size_t)w] > maxvw) continue;
                match[(size_t)v] = w; match[(size_t)w] = v;
                break;
            }
        }
    }
    f.cmap.assign((size_t)n, -1);
    int64_t nc = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t v = perm[(size_t)i];
        if (f.cmap[(size_t)v] >= 0) continue;
        if (match[(size_t)v] < 0) match[(size_t)v] = v;
        f.cmap[(size_t)v] = (int32_t)nc;
        f.cmap[(size_t)match[(size_t)v]] = (int32_t)nc;
        nc++;
    }
    if (nc > n - n / 10) return false;
    c = Level();
    c.n = nc;
    c.totw = f.totw;
    c.vw.assign((size_t)nc, 0);
    c.xadj.assign((size_t)nc + 1, 0);
    // members of every coarse vertex (first = the smaller id; the list is rebuilt from match[])
    std::vector<int32_t> first((size_t)nc, -1);
    for (int64_t v = 0; v < n; v++) {
        const int32_t cv = f.cmap[(size_t)v];
        c.vw[(size_t)cv] += f.vw[(size_t)v];
        if (first[(size_t)cv] < 0) first[(size_t)cv] = (int32_t)v;
    }
    std::vector<int32_t> slot((size_t)nc, -1);
    c.adj.reserve(f.adj.size() / 2 + 16);
    c.ew.reserve(f.adj.size() / 2 + 16);
    for (int64_t cv = 0; cv < nc; cv++) {
        const int64_t base = (int64_t)c.adj.size();
        const int32_t a = first[(size_t)cv], b = match[(size_t)a];
        for (int k = 0; k < (b != a ? 2 : 1); k++) {
            const int32_t v = k ? b : a;
            for (int64_t j = f.xadj[(size_t)v]; j < f.xadj[(size_t)v + 1]; j++) {
                const int32_t cu = f.cmap[(size_t)f.adj[(size_t)j]];
                if (cu == cv) continue;
                if (slot[(size_t)cu] < base) {                 // (slots are indices into c.adj: anything below `base` is stale)
                    slot[(size_t)cu] = (int32_t)c.adj.size();
                    c.adj.push_back(cu);
                    c.ew.push_back(f.ew[(size_t)j]);
                } else c.ew[(size_t)slot[(size_t)cu]] += f.ew[(size_t)j];
            }
        }
        c.xadj[(size_t)cv + 1] = (int64_t)c.adj.size();
    }
    return true;
}

// (3) a linear arrangement of a (small) weighted graph: maximum-adjacency walk, then rounds of "half way to the
// weighted median of the neighbours' positions, re-rank"; positions are in units of vertex weight
static void linear_order(const Level &g, std::vector<int32_t> &ord, int rounds) {
    const int64_t n = g.n;
    std::vector<int64_t> score((size_t)n, 0);
    std::vector<uint8_t> done((size_t)n, 0);
    std::vector<int32_t> comp((size_t)n, 0);
    ord.clear();
    ord.reserve((size_t)n);
    typedef std::pair<int64_t, int64_t> Item;                    // (score, -arrival)
    std::priority_queue<Item> heap;
    std::vector<std::pair<int32_t, int64_t>> entries;            // arrival -> (vertex, score at push)
    int32_t ncomp = 0;
    for (int64_t s0 = 0; s0 < n; s0++) {
        if (done[(size_t)s0]) continue;
        // start from a far end of the component: last vertex of a breadth-first walk from s0
        int32_t start = (int32_t)s0;
        {
            std::vector<int32_t> q(1, (int32_t)s0);
            std::vector<int32_t> seen;
            comp[(size_t)s0] = -1;
            for (size_t h = 0; h < q.size(); h++)
                for (int64_t j = g.xadj[(size_t)q[h]]; j < g.xadj[(size_t)q[h] + 1]; j++) {
                    const int32_t u = g.adj[(size_t)j];
                    if (comp[(size_t)u] != -1 && !done[(size_t)u]) { comp[(size_t)u] = -1; q.push_back(u); }
                }
            start = q.back();
        }
        entries.emplace_back(start, 0);
        heap.emplace(0, -(int64_t)(entries.size() - 1));
        while (!heap.empty()) {
            const Item it = heap.top();
            heap.pop();
            const auto en = entries[(size_t)(-it.second)];
            const int32_t v = en.first;
            if (done[(size_t)v] || en.second != score[(size_t)v]) continue;
            done[(size_t)v] = 1;
            comp[(size_t)v] = ncomp;
            ord.push_back(v);
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) {
                const int32_t u = g.adj[(size_t)j];
                if (done[(size_t)u]) continue;
                score[(size_t)u] += g.ew[(size_t)j];
                entries.emplace_back(u, score[(size_t)u]);
                heap.emplace(score[(size_t)u], -(int64_t)(entries.size() - 1));
            }
        }
        ncomp++;
    }
    std::vector<int64_t> cstart((size_t)ncomp + 1, 0);
    for (int64_t v = 0; v < n; v++) cstart[(size_t)comp[(size_t)v] + 1]++;
    for (int32_t k = 0; k < ncomp; k++) cstart[(size_t)k + 1] += cstart[(size_t)k];
    std::vector<double> x((size_t)n), xn((size_t)n);
    auto place = [&]() { double cum = 0; for (int64_t i = 0; i < n; i++) { const int32_t v = ord[(size_t)i]; x[(size_t)v] = cum + 0.5 * g.vw[(size_t)v]; cum += g.vw[(size_t)v]; } };
    place();
    for (int r = 0; r < rounds; r++) {
        nsk::parallel_for(n, [&](int64_t b0, int64_t b1, int) {
            std::vector<std::pair<double, int32_t>> nb;
            for (int64_t v = b0; v < b1; v++) {
                nb.clear();
                int64_t tot = 0;
                for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) { nb.emplace_back(x[(size_t)g.adj[(size_t)j]], g.ew[(size_t)j]); tot += g.ew[(size_t)j]; }
                double m = x[(size_t)v];
                if (!nb.empty()) {
                    std::sort(nb.begin(), nb.end());
                    int64_t acc = 0;
                    for (size_t i = 0; i < nb.size(); i++) {
                        acc += nb[i].second;
                        if (2 * acc >= tot) { m = (2 * acc == tot && i + 1 < nb.size()) ? 0.5 * (nb[i].first + nb[i + 1].first) : nb[i].first; break; }
                    }
                }
                xn[(size_t)v] = 0.5 * x[(size_t)v] + 0.5 * m;
            }
        }, 1024);
        for (int32_t k = 0; k < ncomp; k++)
            std::stable_sort(ord.begin() + (std::ptrdiff_t)cstart[(size_t)k], ord.begin() + (std::ptrdiff_t)cstart[(size_t)k + 1],
                             [&](int32_t a, int32_t b) { return xn[(size_t)a] < xn[(size_t)b]; });
        place();
    }
}

// recursive bisection of the vertices `verts` of g into parts [p0, p1): the induced subgraph is arranged linearly and
// cut where the cumulative weight reaches the parts' share; bound[] = prefix sums of the parts' target weights
static void bisect(const Level &g, const std::vector<int32_t> &verts, int p0, int p1, const std::vector<int64_t> &bound,
                   std::vector<uint8_t> &where) {
    if (p1 - p0 <= 1 || verts.empty()) { for (int32_t v : verts) where[(size_t)v] = (uint8_t)p0; return; }
    Level sub;
    sub.n = (int64_t)verts.size();
    std::vector<int32_t> local((size_t)g.n, -1);
    for (size_t i = 0; i < verts.size(); i++) local[(size_t)verts[i]] = (int32_t)i;
    sub.xadj.assign(verts.size() + 1, 0);
    sub.vw.resize(verts.size());
    for (size_t i = 0; i < verts.size(); i++) {
        const int32_t v = verts[i];
        sub.vw[i] = g.vw[(size_t)v];
        sub.totw += g.vw[(size_t)v];
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++)
            if (local[(size_t)g.adj[(size_t)j]] >= 0) { sub.adj.push_back(local[(size_t)g.adj[(size_t)j]]); sub.ew.push_back(g.ew[(size_t)j]); }
        sub.xadj[i + 1] = (int64_t)sub.adj.size();
    }
    std::vector<int32_t> ord;
    linear_order(sub, ord, 20);
    const int pm = p0 + (p1 - p0) / 2;
    const double share = (double)(bound[(size_t)pm] - bound[(size_t)p0]) / (double)std::max<int64_t>(1, bound[(size_t)p1] - bound[(size_t)p0]);
    const int64_t cutw = (int64_t)(share * (double)sub.totw);
    std::vector<int32_t> left, right;
    int64_t cum = 0;
    for (size_t i = 0; i < ord.size(); i++) {
        const int32_t lv = ord[i];
        (cum + sub.vw[(size_t)lv] / 2 < cutw ? left : right).push_back(verts[(size_t)lv]);
        cum += sub.vw[(size_t)lv];
    }
    bisect(g, left, p0, pm, bound, where);
    bisect(g, right, pm, p1, bound, where);
}

struct Parts {
    int k;
    std::vector<int64_t> target, pw;     // per part: the weight the shard formula gives it, the weight it has
};

static int64_t edge_cut(const Level &g, const std::vector<uint8_t> &where) {
    int64_t cut = 0;
    for (int64_t v = 0; v < g.n; v++)
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++)
            if (where[(size_t)g.adj[(size_t)j]] != where[(size_t)v]) cut += g.ew[(size_t)j];
    return cut / 2;
}

// (4) greedy k-way boundary refinement of the edge cut; a part may weigh up to target * (1 + eps) (+ slack)
static void refine_cut(const Level &g, std::vector<uint8_t> &where, Parts &P, double eps, int passes, Rng &rng) {
    const int64_t n = g.n;
    const int k = P.k;
    std::vector<int64_t> maxw((size_t)k);
    int32_t heaviest = 0;
    for (int64_t v = 0; v < n; v++) heaviest = std::max(heaviest, g.vw[(size_t)v]);
    for (int p = 0; p < k; p++) maxw[(size_t)p] = (int64_t)(P.target[(size_t)p] * (1.0 + eps)) + (heaviest > 1 ? heaviest / 2 : 0);
    std::vector<int64_t> conn((size_t)k, 0);
    std::vector<int32_t> cand;
    for (int pass = 0; pass < passes; pass++) {
        // the pass visits, in random order, the vertices that can move: boundary vertices and members of overweight parts
        cand.clear();
        for (int64_t v = 0; v < n; v++) {
            const int own = where[(size_t)v];
            bool take = P.pw[(size_t)own] > maxw[(size_t)own];
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1] && !take; j++) take = where[(size_t)g.adj[(size_t)j]] != own;
            if (take) cand.push_back((int32_t)v);
        }
        for (int64_t i = (int64_t)cand.size() - 1; i > 0; i--) std::swap(cand[(size_t)i], cand[(size_t)rng.below((uint64_t)i + 1)]);
        int64_t moves = 0;
        for (size_t i = 0; i < cand.size(); i++) {
            const int32_t v = cand[i];
            const int own = where[(size_t)v];
            const bool over = P.pw[(size_t)own] > maxw[(size_t)own];
            std::fill(conn.begin(), conn.end(), 0);
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) conn[(size_t)where[(size_t)g.adj[(size_t)j]]] += g.ew[(size_t)j];
            const int64_t vw = g.vw[(size_t)v];
            int best = -1;
            for (int q = 0; q < k; q++) {
                if (q == own || P.pw[(size_t)q] + vw > maxw[(size_t)q]) continue;
                if (!over && conn[(size_t)q] == 0) continue;
                if (best < 0 || conn[(size_t)q] > conn[(size_t)best] ||
                    (conn[(size_t)q] == conn[(size_t)best] && P.pw[(size_t)q] * P.target[(size_t)best] < P.pw[(size_t)best] * P.target[(size_t)q])) best = q;
            }
            if (best < 0) continue;
            const int64_t gain = conn[(size_t)best] - conn[(size_t)own];
            const bool lighter = (P.pw[(size_t)best] + vw) * P.target[(size_t)own] < P.pw[(size_t)own] * P.target[(size_t)best];
            if (gain > 0 || (gain == 0 && lighter) || (over && lighter)) {
                where[(size_t)v] = (uint8_t)best;
                P.pw[(size_t)own] -= vw;
                P.pw[(size_t)best] += vw;
                moves++;
            }
        }
        if (!moves) break;
    }
}

// (5a) exact sizes at the finest level (vertex weight 1): part g must hold target[g] vertices.  The parts' quotient
// graph (weight = edges cut between two parts) gets a maximum spanning tree; the flow over a tree edge is fixed by the
// excess of the subtree below it, and the vertices that cross are taken best edge-cut gain first, gains updated as
// their neighbours follow.
static void move_between(const Level &g, std::vector<uint8_t> &where, Parts &P, int from, int to, int64_t need) {
    typedef std::pair<int64_t, int32_t> Item;                    // (gain, vertex)
    auto gain_of = [&](int32_t v) {
        int64_t gn = 0;
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) {
            const int w = where[(size_t)g.adj[(size_t)j]];
            if (w == to) gn += g.ew[(size_t)j]; else if (w == from) gn -= g.ew[(size_t)j];
        }
        return gn;
    };
    std::priority_queue<Item> heap;
    std::vector<int32_t> rest;                                   // members of `from` with no neighbour in `to`
    for (int64_t v = 0; v < g.n; v++) {
        if (where[(size_t)v] != from) continue;
        bool adj_to = false;
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) if (where[(size_t)g.adj[(size_t)j]] == to) { adj_to = true; break; }
        if (adj_to) heap.emplace(gain_of((int32_t)v), (int32_t)v); else rest.push_back((int32_t)v);
    }
    // (members without any link cost nothing to move: they go as soon as the best linked vertex would cost something)
    std::stable_sort(rest.begin(), rest.end(), [&](int32_t a, int32_t c) {
        return g.xadj[(size_t)a + 1] - g.xadj[(size_t)a] < g.xadj[(size_t)c + 1] - g.xadj[(size_t)c]; });
    size_t rest_at = 0;
    while (need > 0) {
        while (rest_at < rest.size() && where[(size_t)rest[rest_at]] != from) rest_at++;
        const bool free_one = rest_at < rest.size() && g.xadj[(size_t)rest[rest_at] + 1] == g.xadj[(size_t)rest[rest_at]];
        if (free_one && (heap.empty() || heap.top().first <= 0)) {
            where[(size_t)rest[rest_at++]] = (uint8_t)to;
            P.pw[(size_t)from]--; P.pw[(size_t)to]++;
            need--;
            continue;
        }
        if (heap.empty()) {
            // nothing of `from` touches `to` (any more): the members with the fewest links first
            if (rest_at >= rest.size()) break;                   // (cannot happen: `from` holds more than it gives)
            heap.emplace(gain_of(rest[rest_at]), rest[rest_at]);
            rest_at++;
        }
        const Item it = heap.top();
        heap.pop();
        const int32_t v = it.second;
        if (where[(size_t)v] != from) continue;
        const int64_t gn = gain_of(v);
        if (gn != it.first) { heap.emplace(gn, v); continue; }   // stale: re-queue with today's gain
        where[(size_t)v] = (uint8_t)to;
        P.pw[(size_t)from]--; P.pw[(size_t)to]++;
        need--;
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++) {
            const int32_t u = g.adj[(size_t)j];
            if (where[(size_t)u] == from) heap.emplace(gain_of(u), u);
        }
    }
}

static bool balance_round(const Level &g, std::vector<uint8_t> &where, Parts &P);
static void balance_tree(const Level &g, std::vector<uint8_t> &where, Parts &P) {
    // (a part that has to pass on more than it holds when its turn comes is served in the next round)
    for (int round = 0; round < P.k + 2; round++)
        if (balance_round(g, where, P)) return;
}
static bool balance_round(const Level &g, std::vector<uint8_t> &where, Parts &P) {
    const int k = P.k;
    bool exact = true;
    for (int p = 0; p < k; p++) exact &= P.pw[(size_t)p] == P.target[(size_t)p];
    if (exact) return true;
    std::vector<int64_t> conn((size_t)k * k, 0);
    for (int64_t v = 0; v < g.n; v++)
        for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1]; j++)
            conn[(size_t)where[(size_t)v] * k + where[(size_t)g.adj[(size_t)j]]] += g.ew[(size_t)j];
    // maximum spanning tree from part 0 (Prim; parts nothing links to are hung on with weight 0)
    std::vector<int> parent((size_t)k, -1), seq;
    std::vector<int64_t> key((size_t)k, -1);
    std::vector<uint8_t> in((size_t)k, 0);
    key[0] = 0;
    for (int it = 0; it < k; it++) {
        int b = -1;
        for (int p = 0; p < k; p++) if (!in[(size_t)p] && (b < 0 || key[(size_t)p] > key[(size_t)b])) b = p;
        in[(size_t)b] = 1;
        seq.push_back(b);
        for (int p = 0; p < k; p++)
            if (!in[(size_t)p] && conn[(size_t)b * k + p] > key[(size_t)p]) { key[(size_t)p] = conn[(size_t)b * k + p]; parent[(size_t)p] = b; }
    }
    for (int p = 1; p < k; p++) if (parent[(size_t)p] < 0) parent[(size_t)p] = 0;
    // subtree excess, children before parents (reverse of the order Prim added them in)
    std::vector<int64_t> excess((size_t)k);
    for (int p = 0; p < k; p++) excess[(size_t)p] = P.pw[(size_t)p] - P.target[(size_t)p];
    for (int i = k - 1; i > 0; i--) {
        const int p = seq[(size_t)i], q = parent[(size_t)p];
        const int64_t f = excess[(size_t)p];
        excess[(size_t)q] += f;
        if (f > 0) move_between(g, where, P, p, q, f);
        else if (f < 0) move_between(g, where, P, q, p, -f);
    }
    return false;
}

// (5b) refinement of the communication volume itself, sizes kept.  Factor lists give the exact objective:
// vol = sum over variables of |{ parts other than the variable's own that hold a co-member of one of its factors }|.
struct VolumeRefiner {
    int64_t nvar, nfactor;
    const nsk_factor *factor;
    const nsk_ftv *fmap;
    std::vector<int64_t> voff, vfac;     // variable -> factors (non-NOOP, every (variable, factor) pair once)
    std::vector<uint8_t> &where;
    int k;

    VolumeRefiner(int64_t nv, int64_t nf, const nsk_factor *f, const nsk_ftv *m, std::vector<uint8_t> &w, int parts)
        : nvar(nv), nfactor(nf), factor(f), fmap(m), where(w), k(parts) {
        voff.assign((size_t)nvar + 1, 0);
        auto each = [&](auto &&emit) {
            for (int64_t f2 = 0; f2 < nfactor; f2++) {
                if (factor[f2].factorFunction == -1) continue;
                const int64_t s = factor[f2].ftv_offset, e = s + factor[f2].arity;
                for (int64_t l = s; l < e; l++) {
                    bool dup = false;
                    for (int64_t l2 = s; l2 < l && l - s <= 16; l2++) dup |= fmap[l2].vid == fmap[l].vid;
                    if (!dup) emit(fmap[l].vid, f2);
                }
            }
        };
        each([&](int64_t v, int64_t) { voff[(size_t)v + 1]++; });
        for (int64_t v = 0; v < nvar; v++) voff[(size_t)v + 1] += voff[(size_t)v];
        vfac.resize((size_t)voff[(size_t)nvar]);
        std::vector<int64_t> fill(voff.begin(), voff.end() - 1);
        each([&](int64_t v, int64_t f2) { vfac[(size_t)fill[(size_t)v]++] = f2; });
    }
    // parts that must be sent variable u's value (bit set), with variable `mv` counted as living in part `mp`
    uint64_t need_of(int64_t u, int64_t mv, int mp) const {
        const int own = u == mv ? mp : where[(size_t)u];
        uint64_t need = 0;
        for (int64_t j = voff[(size_t)u]; j < voff[(size_t)u + 1]; j++) {
            const int64_t f = vfac[(size_t)j];
            const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
            for (int64_t l = s; l < e; l++) {
                const int64_t w = fmap[l].vid;
                need |= 1ull << (w == mv ? mp : where[(size_t)w]);
            }
        }
        return need & ~(1ull << own);
    }
    // the variables whose need-sets a move of v can change: v and its co-members
    void affected(int64_t v, std::vector<int64_t> &out) const {
        out.clear();
        out.push_back(v);
        for (int64_t j = voff[(size_t)v]; j < voff[(size_t)v + 1]; j++) {
            const int64_t f = vfac[(size_t)j];
            const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
            for (int64_t l = s; l < e; l++) if (fmap[l].vid != v) out.push_back(fmap[l].vid);
        }
        std::sort(out.begin() + 1, out.end());
        out.erase(std::unique(out.begin() + 1, out.end()), out.end());
    }
    // change of the volume when v moves to part q (negative = better); `cand` receives the parts v touches
    int64_t delta(int64_t v, int q, const std::vector<int64_t> &aff) const {
        int64_t d = 0;
        const int own = where[(size_t)v];
        for (int64_t u : aff)
            d += __builtin_popcountll(need_of(u, v, q)) - __builtin_popcountll(need_of(u, v, own));
        return d;
    }
    uint64_t touched_parts(int64_t v) const { return need_of(v, -1, 0) ; }
    int64_t total() const {
        int64_t vol = 0;
        for (int64_t v = 0; v < nvar; v++) vol += __builtin_popcountll(need_of(v, -1, 0));
        return vol;
    }
};

static void refine_volume(const Level &g, VolumeRefiner &R, Parts &P, int passes, int64_t slack, Rng &rng) {
    const int64_t n = R.nvar;
    const int k = P.k;
    std::vector<int32_t> perm;
    std::vector<int64_t> aff, aff2;
    std::vector<int8_t> hint;
    // a move into a part that is full is remembered per (from, to) pair and carried out when a move the other way
    // turns up with a combined improvement: sizes stay exact
    struct Pending { int64_t v, d; };
    std::vector<std::vector<Pending>> pend((size_t)k * k);
    for (int pass = 0; pass < passes; pass++) {
        // candidates: the boundary vertices of the variable graph (a variable with every co-member at home cannot gain)
        perm.clear();
        for (int64_t v = 0; v < n; v++) {
            const int own = R.where[(size_t)v];
            bool take = false;
            for (int64_t j = g.xadj[(size_t)v]; j < g.xadj[(size_t)v + 1] && !take; j++) take = R.where[(size_t)g.adj[(size_t)j]] != own;
            if (take) perm.push_back((int32_t)v);
        }
        for (int64_t i = (int64_t)perm.size() - 1; i > 0; i--) std::swap(perm[(size_t)i], perm[(size_t)rng.below((uint64_t)i + 1)]);
        for (auto &pl : pend) pl.clear();
        // every candidate's best destination against the state at the start of the pass (read-only: host threads) ...
        hint.assign(perm.size(), -1);
        nsk::parallel_for((int64_t)perm.size(), [&](int64_t b0, int64_t b1, int) {
            std::vector<int64_t> af;
            for (int64_t i = b0; i < b1; i++) {
                const int64_t v = perm[(size_t)i];
                uint64_t cand = R.touched_parts(v);
                if (!cand) continue;
                R.affected(v, af);
                int64_t bd = 0;
                while (cand) {
                    const int q = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    const int64_t d = R.delta(v, q, af);
                    if (d < bd) { bd = d; hint[(size_t)i] = (int8_t)q; }
                }
            }
        }, 2048);
        // ... then the promising ones one after the other, re-evaluated against the state as it is now
        int64_t moves = 0;
        for (size_t i = 0; i < perm.size(); i++) {
            if (hint[i] < 0) continue;
            const int64_t v = perm[i];
            const int own = R.where[(size_t)v];
            const int q = hint[i];
            if (q == own) continue;
            R.affected(v, aff);
            const int64_t d = R.delta(v, q, aff);
            if (d >= 0) continue;
            const bool room = P.pw[(size_t)q] + 1 <= P.target[(size_t)q] + slack;
            const int best = room ? q : -1, best_full = room ? -1 : q;
            const int64_t bd = d, bd_full = d;
            // a plain move: must improve, and keep both parts within `slack` of their sizes
            if (best >= 0 && bd < 0 && P.pw[(size_t)own] - 1 >= P.target[(size_t)own] - slack) {
                R.where[(size_t)v] = (uint8_t)best;
                P.pw[(size_t)own]--; P.pw[(size_t)best]++;
                moves++;
                continue;
            }
            if (best_full >= 0 && bd_full < 0) {
                // look for a partner waiting to go the other way
                auto &back = pend[(size_t)best_full * k + own];
                bool paired = false;
                for (int tries = 0; tries < 8 && !back.empty() && !paired; tries++) {
                    const Pending pb = back.back();
                    back.pop_back();
                    if (R.where[(size_t)pb.v] != best_full) { tries--; continue; }
                    // evaluate the pair exactly: move v, then the partner
                    R.where[(size_t)v] = (uint8_t)best_full;
                    R.affected(pb.v, aff2);
                    const int64_t d2 = R.delta(pb.v, own, aff2);
                    if (bd_full + d2 < 0) {
                        R.where[(size_t)pb.v] = (uint8_t)own;
                        moves += 2;
                        paired = true;
                    } else R.where[(size_t)v] = (uint8_t)own;
                }
                if (!paired) pend[(size_t)own * k + best_full].push_back(Pending{v, bd_full});
            }
        }
        if (!moves) break;
    }
}

}  // namespace

// order[new id] = old id (the shard formula's range g of the new ids is part g); part[old id] = g (may be NULL);
// stats (may be NULL): [0] levels, [1] coarsest vertices, [2] edge cut after uncoarsening, [3] volume before and
// [4] after the volume refinement (exact sizes both)
extern "C" int nsk_graph_partition(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge,
                                   const nsk_ftv *fmap, int nparts, uint64_t seed, int64_t *order, int64_t *part,
                                   int64_t *stats) {
    if (nvar < 0 || nfactor < 0 || nedge < 0 || !order || (nfactor && !factor) || (nedge && !fmap) || nparts < 1 || nparts > 64 ||
        nvar >= (int64_t)1 << 31) {
        nsk::set_error("nsk_graph_partition: bad argument (1 .. 64 parts, fewer than 2^31 variables)");
        return NSK_E_INVALID;
    }
    for (int64_t f = 0; f < nfactor; f++) {
        const int64_t s = factor[f].ftv_offset, e = s + factor[f].arity;
        if (factor[f].arity < 0 || s < 0 || e > nedge) { nsk::set_error("nsk_graph_partition: factor members outside fmap"); return NSK_E_INDEX; }
        for (int64_t l = s; l < e; l++)
            if (fmap[l].vid < 0 || fmap[l].vid >= nvar) { nsk::set_error("nsk_graph_partition: member outside variables"); return NSK_E_INDEX; }
    }
    Rng rng{seed * 0x2545F4914F6CDD1Dull + 0x1234567ull};
    const int k = nparts;
    Parts P;
    P.k = k;
    P.target.resize((size_t)k);
    for (int g = 0; g < k; g++) P.target[(size_t)g] = (int64_t)(((__int128)(g + 1) * nvar) / k) - (int64_t)(((__int128)g * nvar) / k);
    std::vector<uint8_t> where((size_t)nvar, 0);
    int64_t st[5] = {0, 0, 0, 0, 0};
    if (k > 1 && nvar > 0) {
        const bool verbose = getenv("NSK_VERBOSE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto tick = [&](const char *what) {
            const auto t1 = std::chrono::steady_clock::now();
            if (verbose) fprintf(stderr, "[nsk] partition: %-34s %7.2f s\n", what, std::chrono::duration<double>(t1 - t0).count());
            t0 = t1;
        };
        std::vector<Level> lv(1);
        build_fine(nvar, nfactor, factor, fmap, lv[0]);
        // Coarsening goes all the way down (a few thousand vertices); the partition starts at TWO depths: the deepest
        // level -- meshes: the cut of a very coarse graph is straight -- and a middle level of about nvar / 12 vertices --
        // banded graphs laced with long edges (config #5): the coarse levels' greedy refinement cannot repair an
        // arrangement, so the linear arrangement has to see the local structure itself (shuffled 1M-variable config-#5 graph,
        // 8 parts, volume against the generator's own ids: 1.09 / 1.09 / 1.03 / 0.98 when it starts at 2 000 / 8 000 / 30 000 /
        // 100 000 vertices).  The candidates are compared by edge cut at the middle level.
        const char *ct_env = nsk::diag_env("NSK_PART_COARSEN_TO");
        const int64_t deep_to = std::max<int64_t>(2000, 60 * (int64_t)k);
        const int64_t mid_to = ct_env ? atoll(ct_env) : std::min<int64_t>(200000, std::max<int64_t>(deep_to, nvar / 12));
        size_t mid = 0;
        while (lv.back().n > deep_to && lv.size() < 48) {
            if (lv.back().n > mid_to) mid = lv.size();
            Level c;
            if (!coarsen(lv.back(), c, std::max<int64_t>(1, (3 * lv[0].totw) / (2 * deep_to)), rng)) break;
            lv.push_back(std::move(c));
            if (verbose) fprintf(stderr, "[nsk] partition: level %d: %lld vertices, %lld edges\n", (int)lv.size() - 1, (long long)lv.back().n, (long long)lv.back().adj.size() / 2);
        }
        mid = std::min(mid, lv.size() - 1);
        st[0] = (int64_t)lv.size();
        st[1] = lv.back().n;
        tick("variable graph + coarsening");
        std::vector<int64_t> bound((size_t)k + 1, 0);
        for (int p = 0; p < k; p++) bound[(size_t)p + 1] = bound[(size_t)p] + P.target[(size_t)p];
        // (3) a partition of level `at`: a chain -- one linear arrangement cut into k ranges -- or a recursive bisection --
        // every half re-arranged along its own long direction
        auto initial = [&](size_t at, int variant, std::vector<uint8_t> &w) {
            const Level &g = lv[at];
            w.assign((size_t)g.n, 0);
            if (variant == 0) {
                std::vector<int32_t> ord;
                linear_order(g, ord, 20);
                int64_t cum = 0;
                int p = 0;
                for (int64_t i = 0; i < g.n; i++) {
                    const int32_t v = ord[(size_t)i];
                    const int64_t midw = cum + g.vw[(size_t)v] / 2;
                    while (p + 1 < k && midw >= bound[(size_t)p + 1]) p++;
                    w[(size_t)v] = (uint8_t)p;
                    cum += g.vw[(size_t)v];
                }
            } else {
                std::vector<int32_t> all((size_t)g.n);
                for (int64_t v = 0; v < g.n; v++) all[(size_t)v] = (int32_t)v;
                bisect(g, all, 0, k, bound, w);
            }
        };
        // (4) refine w at level `from`, then project and refine down to level `to`
        auto uncoarsen = [&](std::vector<uint8_t> &w, size_t from, size_t to, Parts &Q, Rng &r) {
            for (size_t li = from + 1; li-- > to;) {
                const Level &g = lv[li];
                if (li < from) {
                    std::vector<uint8_t> w2((size_t)g.n);
                    for (int64_t v = 0; v < g.n; v++) w2[(size_t)v] = w[(size_t)g.cmap[(size_t)v]];
                    w.swap(w2);
                }
                Q.pw.assign((size_t)k, 0);
                for (int64_t v = 0; v < g.n; v++) Q.pw[(size_t)w[(size_t)v]] += g.vw[(size_t)v];
                refine_cut(g, w, Q, li == 0 ? 0.005 : 0.03, li == 0 ? 6 : 10, r);
            }
        };
        // two finalists -- the better start of either depth, by edge cut at the middle level -- go all the way up; the edge
        // cut at the finest level decides (the cut at the middle level does not predict it across depths)
        {
            std::vector<uint8_t> fin[2];
            int64_t fin_cut[2] = {-1, -1};
            for (int c = 0; c < 4; c++) {
                const size_t at = c < 2 ? lv.size() - 1 : mid;
                if (c >= 2 && mid == lv.size() - 1) break;
                std::vector<uint8_t> w;
                initial(at, c & 1, w);
                Parts Q = P;
                Rng r2{rng.s + 77u * (unsigned)c};
                uncoarsen(w, at, mid, Q, r2);
                const int64_t cut = edge_cut(lv[mid], w);
                if (verbose) fprintf(stderr, "[nsk] partition: %s at %lld vertices: edge cut %lld at %lld vertices\n", (c & 1) ? "recursive bisection" : "chain",
                                     (long long)lv[at].n, (long long)cut, (long long)lv[mid].n);
                if (fin_cut[c >> 1] < 0 || cut < fin_cut[c >> 1]) { fin_cut[c >> 1] = cut; fin[c >> 1].swap(w); }
            }
            tick("initial partitions");
            int64_t best_cut = -1;
            for (int c = 0; c < 2; c++) {
                if (fin_cut[c] < 0) continue;
                std::vector<uint8_t> &w = fin[c];
                Parts Q = P;
                Rng r2{rng.s + 1234u * (unsigned)(c + 1)};
                if (mid > 0) {
                    std::vector<uint8_t> w2((size_t)lv[mid - 1].n);
                    for (int64_t v = 0; v < lv[mid - 1].n; v++) w2[(size_t)v] = w[(size_t)lv[mid - 1].cmap[(size_t)v]];
                    w.swap(w2);
                    uncoarsen(w, mid - 1, 0, Q, r2);
                } else {
                    Q.pw.assign((size_t)k, 0);
                    for (int64_t v = 0; v < nvar; v++) Q.pw[(size_t)w[(size_t)v]]++;
                }
                const int64_t cut = edge_cut(lv[0], w);
                if (verbose) fprintf(stderr, "[nsk] partition: finalist from the %s level: edge cut %lld\n", c ? "middle" : "deepest", (long long)cut);
                if (best_cut < 0 || cut < best_cut) { best_cut = cut; where.swap(w); P.pw = Q.pw; }
            }
        }
        for (size_t li = lv.size() - 1; li > 0; li--) { Level drop; std::swap(lv[li], drop); }
        st[2] = edge_cut(lv[0], where);
        tick("uncoarsening");
        balance_tree(lv[0], where, P);
        tick("exact sizes");
        VolumeRefiner R(nvar, nfactor, factor, fmap, where, k);
        st[3] = R.total();
        refine_volume(lv[0], R, P, 4, std::max<int64_t>(1, nvar / k / 500), rng);
        tick("volume refinement (slack)");
        // (back to exact sizes, then size-preserving exchanges only)
        balance_tree(lv[0], where, P);
        refine_volume(lv[0], R, P, 4, 0, rng);
        st[4] = R.total();
        tick("volume refinement (exact sizes)");
    }
    // the order: parts one after the other, ids ascending inside
    std::vector<int64_t> at((size_t)k + 1, 0);
    for (int64_t v = 0; v < nvar; v++) at[(size_t)where[(size_t)v] + 1]++;
    for (int g = 0; g < k; g++) {
        if (at[(size_t)g + 1] != P.target[(size_t)g] && k > 1) { nsk::set_error("nsk_graph_partition: internal error (part sizes)"); return NSK_E_INVALID; }
        at[(size_t)g + 1] += at[(size_t)g];
    }
    for (int64_t v = 0; v < nvar; v++) order[at[(size_t)where[(size_t)v]]++] = v;
    if (part) for (int64_t v = 0; v < nvar; v++) part[v] = where[(size_t)v];
    if (stats) memcpy(stats, st, sizeof(st));
    return NSK_OK;
}
