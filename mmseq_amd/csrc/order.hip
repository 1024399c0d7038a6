// order.hip -- a transcript order derived from the hit graph, for callers that pass none.
//
// The reference numbers transcripts in first-seen order (src/mmseq.cpp:399-408): the isoforms of a gene -- the transcripts that share
// reads -- end up anywhere in the index range, every row spans the whole range, and only the CSR-tile kernel applies (29 x slower at
// config 3).  A caller that knows the genes passes tx_order (the CLI does).  For one that does not, the library looks at which
// transcripts occur together in rows and lays them out so that they are neighbours:
//   device   a content-hash sample of the rows (a pure function of the SET of rows: the order does not depend on how the caller
//            delivered them) emits, per row, the edges between its smallest transcript and its other transcripts, both directions;
//            radix sort + unique leave the adjacency lists of the co-occurrence graph, sorted (rocPRIM)
//   host     a breadth-first level structure per connected component from a pseudo-peripheral vertex (George & Liu), hubs left out,
//            every level ordered by (first parent, number of parents, degree, index): the Cuthill-McKee idea with the within-level
//            order a banded graph needs -- rows of transcripts that lie within +-64 of each other in SOME order come back within an
//            LDS window of each other, and the transcripts of a gene family next to each other
// The result is used exactly like a caller's tx_order; nothing crossing the ABI changes numbering.
#include <algorithm>
#include <cstdint>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include "mmg_launch.h"

namespace mmg {

constexpr uint32_t ORDER_MAX_ROW = 255;      // longer rows never fit an LDS window whatever the order, and would link everything to everything
constexpr uint64_t ORDER_SAMPLE_HITS = 1ull << 27; // hits of the sampled rows (two edges per hit: 2 GB of keys to sort at most)

__device__ __forceinline__ uint64_t order_mix(uint32_t c)
{
    uint64_t x = ((uint64_t)c + 1) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    return x * 0xBF58476D1CE4E5B9ull;
}

// edges of row r: 2 x (hits other than its smallest transcript) if the row is in the sample, else 0.  EMIT: write them.
// label: NULL = the vertices of the graph are the column ids; else label[c] = the vertex (group) of column c -- the graph of the GROUPS
// that share rows (a row inside one group contributes nothing)
template <typename IdxT, bool EMIT>
__global__ __launch_bounds__(256) void k_order_edges(uint64_t m, const IdxT *__restrict__ rp, const uint32_t *__restrict__ col, uint64_t mask,
                                                     const uint32_t *__restrict__ label,
                                                     uint32_t *__restrict__ cnt, const uint64_t *__restrict__ off, uint64_t *__restrict__ keys)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const uint64_t b = rp[r], e = rp[r + 1], L = e - b;
    uint32_t n = 0;
    if (L >= 2 && L <= ORDER_MAX_ROW) {
        uint64_t h = 0;
        uint32_t lo = 0xffffffffu;
        for (uint64_t j = b; j < e; ++j) { const uint32_t c = col[j]; h += order_mix(c); lo = min(lo, label ? label[c] : c); }   // commutative: a row is a set
        h ^= h >> 32;
        if ((h & mask) == 0) {
            uint64_t o = EMIT ? off[r] : 0;
            for (uint64_t j = b; j < e; ++j) {
                const uint32_t c = label ? label[col[j]] : col[j];
                if (c == lo) continue;
                if (EMIT) { keys[o++] = ((uint64_t)lo << 32) | c; keys[o++] = ((uint64_t)c << 32) | lo; }
                n += 2;
            }
        }
    }
    if (!EMIT) cnt[r] = n;
}

// Sorted, duplicate-free directed edges (u << 32 | v, both directions of every undirected edge) of the co-occurrence graph of a device
// CSR whose column ids are the CALLER's transcript ids (d_label NULL), or of the groups d_label[column id] (device array).  edges: host
// vector.  Returns hipSuccess with an empty vector when there is nothing to link.
hipError_t order_cooccurrence_edges(bool idx64, uint64_t m, uint64_t nnz, const void *d_rp, const uint32_t *d_col, std::vector<uint64_t> &edges, hipStream_t s,
                                    const uint32_t *d_label)
{
    edges.clear();
    if (m == 0 || nnz == 0) return hipSuccess;
    uint64_t mask = 0;
    while ((nnz >> __builtin_popcountll(mask)) > ORDER_SAMPLE_HITS) mask = (mask << 1) | 1;
    uint32_t *d_cnt = nullptr;
    uint64_t *d_off = nullptr, *d_keys = nullptr, *d_sorted = nullptr, *d_nuniq = nullptr;
    void *d_tmp = nullptr;
    auto done = [&](hipError_t rc) {
        for (void *x : {(void *)d_cnt, (void *)d_off, (void *)d_keys, (void *)d_sorted, (void *)d_nuniq, d_tmp}) if (x) (void)hipFree(x);
        return rc;
    };
    hipError_t e;
#define O_TRY(expr) do { e = (expr); if (e != hipSuccess) return done(e); } while (0)
    O_TRY(hipMalloc((void **)&d_cnt, m * 4));
    O_TRY(hipMalloc((void **)&d_off, (m + 1) * 8));
    const unsigned g = (unsigned)((m + 255) / 256);
    if (idx64) hipLaunchKernelGGL((k_order_edges<uint64_t, false>), dim3(g), dim3(256), 0, s, m, (const uint64_t *)d_rp, d_col, mask, d_label, d_cnt, (const uint64_t *)nullptr, (uint64_t *)nullptr);
    else hipLaunchKernelGGL((k_order_edges<uint32_t, false>), dim3(g), dim3(256), 0, s, m, (const uint32_t *)d_rp, d_col, mask, d_label, d_cnt, (const uint64_t *)nullptr, (uint64_t *)nullptr);
    O_TRY(hipGetLastError());
    O_TRY(layout_scan_lens(m, d_cnt, d_off, s));
    uint64_t E = 0;
    O_TRY(hipMemcpy(&E, d_off + m, 8, hipMemcpyDeviceToHost));
    if (E == 0) return done(hipSuccess);
    O_TRY(hipMalloc((void **)&d_keys, E * 8));
    O_TRY(hipMalloc((void **)&d_sorted, E * 8));
    if (idx64) hipLaunchKernelGGL((k_order_edges<uint64_t, true>), dim3(g), dim3(256), 0, s, m, (const uint64_t *)d_rp, d_col, mask, d_label, (uint32_t *)nullptr, (const uint64_t *)d_off, d_keys);
    else hipLaunchKernelGGL((k_order_edges<uint32_t, true>), dim3(g), dim3(256), 0, s, m, (const uint32_t *)d_rp, d_col, mask, d_label, (uint32_t *)nullptr, (const uint64_t *)d_off, d_keys);
    O_TRY(hipGetLastError());
    size_t tmp = 0;
    O_TRY(rocprim::radix_sort_keys(nullptr, tmp, d_keys, d_sorted, E, 0, 64, s));
    O_TRY(hipMalloc(&d_tmp, tmp ? tmp : 8));
    O_TRY(rocprim::radix_sort_keys(d_tmp, tmp, d_keys, d_sorted, E, 0, 64, s));
    O_TRY(hipStreamSynchronize(s));
    (void)hipFree(d_tmp); d_tmp = nullptr;
    O_TRY(hipMalloc((void **)&d_nuniq, 8));
    tmp = 0;
    O_TRY(rocprim::unique(nullptr, tmp, d_sorted, d_keys, d_nuniq, E, rocprim::equal_to<uint64_t>(), s));
    O_TRY(hipMalloc(&d_tmp, tmp ? tmp : 8));
    O_TRY(rocprim::unique(d_tmp, tmp, d_sorted, d_keys, d_nuniq, E, rocprim::equal_to<uint64_t>(), s));
    uint64_t nu = 0;
    O_TRY(hipMemcpyAsync(&nu, d_nuniq, 8, hipMemcpyDeviceToHost, s));
    O_TRY(hipStreamSynchronize(s));
    edges.resize(nu);
    O_TRY(hipMemcpy(edges.data(), d_keys, nu * 8, hipMemcpyDeviceToHost));
#undef O_TRY
    return done(hipSuccess);
}

// Position of every transcript in the derived order: pos[t] for t < n (a permutation of 0..n-1).  edges: sorted unique u << 32 | v.
//   hubs       a transcript that shares rows with far more others than is usual (more than max(hub_floor, 8 x the median degree of the linked
//              transcripts): a repeat element, a ubiquitous paralogue) would tie every gene family into one component whose level
//              structure fans out from it; it is left out of the traversal and placed behind everything (its rows get a far hit).
//              hub_floor: 256 for a graph of transcripts (the isoforms and window neighbours of a transcript are a hundred-odd), 32 for a
//              graph of GROUPS (a gene that shares reads with more than 32 other genes is no family member: spec version 8)
//   a level    is ordered by (position of the FIRST parent in the level before it, number of parents: most first, degree, index): the
//              children of one parent stay together -- gene families, trees -- and in a band every vertex has its own first parent
//              (the vertex 64 before it), which puts the level in the order of the band
void order_from_edges(uint32_t n, const std::vector<uint64_t> &edges, std::vector<uint32_t> &pos, uint32_t hub_floor)
{
    std::vector<uint64_t> ptr(n + 1, 0);
    for (uint64_t k : edges) ptr[(k >> 32) + 1]++;
    for (uint32_t v = 0; v < n; ++v) ptr[v + 1] += ptr[v];
    auto deg = [&](uint32_t v) { return (uint32_t)(ptr[v + 1] - ptr[v]); };
    auto nb = [&](uint64_t i) { return (uint32_t)edges[i]; };          // edges are sorted by u: the adjacency list of u is a slice
    uint32_t hub_deg = hub_floor;
    {
        std::vector<uint32_t> dd;
        for (uint32_t v = 0; v < n; ++v) if (deg(v)) dd.push_back(deg(v));
        if (!dd.empty()) {
            std::nth_element(dd.begin(), dd.begin() + dd.size() / 2, dd.end());
            hub_deg = std::max<uint32_t>(hub_floor, 8u * dd[dd.size() / 2]);
        }
    }
    std::vector<uint32_t> stamp(n, 0), level, next, order;
    std::vector<uint8_t> placed(n, 0);                                  // 1 placed, 2 hub (not traversed)
    for (uint32_t v = 0; v < n; ++v) if (deg(v) > hub_deg) placed[v] = 2;
    order.reserve(n);
    uint32_t epoch = 0;
    // levels of a BFS from `root` inside the unplaced part of its component; returns the number of levels and the min-degree vertex of the last
    auto eccentricity = [&](uint32_t root, uint32_t &far) {
        ++epoch;
        level.assign(1, root);
        stamp[root] = epoch;
        uint32_t levels = 0;
        while (!level.empty()) {
            ++levels;
            far = level[0];
            for (uint32_t v : level) if (deg(v) < deg(far) || (deg(v) == deg(far) && v < far)) far = v;
            next.clear();
            for (uint32_t u : level)
                for (uint64_t i = ptr[u]; i < ptr[u + 1]; ++i) { const uint32_t v = nb(i); if (stamp[v] != epoch && !placed[v]) { stamp[v] = epoch; next.push_back(v); } }
            level.swap(next);
        }
        return levels;
    };
    std::vector<uint32_t> by_degree(n);
    for (uint32_t v = 0; v < n; ++v) by_degree[v] = v;
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](uint32_t a, uint32_t b) { return deg(a) < deg(b); });
    std::vector<uint32_t> parents(n, 0), first_parent(n, 0);            // parents of v in the level before it, and the first of them (its index in that level)
    for (uint32_t cand : by_degree) {
        if (placed[cand] || deg(cand) == 0) continue;
        // pseudo-peripheral start: walk to a farthest vertex while the eccentricity grows
        uint32_t root = cand, far = cand, ecc = eccentricity(root, far);
        for (int rounds = 0; rounds < 8; ++rounds) {
            uint32_t far2 = far;
            const uint32_t e2 = eccentricity(far, far2);
            if (e2 <= ecc) break;
            root = far; ecc = e2; far = far2;
        }
        level.assign(1, root);
        placed[root] = 1;
        while (!level.empty()) {
            for (uint32_t u : level) order.push_back(u);
            next.clear();
            for (uint32_t iu = 0; iu < (uint32_t)level.size(); ++iu) {
                const uint32_t u = level[iu];
                for (uint64_t i = ptr[u]; i < ptr[u + 1]; ++i) {
                    const uint32_t v = nb(i);
                    if (placed[v]) continue;
                    if (parents[v]++ == 0) { first_parent[v] = iu; next.push_back(v); }
                }
            }
            std::sort(next.begin(), next.end(), [&](uint32_t a, uint32_t b) {
                if (first_parent[a] != first_parent[b]) return first_parent[a] < first_parent[b];
                if (parents[a] != parents[b]) return parents[a] > parents[b];
                if (deg(a) != deg(b)) return deg(a) < deg(b);
                return a < b;
            });
            for (uint32_t v : next) placed[v] = 1;
            level.swap(next);
        }
    }
    for (uint32_t v = 0; v < n; ++v) if (placed[v] != 1) order.push_back(v);  // hubs, and transcripts that share no row with another: anywhere
    pos.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) pos[order[i]] = i;
}

} // namespace mmg
