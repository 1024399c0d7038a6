// Native (C++, no Python, no torch) test of the multi-GPU entry points of libmmgibbs: a group of the visible devices, the
// read-shard loop with its RCCL int32 all-reduce, chains mode with the pooled fp64 moments, and mmg_shard_bounds.
// On a 1-GPU box the group has one device: RCCL is still initialised (ncclCommInitAll) and the same code path runs.
// usage: test_group [n_devices]      exit code 0 = all checks passed
#include "../../include/mmgibbs.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(expr) do { int _rc = (expr); if (_rc) { fprintf(stderr, "FAILED %s: %s\n", #expr, mmg_last_error()); return 1; } } while (0)
#define REQUIRE(cond) do { if (!(cond)) { fprintf(stderr, "FAILED requirement %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    int ndev = 0;
    CHECK(mmg_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    int G = argc > 1 ? atoi(argv[1]) : ndev;
    if (G > ndev) G = ndev;
    std::vector<int> devs(G);
    for (int i = 0; i < G; ++i) devs[i] = i;

    // ---- the reference chain: one device, the whole problem
    const uint64_t R = 200000;
    const uint32_t T = 3000;
    mmg_synth_desc sd;
    memset(&sd, 0, sizeof sd);
    sd.seed = 1234; sd.rows = R; sd.row0 = 0; sd.n = T; sd.avg_hits = 6.0; sd.uniform = 0; sd.sorted = 1; sd.mapped_reads = R;
    mmg_problem *full = nullptr;
    CHECK(mmg_problem_create_synthetic(&sd, 0, &full));
    mmg_problem_info inf;
    CHECK(mmg_problem_info_get(full, &inf));
    std::vector<uint64_t> rp(inf.m + 1);
    std::vector<uint32_t> ci(inf.nnz);
    std::vector<double> l(T), mu0(T);
    CHECK(mmg_problem_download(full, rp.data(), ci.data(), nullptr));
    CHECK(mmg_problem_get_l(full, l.data()));
    CHECK(mmg_problem_start_values(full, mu0.data(), nullptr));
    mmg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.alpha = cfg.beta = 0.1; cfg.seed = 99; cfg.n_chains = 1; cfg.chain_base = 0; cfg.gibbs_iter = 32; cfg.trace_len = 32; cfg.keep_trace = 1;
    mmg_sampler *ref = nullptr;
    CHECK(mmg_sampler_create(full, &cfg, mu0.data(), &ref));
    CHECK(mmg_sampler_run(ref, 32));
    std::vector<double> tr_ref((size_t)T * 32), sl_ref(T), sl2_ref(T);
    int64_t ns_ref = 0;
    CHECK(mmg_sampler_get_trace(ref, 0, tr_ref.data()));
    CHECK(mmg_sampler_get_moments(ref, 0, sl_ref.data(), sl2_ref.data(), &ns_ref));

    // ---- shard bounds: even boundaries, balanced by hits, cover every row once
    std::vector<uint64_t> bounds(G + 1);
    CHECK(mmg_shard_bounds(rp.data(), inf.m, G, bounds.data()));
    REQUIRE(bounds[0] == 0 && bounds[G] == inf.m);
    for (int i = 0; i < G; ++i) {
        REQUIRE(bounds[i] <= bounds[i + 1] && (bounds[i] % 2 == 0));
        const double share = (double)(rp[bounds[i + 1]] - rp[bounds[i]]) / (double)inf.nnz;
        REQUIRE(share > 0.8 / G && share < 1.2 / G);
    }

    // ---- read-shard mode over the group: stored rows cut at the bounds, kept as they are, one chain
    mmg_group *grp = nullptr;
    CHECK(mmg_group_create(devs.data(), G, &grp));
    int gs = 0;
    CHECK(mmg_group_size(grp, &gs));
    REQUIRE(gs == G);
    std::vector<mmg_problem *> shard(G);
    std::vector<mmg_sampler *> smp(G);
    for (int i = 0; i < G; ++i) {
        std::vector<uint64_t> srp(bounds[i + 1] - bounds[i] + 1);
        for (size_t r = 0; r < srp.size(); ++r) srp[r] = rp[bounds[i] + r] - rp[bounds[i]];
        mmg_problem_desc pd;
        memset(&pd, 0, sizeof pd);
        pd.m = bounds[i + 1] - bounds[i]; pd.n = T; pd.row_ptr = srp.data(); pd.col_idx = ci.data() + rp[bounds[i]]; pd.k = nullptr; pd.l = l.data();
        pd.row_id_base = bounds[i]; pd.layout = MMG_LAYOUT_KEEP_ROWS; pd.tx_order = nullptr;
        CHECK(mmg_problem_create(&pd, devs[i], &shard[i]));
        CHECK(mmg_sampler_create(shard[i], &cfg, mu0.data(), &smp[i]));
    }
    CHECK(mmg_group_run_sharded(grp, smp.data(), 32));
    double enq_sharded = 0.0;
    CHECK(mmg_group_enqueue_us(grp, &enq_sharded));
    for (int i = 0; i < G; ++i) {
        std::vector<double> tr((size_t)T * 32);
        CHECK(mmg_sampler_get_trace(smp[i], 0, tr.data()));
        REQUIRE(memcmp(tr.data(), tr_ref.data(), tr.size() * sizeof(double)) == 0); // bit-identical to the unsharded chain, on every device
    }
    for (int i = 0; i < G; ++i) { mmg_sampler_destroy(smp[i]); mmg_problem_destroy(shard[i]); }

    // ---- chains mode: chain i on device i over the full problem; pooled moments = sum of the single-chain moments
    std::vector<mmg_problem *> rep(G);
    std::vector<double> want_sl(T, 0.0), want_sl2(T, 0.0);
    for (int i = 0; i < G; ++i) {
        if (i == 0) rep[i] = full;
        else {
            mmg_problem_desc pd;
            memset(&pd, 0, sizeof pd);
            pd.m = inf.m; pd.n = T; pd.row_ptr = rp.data(); pd.col_idx = ci.data(); pd.l = l.data(); pd.layout = MMG_LAYOUT_KEEP_ROWS;
            CHECK(mmg_problem_create(&pd, devs[i], &rep[i]));
        }
        mmg_config c2 = cfg;
        c2.chain_base = i;
        CHECK(mmg_sampler_create(rep[i], &c2, mu0.data(), &smp[i]));
    }
    CHECK(mmg_group_run_chains(grp, smp.data(), 32));
    double enq_chains = 0.0;
    CHECK(mmg_group_enqueue_us(grp, &enq_chains));
    for (int i = 0; i < G; ++i) {
        std::vector<double> a(T), b(T);
        int64_t ns = 0;
        CHECK(mmg_sampler_get_moments(smp[i], 0, a.data(), b.data(), &ns));
        REQUIRE(ns == 32);
        if (i == 0) REQUIRE(memcmp(a.data(), sl_ref.data(), T * sizeof(double)) == 0); // chain 0 on device 0 is the reference chain
        for (uint32_t t = 0; t < T; ++t) { want_sl[t] += a[t]; want_sl2[t] += b[t]; }
    }
    std::vector<double> got_sl(T), got_sl2(T);
    int64_t ns_all = 0;
    CHECK(mmg_group_pool_moments(grp, smp.data(), got_sl.data(), got_sl2.data(), &ns_all));
    REQUIRE(ns_all == 32 * (int64_t)G);
    {   // pooling is idempotent: the samplers keep their own moments
        std::vector<double> again(T), again2(T);
        int64_t ns2 = 0;
        CHECK(mmg_group_pool_moments(grp, smp.data(), again.data(), again2.data(), &ns2));
        REQUIRE(ns2 == ns_all && memcmp(again.data(), got_sl.data(), T * sizeof(double)) == 0 && memcmp(again2.data(), got_sl2.data(), T * sizeof(double)) == 0);
    }
    for (uint32_t t = 0; t < T; ++t) {
        const double d1 = got_sl[t] - want_sl[t], d2 = got_sl2[t] - want_sl2[t];
        REQUIRE(d1 * d1 <= 1e-20 * want_sl[t] * want_sl[t] && d2 * d2 <= 1e-20 * want_sl2[t] * want_sl2[t]); // fp64 sums in RCCL's order
    }
    for (int i = 0; i < G; ++i) { mmg_sampler_destroy(smp[i]); if (i) mmg_problem_destroy(rep[i]); }
    mmg_sampler_destroy(ref);
    mmg_problem_destroy(full);
    mmg_group_destroy(grp);
    printf("test_group OK: %d device(s), read-shard chain bit-identical to the unsharded chain, pooled moments match; host enqueue per "
           "device-iteration: %.1f us (read-shard: K1, all-reduce, K2), %.1f us (chains: K1, K2)\n", G, enq_sharded, enq_chains);
    return 0;
}
