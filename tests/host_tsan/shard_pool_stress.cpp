// ThreadSanitizer stress of csrc/shard_pool.h -- the persistent host workers behind ft8gpu_decode_batch_multi[_dev]
// (api_multi.hip: run_shards).  The header has no HIP in it, so this file is the real class under g++ -fsanitize=thread.
// Round 3's review found a use-after-return between a worker's notification and a caller's stack latch in this class;
// this harness reproduces run_shards' calling pattern -- N posters, each posting M shards with a latch on its own stack,
// failing shards, zero-count shards, shard 0 on the caller -- thousands of times.
#include "shard_pool.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

std::atomic<long> g_jobs_run{0};

// what run_shards does (api_multi.hip), with the decode replaced by arithmetic on the shard's own slice
int run_shards_like(int poster, int round, int nshards, std::vector<long> &sink) {
    std::vector<int> rc((size_t)nshards, 0);
    std::vector<std::string> why((size_t)nshards);
    std::vector<int> count((size_t)nshards);
    for (int g = 0; g < nshards; ++g) count[g] = ((poster + round + g) % 5 == 0) ? 0 : 1 + (poster * 7 + round * 3 + g) % 9;   // some shards are empty
    auto work = [&](int g) {
        long acc = 0;
        for (int k = 0; k < count[g] * 50; ++k) acc += (long)k * (g + 1);
        sink[(size_t)g] = acc;                                   // every shard owns its slot: no sharing between shards
        if ((poster + round * 3 + g) % 11 == 0) { rc[g] = -1; why[g] = "shard failed on purpose"; }   // error text handed to the caller's thread
        g_jobs_run.fetch_add(1, std::memory_order_relaxed);
    };
    ShardPool &pool = ShardPool::instance();
    ShardPool::Latch latch;                                      // on THIS stack frame, as in run_shards
    for (int g = 1; g < nshards; ++g) {
        if (count[g] <= 0) continue;
        if (!pool.post([&work, g] { work(g); }, &latch)) work(g);
    }
    if (count[0] > 0) work(0);
    latch.wait();
    int failed = 0;
    for (int g = 0; g < nshards; ++g) if (rc[g]) failed += (int)why[g].size() > 0;
    return failed;
}

}  // namespace

int main(int argc, char **argv) {
    const int posters = argc > 1 ? atoi(argv[1]) : 6;
    const int rounds = argc > 2 ? atoi(argv[2]) : 400;
    const int shards = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<std::thread> th;
    std::atomic<long> failures{0};
    for (int p = 0; p < posters; ++p)
        th.emplace_back([&, p] {
            std::vector<long> sink((size_t)shards, 0);
            for (int r = 0; r < rounds; ++r) failures.fetch_add(run_shards_like(p, r, 1 + (p + r) % shards, sink));
        });
    for (auto &t : th) t.join();
    const int workers = ShardPool::instance().workers();
    printf("shard_pool_stress ok: %d posters x %d rounds, %ld jobs, %ld failing shards reported, %d workers\n", posters, rounds,
           g_jobs_run.load(), failures.load(), workers);
    // The pool aims at one waiting worker per queued job, i.e. about posters * (shards - 1) threads.  It can overshoot a little:
    // a worker that has just finished is not idle again until it re-takes the pool's mutex, and a poster that gets there first
    // starts another one.  The overshoot is bounded (extra workers make the next one less likely); a runaway would not be.
    const int expected = posters * (shards - 1);
    if (!(workers >= 1 && workers <= 4 * expected + 4)) { fprintf(stderr, "unexpected worker count %d (about %d expected)\n", workers, expected); return 1; }
    return g_jobs_run.load() > 0 ? 0 : 1;
}
