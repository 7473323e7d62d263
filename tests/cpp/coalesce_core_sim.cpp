// Do callers with unrelated phases converge on full teams?  The coalescing core under a backend that only takes time the way
// the GPU does: a round of k lanes costs base + per_lane * k, at most `turns` rounds run at once.  Threads prove back to back,
// no failures; after a warm-up the statistics are reset and the steady state is measured.
// usage: coalesce_core_sim [threads=16] [max_lanes=8] [groups=4] [gather_us=1000] [seconds=1.0]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "../../uzkge_amd/csrc/coalesce_core.hpp"

namespace {
struct Sim {
    struct MemberData { int id = 0; };
    struct CohortData { uint32_t lanes = 0; };
    std::mutex mu;
    std::condition_variable cv;
    uint32_t running = 0, turns = 4;
    std::string last_error() { return ""; }
    int open(CohortData& cd, MemberData&, uint32_t lanes) { cd.lanes = lanes; return 0; }
    void close(CohortData&) {}
    int move_out(CohortData&, uint32_t, MemberData&, CohortData& solo) { solo.lanes = 1; return 0; }
    int run(CohortData&, int, uint32_t lanes, void* const*, const uint8_t* present, int*, std::string*) {
        { std::unique_lock<std::mutex> lk(mu); while (running >= turns) cv.wait(lk); ++running; }
        uint32_t k = 0;
        for (uint32_t l = 0; l < lanes; ++l) k += present[l];
        std::this_thread::sleep_for(std::chrono::microseconds(300 + 100 * k));
        { std::lock_guard<std::mutex> lk(mu); --running; }
        cv.notify_one();
        return 0;
    }
};
using Core = uzk::CoalesceCore<Sim>;
}  // namespace

int main(int argc, char** argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 16;
    const uint32_t max_lanes = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 8, groups = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 4;
    const uint32_t gather = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 1000;
    const double seconds = argc > 5 ? std::atof(argv[5]) : 1.0;
    Sim backend;
    backend.turns = groups;
    Core core(backend, 5);
    core.configure(max_lanes, gather, 20000, groups);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> proofs{0};
    auto worker = [&](int t) {
        Core::Member m;
        m.group = 1;
        m.data.id = t;
        core.add(&m);
        Core::Key key;
        key.group = 1;
        key.blob = "c";
        std::this_thread::sleep_for(std::chrono::microseconds(137 * t));       // unrelated phases to begin with
        int arg = 0;
        std::string msg;
        while (!stop.load()) {
            for (int r = 1; r <= 5; ++r) core.enter(&m, r, r == 1 ? &key : nullptr, &arg, &msg);
            proofs++;
        }
        core.remove(&m);
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    core.reset_stats();
    const uint64_t p0 = proofs.load();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const auto st = core.stats();
    const uint64_t p1 = proofs.load();
    stop.store(true);
    for (auto& th : pool) th.join();
    const uint32_t teams = std::max<uint32_t>(groups, ((uint32_t)threads + max_lanes - 1) / max_lanes);
    const double ideal = (double)threads / std::min<uint32_t>(teams, (uint32_t)threads);
    std::printf("{\"threads\": %d, \"proofs_per_s\": %.0f, \"lanes_per_round\": %.2f, \"ideal_lanes_per_round\": %.2f, \"gather_us_per_cohort\": %.0f, \"gap_us\": %.0f, \"moved_out\": %llu}\n",
                threads, (double)(p1 - p0) / seconds, st.rounds ? (double)st.lanes / st.rounds : 0.0, ideal, st.cohorts ? (double)st.gather_us / st.cohorts : 0.0,
                st.gaps ? (double)st.gap_us / st.gaps : 0.0, (unsigned long long)st.moved_out);
    return 0;
}
