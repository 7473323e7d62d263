// The coalescing core (uzkge_amd/csrc/coalesce_core.hpp) under a fake backend, on the CPU: many threads, each proving "proofs" of
// R rounds through its own member, with random pauses (stragglers that get moved out), abandoned proofs, lanes that fail on their
// own, and members that come and go.  A proof here is a running hash of its lane's per-round inputs; a thread knows what its
// proof must come to and checks every one.  Built twice by tests/test_coalesce_core.py: plain, and with -fsanitize=thread.
//
// usage: coalesce_core_test [threads=12] [proofs_per_thread=300] [max_lanes=4]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "../../uzkge_amd/csrc/coalesce_core.hpp"

namespace {

constexpr int kRounds = 5;
uint64_t mix(uint64_t s, uint64_t x) {
    s ^= x + 0x9e3779b97f4a7c15ull + (s << 6) + (s >> 2);
    return s * 0xff51afd7ed558ccdull;
}

struct Arg {
    uint64_t in;
    uint64_t* out;
    bool fail;
};

struct Fake {
    struct MemberData { uint64_t own = 0; int id = 0; };
    struct CohortData { std::vector<uint64_t> state; MemberData* solo = nullptr; bool open = false; };
    std::atomic<uint64_t> rounds_run{0}, lanes_run{0}, moved{0}, widest{0}, opened{0}, closed{0};
    std::string last_error() { return "fake failure"; }

    int open(CohortData& cd, MemberData& leader, uint32_t lanes) {
        cd.state.assign(lanes, 0);
        cd.solo = lanes == 1 ? &leader : nullptr;
        cd.open = true;
        opened++;
        return 0;
    }
    void close(CohortData& cd) { if (cd.open) { cd.open = false; closed++; } }
    int move_out(CohortData& from, uint32_t lane, MemberData& to, CohortData& solo) {
        to.own = from.state[lane];
        solo.state.assign(1, to.own);
        solo.solo = &to;
        solo.open = true;
        opened++;
        moved++;
        return 0;
    }
    int run(CohortData& cd, int round, uint32_t lanes, void* const* args, const uint8_t* present, int* lane_rc, std::string* lane_msg) {
        if (lanes != cd.state.size()) return 9;
        thread_local std::mt19937 rng{std::random_device{}()};
        std::this_thread::sleep_for(std::chrono::microseconds(rng() % 120));
        uint32_t n = 0;
        for (uint32_t l = 0; l < lanes; ++l) {
            if (!present[l]) { if (args[l]) return 8; continue; }
            Arg& a = *static_cast<Arg*>(args[l]);
            ++n;
            if (a.fail) { lane_rc[l] = 7; lane_msg[l] = "lane asked to fail"; continue; }
            cd.state[l] = mix(round == 1 ? 0 : cd.state[l], a.in);
            *a.out = cd.state[l];
        }
        rounds_run++;
        lanes_run += n;
        uint64_t w = widest.load();
        while (n > w && !widest.compare_exchange_weak(w, n)) {}
        return 0;
    }
};

using Core = uzk::CoalesceCore<Fake>;

}  // namespace

int main(int argc, char** argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 12;
    const int proofs = argc > 2 ? std::atoi(argv[2]) : 300;
    const int max_lanes = argc > 3 ? std::atoi(argv[3]) : 4;
    Fake backend;
    Core core(backend, kRounds);
    core.configure((uint32_t)max_lanes, 100, 400, 2);
    std::atomic<uint64_t> good{0}, failed_on_purpose{0}, abandoned{0}, bad{0}, misuse{0};
    auto worker = [&](int t) {
        std::mt19937_64 rng(1234 + t);
        auto member = std::make_unique<Core::Member>();
        member->group = 1;
        member->data.id = t;
        core.add(member.get());
        Core::Key key;
        key.group = 1;
        for (int p = 0; p < proofs; ++p) {
            key.blob = (rng() % 8 == 0) ? "circuit-b" : "circuit-a";            // mostly one circuit: cohorts form
            const bool will_fail = rng() % 23 == 0, will_abandon = rng() % 29 == 0, dawdle = rng() % 17 == 0;
            const int fail_round = 1 + (int)(rng() % kRounds), abandon_after = 1 + (int)(rng() % (kRounds - 1));
            uint64_t expect = 0, got = 0;
            bool over = false;
            for (int r = 1; r <= kRounds && !over; ++r) {
                if (dawdle && r > 1 && rng() % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(900));     // longer than straggler_wait: moved out
                Arg a{rng(), &got, will_fail && r == fail_round};
                std::string msg;
                const int rc = core.enter(member.get(), r, r == 1 ? &key : nullptr, &a, &msg);
                if (a.fail) {
                    if (rc != 7 || msg != "lane asked to fail") bad++;
                    else failed_on_purpose++;
                    // the proof is gone: its next round must be refused
                    Arg b{1, &got, false};
                    if (r < kRounds && core.enter(member.get(), r + 1, nullptr, &b, &msg) != Core::kErrParameter) bad++;
                    else misuse++;
                    over = true;
                    break;
                }
                if (rc != 0) { bad++; over = true; break; }
                expect = mix(r == 1 ? 0 : expect, a.in);
                if (got != expect) { bad++; over = true; break; }
                if (will_abandon && r == abandon_after) { abandoned++; over = true; }      // the next round 1 drops this proof
            }
            if (!over) good++;
            if (rng() % 61 == 0) {              // the prover goes away and a new one comes
                if (!core.remove(member.get())) bad++;
                member = std::make_unique<Core::Member>();
                member->group = 1;
                member->data.id = t;
                core.add(member.get());
            }
        }
        if (!core.remove(member.get())) bad++;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
    std::printf("{\"threads\": %d, \"proofs\": %llu, \"failed_on_purpose\": %llu, \"abandoned\": %llu, \"refused_out_of_order\": %llu, \"bad\": %llu, "
                "\"rounds_run\": %llu, \"lanes_run\": %llu, \"widest_round\": %llu, \"moved_out\": %llu, \"cohorts_opened\": %llu, \"cohorts_closed\": %llu}\n",
                threads, (unsigned long long)good.load(), (unsigned long long)failed_on_purpose.load(), (unsigned long long)abandoned.load(),
                (unsigned long long)misuse.load(), (unsigned long long)bad.load(), (unsigned long long)backend.rounds_run.load(), (unsigned long long)backend.lanes_run.load(),
                (unsigned long long)backend.widest.load(), (unsigned long long)backend.moved.load(), (unsigned long long)backend.opened.load(),
                (unsigned long long)backend.closed.load());
    const bool ok = bad.load() == 0 && good.load() > 0 && backend.opened.load() == backend.closed.load();
    std::printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
