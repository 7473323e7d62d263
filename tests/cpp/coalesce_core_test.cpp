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

// `coalesce_core_test policy`: the scheduling statements the GPU suite used to make with wall-clock bounds on a foreign host
// (VERDICT r5), made here against the stub backend and as ORDER of events wherever possible:
//   1. a prover that is alone never waits for company, however long the gathering wait is;
//   2. two provers: the cohort leaves the moment it is full (no gathering wait is sat out);
//   3. one of them stays away between rounds 2 and 3: the other's round 3 returns BEFORE the dawdler comes back (it was moved
//      out after straggler_wait), the shared rounds were two wide, exactly one lane was moved out, both proofs are right.
static int policy() {
    using Ms = std::chrono::milliseconds;
    Fake backend;
    Core core(backend, kRounds);
    core.configure(2, 5000000, 20000, 1);                       // gather for up to 5 s, wait 20 ms for a straggler, one team
    Core::Key key; key.group = 1; key.blob = "circuit";
    int bad = 0;
    auto prove_round = [&](Core::Member& m, int r, uint64_t in, uint64_t* got) {
        Arg a{in, got, false};
        std::string msg;
        return core.enter(&m, r, r == 1 ? &key : nullptr, &a, &msg);
    };
    {   // 1.
        Core::Member m; m.group = 1; core.add(&m);
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t got = 0, expect = 0;
        for (int r = 1; r <= kRounds; ++r) { if (prove_round(m, r, 100 + r, &got) != 0) ++bad; expect = mix(r == 1 ? 0 : expect, 100 + r); }
        if (got != expect) ++bad;
        if (std::chrono::steady_clock::now() - t0 > Ms(2500)) { std::printf("policy 1: a lone prover waited\n"); ++bad; }      // half the gathering wait
        if (core.stats().widest != 1) ++bad;
        if (!core.remove(&m)) ++bad;
    }
    core.reset_stats();
    {   // 2. and 3.
        Core::Member m[2]; for (auto& x : m) { x.group = 1; core.add(&x); }
        std::atomic<int> dawdler_back{0}, on_time_round3_done_before_dawdler{0}, errs{0};
        std::atomic<int> at_start{0};
        auto worker = [&](int t) {
            uint64_t got = 0, expect = 0;
            at_start.fetch_add(1);
            while (at_start.load() < 2) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 1; r <= kRounds; ++r) {
                if (t == 1 && r == 3) { std::this_thread::sleep_for(Ms(1500)); dawdler_back.store(1); }
                const uint64_t in = 1000 * (t + 1) + r;
                if (prove_round(m[t], r, in, &got) != 0) errs++;
                expect = mix(r == 1 ? 0 : expect, in);
                if (got != expect) errs++;
                if (r == 1 && std::chrono::steady_clock::now() - t0 > Ms(2500)) { std::printf("policy 2: a full cohort sat out the gathering wait\n"); errs++; }
                if (t == 0 && r == 3 && !dawdler_back.load()) on_time_round3_done_before_dawdler.store(1);
            }
        };
        std::thread a(worker, 0), b(worker, 1);
        a.join(); b.join();
        const Core::Stats st = core.stats();
        if (errs.load()) { std::printf("policy 2/3: %d wrong proofs or late rounds\n", errs.load()); ++bad; }
        if (!on_time_round3_done_before_dawdler.load()) { std::printf("policy 3: the prover that was on time waited for the dawdler\n"); ++bad; }
        if (st.widest != 2 || st.moved_out != 1) { std::printf("policy 3: widest %llu moved_out %llu\n", (unsigned long long)st.widest, (unsigned long long)st.moved_out); ++bad; }
        for (auto& x : m) if (!core.remove(&x)) ++bad;
    }
    if (backend.opened.load() != backend.closed.load()) ++bad;
    std::printf("{\"policy\": true, \"bad\": %d}\n%s\n", bad, bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "policy") return policy();
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
