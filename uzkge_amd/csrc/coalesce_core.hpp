// Gathering concurrent one-proof callers into lockstep launches: the synchronisation core, free of HIP so that it also builds
// with plain g++ under ThreadSanitizer (tests/cpp/coalesce_core_test.cpp drives it with a fake backend).
//
// The reference proves ONE proof per call from whatever thread the application runs (prover_with_lagrange,
// uzkge/src/plonk/prover.rs:88-100; zshuffle's SDK, shuffle/src/sdk.rs:196-214), and at n = 2^14 one proof leaves most of an
// MI355X idle.  The library therefore merges callers that stand at the same round of proofs over the same circuit:
//
//   * the provers of a kind (same size, same device) are dealt into `groups` TEAMS -- a few launch sequences side by side fill the
//     chip better than one wide one, and more than a handful of busy streams cost it dearly; with up to `groups` provers every
//     one is its own team and nothing is ever merged;
//   * round 1: the first caller of a team opens a COHORT and waits a bounded time (gather_wait) for its team-mates -- not at all
//     when none of them is about to start a proof; whoever completes the cohort (it is full, the wait is over, nobody else can
//     come) runs the round for every lane on its own thread and wakes the rest.  Team-mates that proved together finish together
//     and come back together, so in the steady state a cohort re-forms within microseconds;
//   * rounds 2..R: the cohort's members meet again; the last to arrive runs the round.  A member that stays away longer than
//     straggler_wait is MOVED OUT: the backend copies its lane into the member's own workspace, where its proof goes on alone;
//   * a lane whose own data is at fault fails alone; a member that abandons its proof (destroys the prover, starts another
//     proof) leaves the cohort, which goes on without it.
// Every caller gets exactly the result a prover of one proof would have given it.
//
// Backend (template parameter), all called WITHOUT the core's lock held, from the thread that runs the round:
//   struct CohortData;  struct MemberData;                              opaque to the core
//   int  open (CohortData&, MemberData& leader, uint32_t lanes);          choose the workspace of a new cohort (lanes == 1: the leader's own)
//   int  run  (CohortData&, int round, uint32_t lanes, void* const* args, const uint8_t* present, int* lane_rc, std::string* lane_msg);
//                                                                         one round over lanes [0, lanes); args[l] null <=> !present[l];
//                                                                         a non-zero return fails every lane
//   int  move_out(CohortData& from, uint32_t lane, MemberData& to, CohortData& solo);   lane -> the member's own workspace, as a cohort of one
//   void close(CohortData&);                                              the cohort is over (last round done, failed, or everyone left)
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace uzk {

template <class Backend>
class CoalesceCore {
public:
    using Clock = std::chrono::steady_clock;
    static constexpr int kErrParameter = 1;      // UZK_ERR_PARAMETER (static_assert in coalesce.cpp)

    struct Cohort;
    struct Member {
        typename Backend::MemberData data;
        uint64_t group = 0;                      // members of one group can share a cohort at all (same n, same device)
        int team = -1;                           // which of the group's teams it gathers with (dealt at its first proof, re-dealt when the teams get lopsided)
        std::shared_ptr<Cohort> cohort;
        uint32_t lane = 0;
        int round = 0;                           // rounds completed of the proof in flight
        bool in_call = false, moving = false;
        std::chrono::steady_clock::time_point last_seen{};   // when it last left a call
        std::string last_blob;                   // the key of its last proof: a prover between two proofs is expected back with the same one
    };
    // what must agree for two round-1 calls to share a cohort; `blob` is compared bytewise (circuit, hiding degrees, public-input indices ...)
    struct Key {
        uint64_t group = 0;
        std::string blob;
        bool operator==(const Key& o) const { return group == o.group && blob == o.blob; }
    };
    struct LaneRec {
        Member* m = nullptr;                     // null: the member has left, the lane is dead
        bool here = false;
        void* args = nullptr;
        int done_round = 0, rc = 0;
        std::string msg;
    };
    struct Cohort {
        Key key;
        int team = 0;
        bool gathering = true, running = false;
        bool held = false;                       // it has been kept open for the larger part of its team
        int next_round = 1;
        uint32_t alive = 0, arrived = 0;
        std::vector<LaneRec> lanes;
        Clock::time_point deadline, last_done{}, born{};
        std::condition_variable cv;
        typename Backend::CohortData data;
    };

    CoalesceCore(Backend& b, int rounds, uint32_t max_lanes = 8, uint32_t gather_wait_us = 2000, uint32_t straggler_wait_us = 20000, uint32_t groups = 4)
        : backend_(b), rounds_(rounds), max_lanes_(max_lanes), groups_(groups), gather_wait_(gather_wait_us), straggler_wait_(straggler_wait_us),
          recent_(std::max(std::chrono::microseconds(5000), 4 * std::chrono::microseconds(gather_wait_us))) {}

    // groups: how many teams the provers of a kind are dealt into
    void configure(uint32_t max_lanes, uint32_t gather_wait_us, uint32_t straggler_wait_us, uint32_t groups) {
        std::lock_guard<std::mutex> lk(mu_);
        max_lanes_ = std::max<uint32_t>(1, max_lanes);
        groups_ = std::max<uint32_t>(1, groups);
        gather_wait_ = std::chrono::microseconds(gather_wait_us);
        recent_ = std::max(std::chrono::microseconds(5000), 4 * gather_wait_);
        straggler_wait_ = std::chrono::microseconds(std::max<uint32_t>(1, straggler_wait_us));
        stats_ = Stats();
    }
    uint32_t max_lanes() { std::lock_guard<std::mutex> lk(mu_); return max_lanes_; }
    // rounds run, lanes (callers) served by them, the widest round, lanes moved out, cohorts opened -- since the last configure()
    // gap_us: host time between the end of a cohort's round and the start of its next one (its callers wake, take their
    // results, come back with the next challenges), summed over the `gaps` rounds that had a predecessor; gather_us: time cohorts
    // spent open, from their first caller's arrival to the start of their first round
    struct Stats { uint64_t rounds = 0, lanes = 0, widest = 0, moved_out = 0, cohorts = 0, gap_us = 0, gaps = 0, gather_us = 0, first_round_sizes[8] = {}; };
    Stats stats() { std::lock_guard<std::mutex> lk(mu_); return stats_; }
    void reset_stats() { std::lock_guard<std::mutex> lk(mu_); stats_ = Stats(); }

    void add(Member* m) { std::lock_guard<std::mutex> lk(mu_); m->last_seen = Clock::now(); members_.push_back(m); }
    // false: the member is inside a call on another thread (it stays registered)
    bool remove(Member* m) {
        std::unique_lock<std::mutex> lk(mu_);
        if (m->in_call) return false;
        leave_locked(m, lk);
        members_.remove(m);
        return true;
    }
    // the member gives up the proof it has in flight, if any
    void abandon(Member* m) {
        std::unique_lock<std::mutex> lk(mu_);
        if (!m->in_call) leave_locked(m, lk);
    }
    int rounds_done(Member* m) { std::lock_guard<std::mutex> lk(mu_); return m->round; }

    // One round of one member's proof.  Returns the lane's result code; *msg receives its error text.  key: round 1 only.
    int enter(Member* m, int round, const Key* key, void* args, std::string* msg) {
        std::unique_lock<std::mutex> lk(mu_);
        if (m->in_call) { *msg = "the prover is inside another call (one thread at a time per prover)"; return kErrParameter; }
        m->in_call = true;
        struct Out { Member* m; ~Out() { m->in_call = false; m->last_seen = Clock::now(); } } out{m};     // runs with the lock held (lk is destroyed after it)
        while (m->moving) move_cv_.wait(lk);
        std::shared_ptr<Cohort> g;
        if (round == 1) {
            leave_locked(m, lk);                 // a proof in flight is abandoned
            if (m->last_blob != key->blob) m->last_blob = key->blob;
            const auto now = Clock::now();
            const int team_before = m->team;
            deal(m, now);
            trace("arrive", m, nullptr, team_before);
            for (auto& c : gathering_)
                if (c->team == m->team && c->key == *key && c->lanes.size() < max_lanes_) { g = c; break; }
            // whoever joins brings its companions of the last proof a moment behind it: the cohort's time starts again (a cohort
            // that has been held open for the rest of its team would otherwise leave with the first of them)
            if (g) { g->deadline = std::max(g->deadline, now + gather_wait_); g->cv.notify_all(); }      // (a waiter recomputes its timeout)
            if (!g) {
                g = std::make_shared<Cohort>();
                g->key = *key;
                g->team = m->team;
                g->born = now;
                g->deadline = now + gather_wait_;
                gathering_.push_back(g);
                live_.push_back(g);
                trace("open", m, g.get(), 0);
            }
            m->cohort = g;
            m->lane = (uint32_t)g->lanes.size();
            m->round = 0;
            g->lanes.emplace_back();
            g->alive++;
        } else {
            if (!m->cohort || m->round != round - 1) {
                *msg = "the prover has completed " + std::to_string(m->cohort ? m->round : 0) + " round(s) of its proof, this call needs " + std::to_string(round - 1);
                return kErrParameter;
            }
            g = m->cohort;
        }
        {
            LaneRec& l = g->lanes[m->lane];
            l.m = m; l.args = args; l.here = true;
            g->arrived++;
        }
        const uint32_t lane = m->lane;
        for (;;) {
            // (an ejected member's cohort changes under it only while it is NOT in a call, so g stays m's cohort here)
            LaneRec& l = g->lanes[lane];
            if (l.done_round >= round) break;
            if (!g->running) {
                if (g->gathering) {
                    // Go when the cohort is full, when its time is up, or when no team-mate can be expected soon.  While the team has
                    // ANOTHER cohort under way that is at least as large (team-mates whose phases differ: the start, or after one of
                    // them came late) this one stays open until that cohort's proof ends -- its members then join, and from then on
                    // the team stays together; the wait is bounded by three times the wait for a straggler (a proof of a full cohort
                    // on a busy chip takes longer than one straggler_wait).  The larger part never waits for the smaller one beyond
                    // gather_wait.
                    const auto now = Clock::now();
                    const bool hold = larger_part_under_way(g.get());
                    // the part it was held for has just finished: its members come back within microseconds -- a fresh gathering
                    // wait for them (without it the cohort left the moment the hold ended, its own deadline long past, and the
                    // team-mates, arriving a moment later, held for IT: five provers over four teams made 670 proofs/s where four
                    // make 850)
                    if (hold) g->held = true;
                    else if (g->held) { g->held = false; g->deadline = std::max(g->deadline, now + gather_wait_); }
                    const auto deadline = hold ? g->born + 3 * straggler_wait_ : g->deadline;
                    if (g->lanes.size() >= max_lanes_ || now >= deadline || (!hold && mates_in_sight(g.get(), now) == 0)) {
                        trace(now >= deadline ? "go-deadline" : hold ? "go-full" : "go-complete", m, g.get(), (int)hold);
                        run_round(g, lk, false);
                        continue;
                    }
                    wait(g->cv, lk, deadline - now);
                    continue;
                }
                if (g->arrived == g->alive) { run_round(g, lk, false); continue; }
                if (wait(g->cv, lk, straggler_wait_) && !g->running && !g->gathering && g->lanes[lane].done_round < round && g->arrived < g->alive)
                    run_round(g, lk, true);
                continue;
            }
            g->cv.wait(lk);
        }
        LaneRec& l = g->lanes[lane];
        const int rc = l.rc;
        if (rc != 0) *msg = l.msg;
        return rc;
    }

private:
    // true: the wait timed out.  (ThreadSanitizer builds wait on the system clock: GCC 11's runtime does not know
    // pthread_cond_clockwait, which the steady-clock waits use, and would report the mutex as held throughout.)
    template <class Dur>
    static bool wait(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, Dur d) {
#if defined(__SANITIZE_THREAD__)
        return cv.wait_until(lk, std::chrono::system_clock::now() + d) == std::cv_status::timeout;
#else
        return cv.wait_for(lk, d) == std::cv_status::timeout;
#endif
    }

    // UZK_COALESCE_TRACE=1: the gathering decisions on stderr (microseconds since the first event, member, team, cohort, its size)
    void trace(const char* what, const Member* m, const Cohort* g, int extra) {
        static const bool on = std::getenv("UZK_COALESCE_TRACE") != nullptr;
        if (!on) return;
        static const auto t0 = Clock::now();
        std::fprintf(stderr, "[coalesce %8lld] %-11s member %p team %d cohort %p size %zu extra %d\n",
                     (long long)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count(), what, (const void*)m, m ? m->team : -1, (const void*)g,
                     g ? g->lanes.size() : 0, extra);
    }

    static bool at_work(const Member* o, Clock::time_point now, std::chrono::microseconds recent) { return o->cohort || o->in_call || now - o->last_seen < recent; }

    // Teams of m's group: as many as `groups_` (fewer while that would leave a prover alone in its team), more when a team would
    // exceed max_lanes_.  m keeps its team unless it has none or
    // the teams have become lopsided (provers came or went): then it moves to the emptiest one.
    void deal(Member* m, Clock::time_point now) {
        uint32_t active = 0;
        for (const Member* o : members_) if (o->group == m->group && (o == m || at_work(o, now, recent_))) ++active;
        // More provers than teams: no team of ONE -- a lone prover runs the one-proof pipeline (its commits' host part and all),
        // and a mix of such with pairs over the same streams is slower than either (five provers over four teams: 630-700
        // proofs/s, over two: 945; six: 830 / 1000; seven: 990 / 1050 -- profiles/r05_shared_odd_thread_counts.txt).
        uint32_t teams = groups_;
        if (active > groups_ || active >= 4) teams = std::min<uint32_t>(groups_, std::max<uint32_t>(1, active / 2));     // (four provers: two pairs make 900, four alone 855)
        teams = std::max<uint32_t>(teams, (active + max_lanes_ - 1) / max_lanes_);
        std::vector<uint32_t> load(teams, 0);
        for (const Member* o : members_)
            if (o != m && o->group == m->group && o->team >= 0 && (uint32_t)o->team < teams && at_work(o, now, recent_)) load[o->team]++;
        uint32_t best = 0;
        for (uint32_t t = 1; t < teams; ++t) if (load[t] < load[best]) best = t;
        if (m->team < 0 || (uint32_t)m->team >= teams || load[m->team] > load[best]) m->team = (int)best;      // (loads without m: equal teams stay as they are)
    }

    // Team-mates of g's members that could still join before its wait is over: provers between two proofs that were at work a
    // moment ago (within a few gathering waits), and provers in the LAST round of a proof.  Provers in the middle of a proof cannot arrive within gather_wait;
    // provers idle for long belong to threads that are doing something else.
    uint32_t mates_in_sight(const Cohort* g, Clock::time_point now) const {
        uint32_t c = 0;
        for (const Member* o : members_) {
            if (o->group != g->key.group || (o->team != g->team && o->team >= 0) || o->cohort.get() == g) continue;     // (team < 0: not dealt yet -- it may come here)
            // (a team-mate whose last proof was over another circuit is not waited for: it cannot join this group)
            if (!o->cohort) { if ((o->in_call || now - o->last_seen < recent_) && (o->last_blob.empty() || o->last_blob == g->key.blob)) ++c; }
            else if (!o->cohort->gathering && o->cohort->next_round >= rounds_ && o->cohort->key == g->key) ++c;
        }
        return c;
    }
    // the team has another cohort of this kind under way with at least as many members as g
    bool larger_part_under_way(const Cohort* g) const {
        for (const auto& o : live_)
            if (o.get() != g && o->team == g->team && o->key == g->key && !o->gathering && o->alive >= g->lanes.size()) return true;
        return false;
    }

    void finish_cohort(const std::shared_ptr<Cohort>& g, std::unique_lock<std::mutex>& lk) {
        gathering_.remove(g);
        live_.remove(g);
        g->gathering = false;
        for (auto& o : gathering_) o->cv.notify_all();      // one launch sequence fewer: a cohort held open may go now
        lk.unlock();
        backend_.close(g->data);
        lk.lock();
    }

    void leave_locked(Member* m, std::unique_lock<std::mutex>& lk) {
        while (m->moving) move_cv_.wait(lk);
        std::shared_ptr<Cohort> g = m->cohort;
        if (!g) return;
        LaneRec& l = g->lanes[m->lane];
        if (l.here) { l.here = false; g->arrived--; }
        l.m = nullptr;
        g->alive--;
        m->cohort.reset();
        m->round = 0;
        if (g->alive == 0 && !g->running) finish_cohort(g, lk);
        else {
            g->cv.notify_all();                  // the rest may be complete now: a waiter runs the round
            wake_held(g.get());                  // a cohort held open for this one may no longer have a larger part to wait for
        }
    }
    // a live cohort has lost a member without finishing: cohorts held open because of it (larger_part_under_way) look again
    void wake_held(const Cohort* shrunk) {
        for (auto& o : gathering_) if (o.get() != shrunk && o->held) o->cv.notify_all();
    }

    // Runs the cohort's next round on the calling thread.  eject: alive members that have not arrived are moved out first.
    void run_round(const std::shared_ptr<Cohort>& g, std::unique_lock<std::mutex>& lk, bool eject) {
        const bool first = g->gathering;
        if (first) { g->gathering = false; gathering_.remove(g); }
        g->running = true;
        const auto started = Clock::now();
        if (g->last_done != Clock::time_point{}) {
            stats_.gap_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(started - g->last_done).count();
            stats_.gaps++;
        }
        const int r = g->next_round;
        const uint32_t k = (uint32_t)g->lanes.size();
        std::vector<void*> args(k, nullptr);
        std::vector<uint8_t> present(k, 0);
        std::vector<uint32_t> out_lanes;
        for (uint32_t i = 0; i < k; ++i) {
            LaneRec& l = g->lanes[i];
            if (!l.m) continue;
            if (l.here) { args[i] = l.args; present[i] = 1; }
            else if (eject) { l.m->moving = true; out_lanes.push_back(i); }
        }
        std::vector<int> lane_rc(k, 0);
        std::vector<std::string> lane_msg(k);
        std::vector<std::shared_ptr<Cohort>> solos(out_lanes.size());
        std::vector<int> move_rc(out_lanes.size(), 0);
        typename Backend::MemberData* leader = nullptr;
        for (uint32_t i = 0; i < k && !leader; ++i) if (present[i]) leader = &g->lanes[i].m->data;
        lk.unlock();
        int rc = 0;
        if (first) rc = backend_.open(g->data, *leader, k);
        for (size_t j = 0; j < out_lanes.size(); ++j) {
            solos[j] = std::make_shared<Cohort>();
            move_rc[j] = rc ? rc : backend_.move_out(g->data, out_lanes[j], g->lanes[out_lanes[j]].m->data, solos[j]->data);
        }
        if (rc == 0) rc = backend_.run(g->data, r, k, args.data(), present.data(), lane_rc.data(), lane_msg.data());
        std::string all_msg;
        if (rc != 0) all_msg = backend_.last_error();
        lk.lock();
        g->last_done = Clock::now();
        {
            uint64_t served = 0;
            for (uint32_t i = 0; i < k; ++i) served += present[i];
            stats_.rounds++; stats_.lanes += served; stats_.widest = std::max(stats_.widest, served);
            stats_.moved_out += out_lanes.size();
            if (first) { stats_.first_round_sizes[std::min<uint64_t>(served, 8) - 1]++; stats_.cohorts++; stats_.gather_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(started - g->born).count(); }
        }
        for (size_t j = 0; j < out_lanes.size(); ++j) {
            LaneRec& l = g->lanes[out_lanes[j]];
            Member* m = l.m;
            l.m = nullptr;
            g->alive--;
            m->moving = false;
            if (move_rc[j] == 0) {
                std::shared_ptr<Cohort>& h = solos[j];
                live_.push_back(h);
                h->key = g->key; h->team = g->team; h->gathering = false; h->next_round = r; h->alive = 1;
                h->lanes.emplace_back();
                h->lanes[0].m = m;
                m->cohort = h; m->lane = 0;
            } else {
                m->cohort.reset(); m->round = 0;  // its proof is lost: the next round call reports the wrong order
            }
        }
        bool over = rc != 0 || r == rounds_;
        for (uint32_t i = 0; i < k; ++i) {
            LaneRec& l = g->lanes[i];
            if (!present[i] || !l.m) { l.here = false; continue; }
            l.here = false;
            l.done_round = r;
            l.rc = rc ? rc : lane_rc[i];
            l.msg = rc ? all_msg : lane_msg[i];
            Member* m = l.m;
            if (l.rc != 0 || over) { m->cohort.reset(); m->round = 0; l.m = nullptr; g->alive--; }
            else m->round = r;
        }
        g->arrived = 0;
        for (uint32_t i = 0; i < k; ++i) if (g->lanes[i].m && g->lanes[i].here) g->arrived++;
        g->next_round = r + 1;
        g->running = false;
        if (g->alive == 0) finish_cohort(g, lk);
        else if (g->alive < k) wake_held(g.get());       // lanes failed or were moved out
        g->cv.notify_all();
        if (!out_lanes.empty()) move_cv_.notify_all();
    }

    Backend& backend_;
    const int rounds_;
    std::mutex mu_;
    std::condition_variable move_cv_;
    std::list<Member*> members_;
    std::list<std::shared_ptr<Cohort>> gathering_, live_;      // cohorts still taking members; every cohort that is not over
    Stats stats_;
    uint32_t max_lanes_, groups_;
    std::chrono::microseconds gather_wait_, straggler_wait_, recent_;      // recent_: how long after its last call a prover still counts as at work
};

}  // namespace uzk
