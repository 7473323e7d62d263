// uzk_prover_* / uzk_prove_round1..5 (include/uzkge_gpu.h) and the provers of ONE proof that the library may run together.
//
// A prover made with batch >= 2 -- or with sharing switched off -- owns its lanes (prover.cpp, explicit_*).  A prover of one
// proof made while sharing is on (the default) is a SHARED prover: its rounds go through the coalescing core
// (coalesce_core.hpp), which runs the calls of several host threads that stand at the same round of proofs over the same circuit
// as one lockstep launch sequence over a pooled workspace with its own internal context (stream, MSM / NTT workspaces), and
// hands every caller its own outputs.  The reference's call pattern -- one proof per call, from whatever threads the
// application has (uzkge/src/plonk/prover.rs:88-100; shuffle/src/sdk.rs:196-214) -- reaches the lockstep throughput this way
// without a new API.  Alone, a shared prover runs on its own one-lane workspace exactly as a prover of one proof always did.
#include <condition_variable>
#include <cstring>
#include <map>

#include "coalesce_core.hpp"
#include "prover.hpp"

namespace uzk {
// prover.cpp
int explicit_prover_create(uint32_t n, uint32_t batch, uint64_t* prover_out);
bool explicit_prover_known(uint64_t prover);
int explicit_prover_destroy(uint64_t prover);
int explicit_prover_buffer(uint64_t prover, int which, void** d_out, uint64_t* elems_out);
int explicit_round1(uint64_t prover, uint64_t circuit, const void* witness, const void* wsel, int inputs_on_device, const uint32_t* pi_index, const uint64_t* pi_value,
                    uint32_t pi_count, const uint32_t* hiding, const uint64_t* blinds, uzk_g1_jac* cm_out);
int explicit_round2(uint64_t prover, const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinds_z, uzk_g1_jac* cm_z_out);
int explicit_round3(uint64_t prover, const uint64_t* alpha, const uint64_t* t_rands, uzk_g1_jac* cm_t_out);
int explicit_round4(uint64_t prover, const uint64_t* zeta, uint64_t* evals_out, size_t evals_cap);
int explicit_round5(uint64_t prover, const uint64_t* r_scalars, size_t r_count, const uint64_t* alpha_zeta, const uint64_t* alpha_zeta_omega, uzk_g1_jac* openings_out);

namespace {

constexpr uint64_t kSharedBit = 1ull << 62;              // handles of shared provers (the explicit ones count up from 1)
constexpr uint32_t kGatherWaitUs = 2000, kStragglerWaitUs = 20000, kGroups = 4;      // the defaults of uzk_coalesce_config (profiles/r05_gather_sweep*.txt)
constexpr uint32_t kMaxStreams = 4;                      // busy streams of the library's own per device, whatever `groups` is (Turns below)

// A lockstep workspace: the buffers of up to `cap` proofs.  It has no stream of its own: a round runs on whichever of the few
// internal contexts is free (Turns below) -- rounds end synchronised, so nothing of a proof is in flight between two of them.
struct Slot {
    std::shared_ptr<Prover> prover;
    uint32_t n = 0, cap = 0;
    int device = 0;
};

// one lane's arguments of a round, on the caller's stack while it is inside the call
struct A1 {
    Lane1 lane;
    std::shared_ptr<Circuit> cir;
    const uint32_t* pi_index;
    uint32_t pi_count;
    const uint32_t* hiding;
    int on_device;
    hipEvent_t ev;
};

struct Backend {
    struct MemberData {
        uint32_t n = 0;
        int device = 0;
        std::shared_ptr<Slot> home;                      // one lane, made on first need: the proofs this prover runs alone, and where a straggler's lane moves
        hipEvent_t ev = nullptr;                         // orders device-resident inputs of the caller's context before the shared stream
    };
    struct CohortData {
        std::shared_ptr<Slot> slot;
        bool pooled = false;
    };

    // The internal contexts: `max_running` per device, each one stream with its workspaces, made once and one after the other
    // (streams born in the same instant on several threads have been seen to share a hardware queue, where their launch sequences
    // then take turns).  A round takes a free one for its duration.  So at most `max_running` <= 4 streams of the library's own
    // are ever busy: the chip's compute front end has four pipes, and a fifth busy queue shares one with the first -- the two then
    // run at 0.6 of the others' rate and the whole at 0.75 of four streams' throughput (profiles/r05_gaps_lockstep_5x8.txt: 1100
    // proofs/s from five lockstep provers of eight proofs where four make 1450).  Cohorts beyond that number take turns round by
    // round.
    struct Turns {
        std::vector<std::unique_ptr<Ctx>> ctx;
        std::vector<uint8_t> busy;
    };
    std::mutex mu;                                       // the pool, the turns
    std::condition_variable turn_cv;
    std::map<int, Turns> turns;
    uint32_t max_running = 4;
    std::vector<std::shared_ptr<Slot>> pool;
    uint32_t pool_cap = 8;                               // lanes of the workspaces made from now on

    struct Turn {
        Backend& b;
        int device;
        size_t at = 0;
        Ctx* c = nullptr;
        int rc = UZK_OK;
        Turn(Backend& b_, int device_) : b(b_), device(device_) {
            std::unique_lock<std::mutex> lk(b.mu);
            Turns& t = b.turns[device];
            for (;;) {
                for (size_t i = 0; i < t.ctx.size(); ++i)
                    if (!t.busy[i]) { at = i; c = t.ctx[i].get(); t.busy[i] = 1; return; }
                if (t.ctx.size() < b.max_running) {
                    auto nc = std::make_unique<Ctx>();
                    rc = ctx_init_internal(*nc, device);
                    if (rc != UZK_OK) return;
                    t.ctx.push_back(std::move(nc));
                    t.busy.push_back(0);
                    continue;
                }
                b.turn_cv.wait(lk);
            }
        }
        ~Turn() {
            if (!c) return;
            { std::lock_guard<std::mutex> lk(b.mu); b.turns[device].busy[at] = 0; }
            b.turn_cv.notify_one();
        }
    };

    std::string last_error() { return uzk_last_error(); }

    int make_slot(uint32_t n, uint32_t cap, int device, std::shared_ptr<Slot>* out) {
        Turn t(*this, device);
        UZK_TRY(t.rc);
        auto s = std::make_shared<Slot>();
        s->n = n; s->cap = cap; s->device = device;
        CtxScope scope(t.c);
        std::lock_guard<std::mutex> lk(t.c->mu);
        UZK_TRY(require_ready());
        UZK_TRY(prover_alloc(*t.c, n, cap, &s->prover));
        *out = s;
        return UZK_OK;
    }
    int home_of(MemberData& m) {
        if (m.home) return UZK_OK;
        return make_slot(m.n, 1, m.device, &m.home);
    }

    int open(CohortData& cd, MemberData& leader, uint32_t lanes) {
        if (lanes == 1) { UZK_TRY(home_of(leader)); cd.slot = leader.home; cd.pooled = false; return UZK_OK; }
        const uint32_t n = leader.n;
        const int device = leader.device;
        uint32_t cap;
        {
            std::lock_guard<std::mutex> lk(mu);
            cap = std::max(pool_cap, lanes);
            for (size_t i = 0; i < pool.size(); ++i)
                if (pool[i]->n == n && pool[i]->device == device && pool[i]->cap >= lanes) {
                    cd.slot = pool[i];
                    pool.erase(pool.begin() + (long)i);
                    cd.pooled = true;
                    return UZK_OK;
                }
        }
        int rc = make_slot(n, cap, device, &cd.slot);
        if (rc != UZK_OK && cap > lanes) rc = make_slot(n, lanes, device, &cd.slot);     // out of memory for the full width: just this cohort's
        UZK_TRY(rc);
        cd.pooled = true;
        return UZK_OK;
    }

    void close(CohortData& cd) {
        if (!cd.slot) return;
        {
            std::lock_guard<std::mutex> plk(cd.slot->prover->mu);
            prover_end_proof(*cd.slot->prover);
        }
        if (cd.pooled) {
            std::lock_guard<std::mutex> lk(mu);
            pool.push_back(cd.slot);
        }
        cd.slot.reset();
    }

    int move_out(CohortData& from, uint32_t lane, MemberData& to, CohortData& solo) {
        UZK_TRY(home_of(to));
        Slot& s = *from.slot;
        Turn t(*this, s.device);
        UZK_TRY(t.rc);
        CtxScope scope(t.c);
        std::lock_guard<std::mutex> lk(t.c->mu);
        UZK_TRY(require_ready());
        std::lock_guard<std::mutex> p1(s.prover->mu);
        std::lock_guard<std::mutex> p2(to.home->prover->mu);
        Prover& dst = *to.home->prover;
        prover_end_proof(dst);
        UZK_TRY(prover_move_lane(*t.c, *s.prover, lane, dst, 0));
        dst.k = 1;
        dst.dead.assign(1, 0);
        solo.slot = to.home;
        solo.pooled = false;
        return UZK_OK;
    }

    int run(CohortData& cd, int round, uint32_t lanes, void* const* args, const uint8_t* present, int* lane_rc, std::string* lane_msg) {
        Slot& s = *cd.slot;
        Turn t(*this, s.device);
        UZK_TRY(t.rc);
        Ctx& c = *t.c;
        CtxScope scope(&c);
        std::lock_guard<std::mutex> lk(c.mu);
        UZK_TRY(require_ready());
        Prover& p = *s.prover;
        std::lock_guard<std::mutex> plk(p.mu);
        std::vector<LaneStatus> st(lanes);
        uint32_t first = 0;
        while (first < lanes && !present[first]) ++first;
        if (first == lanes) { set_error("shared round without a caller"); return UZK_ERR_PARAMETER; }
        int rc = UZK_OK;
        if (round == 1) {
            const A1& a0 = *static_cast<const A1*>(args[first]);
            std::vector<Lane1> L(lanes);
            for (uint32_t b = 0; b < lanes; ++b) {
                const A1& a = *static_cast<const A1*>(args[b]);      // every lane of a first round has its caller
                L[b] = a.lane;
                if (a.on_device) UZK_HIP(hipStreamWaitEvent(c.stream, a.ev, 0));
            }
            rc = round1_lanes(c, p, a0.cir, lanes, L.data(), a0.on_device, a0.pi_index, a0.pi_count, a0.hiding, st.data());
        } else {
            if (p.k != lanes || p.round != round - 1) { set_error("shared round %d: the workspace holds %u lane(s) after round %d", round, p.k, p.round); return UZK_ERR_PARAMETER; }
            p.owner = &c;                                // the round before ended synchronised: any internal context may run this one
            // a lane whose caller has gone keeps computing on a neighbour's challenges: nothing of it is ever read
            for (uint32_t b = 0; b < lanes; ++b) if (!present[b]) p.dead[b] = 1;
            auto arg = [&](uint32_t b) { return args[present[b] ? b : first]; };
            if (round == 2) { std::vector<Lane2> L(lanes); for (uint32_t b = 0; b < lanes; ++b) L[b] = *static_cast<const Lane2*>(arg(b)); rc = round2_lanes(c, p, L.data(), st.data()); }
            else if (round == 3) { std::vector<Lane3> L(lanes); for (uint32_t b = 0; b < lanes; ++b) L[b] = *static_cast<const Lane3*>(arg(b)); rc = round3_lanes(c, p, L.data(), st.data()); }
            else if (round == 4) { std::vector<Lane4> L(lanes); for (uint32_t b = 0; b < lanes; ++b) L[b] = *static_cast<const Lane4*>(arg(b)); rc = round4_lanes(c, p, L.data(), st.data()); }
            else { std::vector<Lane5> L(lanes); for (uint32_t b = 0; b < lanes; ++b) L[b] = *static_cast<const Lane5*>(arg(b)); rc = round5_lanes(c, p, L.data(), st.data()); }
        }
        for (uint32_t b = 0; b < lanes; ++b) { lane_rc[b] = st[b].rc; lane_msg[b] = st[b].msg; }
        return rc;
    }

    void release_all() {
        std::lock_guard<std::mutex> lk(mu);
        pool.clear();
        for (auto& kv : turns)
            for (auto& c : kv.second.ctx) ctx_release_internal(*c);
        turns.clear();
    }
};

using Core = CoalesceCore<Backend>;
static_assert(Core::kErrParameter == UZK_ERR_PARAMETER, "the core's error code for misuse");

struct Shared1 {                                         // a shared prover
    Core::Member member;
    uint32_t n = 0;
    int device = 0;
};

struct State {
    Backend backend;
    Core core{backend, 5, 8, kGatherWaitUs, kStragglerWaitUs, kGroups};
    std::mutex mu;                                       // the handle table and the switches
    std::map<uint64_t, std::shared_ptr<Shared1>> provers;
    uint64_t next = 1;
    bool enabled = true;
};
State& st() {
    static State* s = new State();                       // never destroyed: threads may still be inside at exit
    return *s;
}
std::shared_ptr<Shared1> find_shared(uint64_t h) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.provers.find(h);
    return it == s.provers.end() ? nullptr : it->second;
}
bool is_shared(uint64_t h) { return (h & kSharedBit) != 0; }

const Fp* as_fp(const uint64_t* p) { return reinterpret_cast<const Fp*>(p); }

int finish(int rc, const std::string& msg) {
    if (rc != UZK_OK) set_error("%s", msg.c_str());
    return rc;
}

void destroy_shared(Shared1& sp) {
    if (sp.member.data.ev) { (void)hipSetDevice(sp.device); (void)hipEventDestroy(sp.member.data.ev); sp.member.data.ev = nullptr; }
    sp.member.data.home.reset();
}

}  // namespace

void coalesce_release_all() {
    State& s = st();
    std::map<uint64_t, std::shared_ptr<Shared1>> ps;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        ps.swap(s.provers);
    }
    bool busy = false;
    for (auto& kv : ps) {
        if (s.core.remove(&kv.second->member)) { destroy_shared(*kv.second); continue; }
        // inside a call on another thread: the core and possibly a live cohort still point at it -- it stays registered, under its
        // handle, with its buffers, and the internal contexts its round is running on stay too
        busy = true;
        std::lock_guard<std::mutex> lk(s.mu);
        s.provers[kv.first] = kv.second;
    }
    if (!busy) s.backend.release_all();
}

}  // namespace uzk

using namespace uzk;

extern "C" {

int uzk_coalesce_config(uint32_t max_lanes, uint32_t gather_wait_us, uint32_t straggler_wait_us, uint32_t groups) try {
    if (max_lanes > kMaxBatch) { set_error("uzk_coalesce_config: at most %u lanes", kMaxBatch); return UZK_ERR_PARAMETER; }
    State& s = st();
    {
        std::lock_guard<std::mutex> lk(s.mu);
        s.enabled = max_lanes >= 2;
    }
    if (max_lanes >= 2) {
        s.core.configure(max_lanes, gather_wait_us ? gather_wait_us : kGatherWaitUs, straggler_wait_us ? straggler_wait_us : kStragglerWaitUs, groups ? groups : kGroups);
        std::lock_guard<std::mutex> lk(s.backend.mu);
        s.backend.pool_cap = max_lanes;
        s.backend.max_running = std::min<uint32_t>(groups ? groups : kGroups, kMaxStreams);
        s.backend.pool.clear();                          // workspaces of another width: made again on demand
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_coalesce_config"); }

int uzk_coalesce_stats(uint64_t out[16]) try {
    if (!out) { st().core.reset_stats(); return UZK_OK; }
    const Core::Stats s = st().core.stats();
    out[0] = s.rounds; out[1] = s.lanes; out[2] = s.widest; out[3] = s.moved_out; out[4] = s.cohorts; out[5] = s.gap_us; out[6] = s.gaps; out[7] = s.gather_us;
    for (int i = 0; i < 8; ++i) out[8 + i] = s.first_round_sizes[i];
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_coalesce_stats"); }

int uzk_prover_create(uint32_t n, uint32_t batch, uint64_t* prover_out) try {
    if (!prover_out) { set_error("uzk_prover_create: null pointer"); return UZK_ERR_PARAMETER; }
    if (n < 16 || n > (1u << 20) || (n & (n - 1)) || batch == 0 || batch > kMaxBatch) {
        set_error("uzk_prover_create: n must be a power of two in 16 .. 2^20 and 1 <= batch <= %u", kMaxBatch);
        return UZK_ERR_PARAMETER;
    }
    State& s = st();
    bool shared;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        // large circuits fill the chip with one proof: sharing would only multiply their workspaces (150 n x 32 B per lane, x 8 lanes
        // x 4 cohorts -- 160 GB at n = 2^20) for nothing; they own their lane
        shared = s.enabled && batch == 1 && n <= UZK_SHARED_MAX_N;
    }
    if (!shared) return explicit_prover_create(n, batch, prover_out);
    int device;
    {
        std::lock_guard<std::mutex> lk(ctx_mutex());
        UZK_TRY(require_ready());
        device = ctx().device;
    }
    auto sp = std::make_shared<Shared1>();
    sp->n = n; sp->device = device;
    sp->member.group = ((uint64_t)n << 8) | (uint64_t)(device & 0xff);
    sp->member.data.n = n; sp->member.data.device = device;
    UZK_HIP(hipSetDevice(device));
    UZK_HIP(hipEventCreateWithFlags(&sp->member.data.ev, hipEventDisableTiming));
    s.core.add(&sp->member);
    std::lock_guard<std::mutex> lk(s.mu);
    const uint64_t h = kSharedBit | s.next++;
    s.provers[h] = sp;
    *prover_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prover_create"); }

int uzk_prover_create_private(uint32_t n, uint32_t batch, uint64_t* prover_out) try {
    if (!prover_out) { set_error("uzk_prover_create_private: null pointer"); return UZK_ERR_PARAMETER; }
    if (n < 16 || n > (1u << 20) || (n & (n - 1)) || batch == 0 || batch > kMaxBatch) {
        set_error("uzk_prover_create_private: n must be a power of two in 16 .. 2^20 and 1 <= batch <= %u", kMaxBatch);
        return UZK_ERR_PARAMETER;
    }
    return explicit_prover_create(n, batch, prover_out);
} catch (...) { return uzk::on_exception("uzk_prover_create_private"); }

int uzk_prover_destroy(uint64_t prover) try {
    if (!is_shared(prover)) return explicit_prover_destroy(prover);
    State& s = st();
    // the handle leaves the table first, under the table's lock: of two threads destroying one handle only one finds it, and only
    // that one goes on to the core and the buffers
    std::shared_ptr<Shared1> sp;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.provers.find(prover);
        if (it != s.provers.end()) { sp = it->second; s.provers.erase(it); }
    }
    if (!sp) { set_error("uzk_prover_destroy: unknown prover %llu", (unsigned long long)prover); return UZK_ERR_PARAMETER; }
    if (!s.core.remove(&sp->member)) {
        { std::lock_guard<std::mutex> lk(s.mu); s.provers[prover] = sp; }
        set_error("uzk_prover_destroy: the prover is inside a call on another thread");
        return UZK_ERR_PARAMETER;
    }
    destroy_shared(*sp);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prover_destroy"); }

int uzk_prover_buffer(uint64_t prover, int which, void** d_out, uint64_t* elems_out) try {
    if (is_shared(prover)) {
        set_error("uzk_prover_buffer: a shared prover's proofs run on pooled lanes; make it with uzk_prover_create_private to inspect its buffers");
        return UZK_ERR_PARAMETER;
    }
    return explicit_prover_buffer(prover, which, d_out, elems_out);
} catch (...) { return uzk::on_exception("uzk_prover_buffer"); }

int uzk_prove_round1(uint64_t prover, uint64_t circuit, const void* witness, const void* wsel, int inputs_on_device,
                     const uint32_t* pi_index, const uint64_t* pi_value, uint32_t pi_count, const uint32_t* hiding,
                     const uint64_t* blinds, uzk_g1_jac* cm_out) try {
    if (!witness || !hiding || !blinds || !cm_out || (pi_count && (!pi_index || !pi_value))) { set_error("uzk_prove_round1: null pointer"); return UZK_ERR_PARAMETER; }
    if (!is_shared(prover)) return explicit_round1(prover, circuit, witness, wsel, inputs_on_device, pi_index, pi_value, pi_count, hiding, blinds, cm_out);
    auto sp = find_shared(prover);
    auto cir = find_circuit(circuit);
    if (!sp || !cir) { set_error("uzk_prove_round1: unknown prover or circuit"); return UZK_ERR_PARAMETER; }
    UZK_TRY(round1_check(*cir, sp->n, wsel != nullptr, pi_index, pi_count, hiding));
    if (cir->device != sp->device) { set_error("uzk_prove_round1: the circuit lives on device %d, the prover on device %d", cir->device, sp->device); return UZK_ERR_PARAMETER; }
    const uint32_t n_first = wsel ? kWires + kWsel : kWires;
    A1 a;
    a.lane = Lane1{witness, wsel, as_fp(pi_value), as_fp(blinds), reinterpret_cast<Jac*>(cm_out)};
    a.cir = cir; a.pi_index = pi_index; a.pi_count = pi_count; a.hiding = hiding; a.on_device = inputs_on_device ? 1 : 0; a.ev = sp->member.data.ev;
    if (inputs_on_device) {
        // the caller's context produced the inputs: whatever stream runs the round waits for what that context has queued so far
        std::lock_guard<std::mutex> lk(ctx_mutex());
        UZK_TRY(require_ready());
        if (ctx().device != sp->device) { set_error("uzk_prove_round1: device-resident inputs from a context on device %d, the prover lives on device %d", ctx().device, sp->device); return UZK_ERR_PARAMETER; }
        UZK_HIP(hipEventRecord(a.ev, ctx().stream));
    }
    Core::Key key;
    key.group = sp->member.group;
    {
        const Circuit* cp = cir.get();
        const uint32_t flags = (wsel ? 1u : 0u) | (inputs_on_device ? 2u : 0u);
        key.blob.append(reinterpret_cast<const char*>(&cp), sizeof cp);
        key.blob.append(reinterpret_cast<const char*>(&flags), sizeof flags);
        key.blob.append(reinterpret_cast<const char*>(hiding), n_first * sizeof(uint32_t));
        key.blob.append(reinterpret_cast<const char*>(&pi_count), sizeof pi_count);
        if (pi_count) key.blob.append(reinterpret_cast<const char*>(pi_index), (size_t)pi_count * sizeof(uint32_t));
    }
    std::string msg;
    return finish(st().core.enter(&sp->member, 1, &key, &a, &msg), "uzk_prove_round1: " + msg);
} catch (...) { return uzk::on_exception("uzk_prove_round1"); }

int uzk_prove_round2(uint64_t prover, const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinds_z, uzk_g1_jac* cm_z_out) try {
    if (!beta || !gamma || !blinds_z || !cm_z_out) { set_error("uzk_prove_round2: null pointer"); return UZK_ERR_PARAMETER; }
    if (!is_shared(prover)) return explicit_round2(prover, beta, gamma, blinds_z, cm_z_out);
    auto sp = find_shared(prover);
    if (!sp) { set_error("uzk_prove_round2: unknown prover"); return UZK_ERR_PARAMETER; }
    Lane2 a{as_fp(beta), as_fp(gamma), as_fp(blinds_z), reinterpret_cast<Jac*>(cm_z_out)};
    std::string msg;
    return finish(st().core.enter(&sp->member, 2, nullptr, &a, &msg), "uzk_prove_round2: " + msg);
} catch (...) { return uzk::on_exception("uzk_prove_round2"); }

int uzk_prove_round3(uint64_t prover, const uint64_t* alpha, const uint64_t* t_rands, uzk_g1_jac* cm_t_out) try {
    if (!alpha || !t_rands || !cm_t_out) { set_error("uzk_prove_round3: null pointer"); return UZK_ERR_PARAMETER; }
    if (!is_shared(prover)) return explicit_round3(prover, alpha, t_rands, cm_t_out);
    auto sp = find_shared(prover);
    if (!sp) { set_error("uzk_prove_round3: unknown prover"); return UZK_ERR_PARAMETER; }
    Lane3 a{as_fp(alpha), as_fp(t_rands), reinterpret_cast<Jac*>(cm_t_out)};
    std::string msg;
    return finish(st().core.enter(&sp->member, 3, nullptr, &a, &msg), "uzk_prove_round3: " + msg);
} catch (...) { return uzk::on_exception("uzk_prove_round3"); }

int uzk_prove_round4(uint64_t prover, const uint64_t* zeta, uint64_t* evals_out, size_t evals_cap) try {
    if (!zeta || !evals_out) { set_error("uzk_prove_round4: null pointer"); return UZK_ERR_PARAMETER; }
    if (!is_shared(prover)) return explicit_round4(prover, zeta, evals_out, evals_cap);
    auto sp = find_shared(prover);
    if (!sp) { set_error("uzk_prove_round4: unknown prover"); return UZK_ERR_PARAMETER; }
    Lane4 a{as_fp(zeta), reinterpret_cast<Fp*>(evals_out), evals_cap};     // checked against the circuit's count where the round runs
    std::string msg;
    return finish(st().core.enter(&sp->member, 4, nullptr, &a, &msg), "uzk_prove_round4: " + msg);
} catch (...) { return uzk::on_exception("uzk_prove_round4"); }

int uzk_prove_round5(uint64_t prover, const uint64_t* r_scalars, size_t r_count, const uint64_t* alpha_zeta, const uint64_t* alpha_zeta_omega,
                     uzk_g1_jac* openings_out) try {
    if (!r_scalars || !alpha_zeta || !alpha_zeta_omega || !openings_out) { set_error("uzk_prove_round5: null pointer"); return UZK_ERR_PARAMETER; }
    if (!is_shared(prover)) return explicit_round5(prover, r_scalars, r_count, alpha_zeta, alpha_zeta_omega, openings_out);
    auto sp = find_shared(prover);
    if (!sp) { set_error("uzk_prove_round5: unknown prover"); return UZK_ERR_PARAMETER; }
    Lane5 a{as_fp(r_scalars), r_count, as_fp(alpha_zeta), as_fp(alpha_zeta_omega), reinterpret_cast<Jac*>(openings_out)};
    std::string msg;
    return finish(st().core.enter(&sp->member, 5, nullptr, &a, &msg), "uzk_prove_round5: " + msg);
} catch (...) { return uzk::on_exception("uzk_prove_round5"); }

}  // extern "C"
