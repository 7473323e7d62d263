// Circuits, provers and the five rounds as the rest of the library sees them (prover.cpp implements, coalesce.cpp drives the
// rounds for provers of one proof that several host threads call at the same time).
#pragma once
#include <memory>
#include <string>

#include "ctx.hpp"

namespace uzk {

constexpr uint32_t kSlots = UZK_CIRCUIT_SLOTS;
constexpr uint32_t kWires = 5, kWsel = 3, kProofSlots = 10;       // slots of a proof's own polynomials: w0..4, w_sel0..2, pi, z
constexpr uint32_t kTail = 6;                                      // blinds || -blinds, three slots each (apply_blind_factors)
constexpr uint32_t kMaxBatch = 64;

struct DevBlock {                                                  // device memory that any thread may drop the last reference to
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~DevBlock();
};
int dev_block(size_t bytes, std::shared_ptr<DevBlock>* out);

// One slot of a circuit: coefficient form (n elements allocated, `len` meaningful) and coset evaluations (6n), both inside a
// block shared with the other slots that were installed by the same call.
struct SlotRef {
    std::shared_ptr<DevBlock> blk;
    const Fp* poly = nullptr;
    const Fp* coset = nullptr;
    uint64_t len = 0;
};
struct TableSet { SlotRef s[kSlots]; };

// The commit bases of a circuit (Lagrange SRS || six monomial powers) and their window tables: owned here, so that a proof in
// flight keeps them after uzk_circuit_release (freed with the last reference).
struct CommitBases {
    int device = 0;
    Ctx::Srs plain;                                                // narrow-window table (or none): provers of one proof
    Ctx::Srs wide;                                                 // the same points under the wide-window table of lockstep batches (d_table null: none)
    ~CommitBases();
};

struct Circuit {
    uint32_t n = 0, m = 0, shuffle = 0, n_slots = 0;
    int device = 0;
    std::shared_ptr<CommitBases> bases;
    Fp k[kWires], anemoi_g, anemoi_g_inv, edwards_a, group_gen, k1_inv, z_h_inv[6];
    std::shared_ptr<DevBlock> fixed;                               // group (n Fp) | permutation (5n u32)
    const uint32_t* d_perm = nullptr;
    const Fp* d_group = nullptr;
    int truncate_t = 0;                                            // uzk_test_circuit_truncate_t: a synthetic circuit no witness satisfies
    std::mutex mu;                                                 // guards `tables`
    std::shared_ptr<const TableSet> tables;
};

// The buffers of up to B proofs that advance in lockstep.  Layouts are per lane: lane b's part of every buffer lies a fixed
// stride behind lane 0's, which is all the lane kernels (rounds.hip) and the strided NTT / MSM batches need.
struct Prover {
    uint32_t n = 0, m = 0, cs = 0, B = 0;
    int device = 0;
    std::mutex mu;
    std::shared_ptr<DevBlock> blk;
    Fp *d_evals = nullptr, *d_coefs = nullptr, *d_coset = nullptr, *d_tq = nullptr, *d_t = nullptr, *d_chunks = nullptr, *d_fold = nullptr,
       *d_tail = nullptr, *d_q = nullptr, *d_r = nullptr, *d_h = nullptr;
    uint32_t* d_counters = nullptr;                                // B x 19 arrival counters of the evaluation kernel (zero between launches)
    uint64_t* d_trim = nullptr;                                    // 2 x 2B trimmed-length results, used alternately
    uint32_t trim_flip = 0;
    ArgArena args;
    // pinned: measured trimmed lengths (t: B, quotients: 2B), the evaluations of round 4 (B x 19), the public inputs of round 1
    uint64_t* h_lens = nullptr;
    Fp* h_evals = nullptr;
    void* h_pi = nullptr;
    size_t h_pi_cap = 0;
    // the proof(s) in flight
    int round = 0;                                                 // rounds completed
    uint32_t k = 0;                                                // lanes in use (<= B)
    Ctx* owner = nullptr;                                          // the context round 1 ran on: the later rounds must come from it
    std::shared_ptr<Circuit> circuit;
    std::shared_ptr<const TableSet> snap;
    uint32_t n_first = 0, np = 0;                                  // committed in round 1 (5 or 8); slots in use (7 or 10)
    uint32_t hiding[kWires + kWsel] = {};
    std::vector<uint8_t> dead;                                     // lanes whose proof has failed (their launches go on, harmlessly)
    std::vector<Fp> beta, gamma, zeta, zeta_omega;
    std::vector<uint64_t> chunk_lens;                              // k x 5
    ~Prover();
    uint32_t sl_pi() const { return n_first; }
    uint32_t sl_z() const { return n_first + 1; }
    Fp* evals(uint32_t b, uint32_t slot) const { return d_evals + ((uint64_t)b * kProofSlots + slot) * n; }
    Fp* coefs(uint32_t b, uint32_t slot) const { return d_coefs + ((uint64_t)b * kProofSlots + slot) * m; }
    Fp* coset(uint32_t b, uint32_t slot) const { return d_coset + ((uint64_t)b * kProofSlots + slot) * m; }
};

// One lane's inputs and outputs of a round (host memory unless said otherwise).
struct Lane1 { const void* witness; const void* wsel; const Fp* pi_value; const Fp* blinds; Jac* cm_out; };
struct Lane2 { const Fp* beta; const Fp* gamma; const Fp* blinds_z; Jac* cm_z_out; };
struct Lane3 { const Fp* alpha; const Fp* t_rands; Jac* cm_t_out; };
struct Lane4 { const Fp* zeta; Fp* evals_out; size_t evals_cap; };
struct Lane5 { const Fp* r_scalars; size_t r_count; const Fp* alpha_zeta; const Fp* alpha_zeta_omega; Jac* openings_out; };
// What became of one lane in a round that went through as a whole (a lane whose own data is at fault -- an unsatisfied
// witness -- fails alone; its neighbours' proofs go on).
struct LaneStatus {
    int rc = UZK_OK;
    std::string msg;
};

int prover_alloc(Ctx& c, uint32_t n, uint32_t B, std::shared_ptr<Prover>* out);
// argument checks shared by every way into round 1 (null pointers are the caller's to check)
int round1_check(const Circuit& cir, uint32_t n, bool has_wsel, const uint32_t* pi_index, uint32_t pi_count, const uint32_t* hiding);
// The rounds over the first k lanes of p, on context c (its lock held by the caller, p.mu too).  st: k entries, or null -- then
// the first lane failure fails the call.  A non-OK return has ended the proof of every lane.
int round1_lanes(Ctx& c, Prover& p, const std::shared_ptr<Circuit>& cir, uint32_t k, const Lane1* L, int inputs_on_device, const uint32_t* pi_index,
                 uint32_t pi_count, const uint32_t* hiding, LaneStatus* st);
int round2_lanes(Ctx& c, Prover& p, const Lane2* L, LaneStatus* st);
int round3_lanes(Ctx& c, Prover& p, const Lane3* L, LaneStatus* st);
int round4_lanes(Ctx& c, Prover& p, const Lane4* L, LaneStatus* st);
int round5_lanes(Ctx& c, Prover& p, const Lane5* L, LaneStatus* st);
// copies everything lane `from` of src holds of its proof in flight into lane `to` of dst (same n; c's stream; synchronises)
int prover_move_lane(Ctx& c, const Prover& src, uint32_t from, Prover& dst, uint32_t to);
void prover_end_proof(Prover& p);
uint32_t evals_per_proof(const Circuit& cir);                      // 15 or 19
uint32_t r_scalars_per_proof(const Circuit& cir);                  // 19 or 43

std::shared_ptr<Circuit> find_circuit(uint64_t h);

}  // namespace uzk
