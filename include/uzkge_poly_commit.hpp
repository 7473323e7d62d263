// uzkge_poly_commit.hpp -- header-only C++ mirror of the reference's polynomial layer for the hot
// path, over the C ABI of uzkge_gpu.h.  Same names, argument meaning and error behaviour as
//   FpPolynomial                uzkge/src/poly_commit/field_polynomial.rs:13-17,86-90,154-159,554-607
//   KZGCommitmentSchemeBN254    uzkge/src/poly_commit/kzg_poly_commitment.rs:170-313
//   UzkgeError                  uzkge/src/errors.rs:5-44
// What lives here is what stays on the host in the Rust integration: trimming,
// zero-padding, domain choice, length checks.  All arithmetic runs on the GPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "uzkge_gpu.h"

namespace uzkge {

enum class UzkgeError { ParameterError = 1, DegreeError = 2, FFTError = 3, CommitmentError = 4, DeviceError = 5, DeserializationError = 6 };

struct UzkgeException : std::runtime_error {
    UzkgeError kind;
    UzkgeException(UzkgeError k, const std::string& what) : std::runtime_error(what), kind(k) {}
};
inline void check(int rc) {
    if (rc != UZK_OK) throw UzkgeException(static_cast<UzkgeError>(rc), uzk_last_error());
}

struct Fr {                       // Montgomery limbs, the wire format
    uint64_t l[4] = {0, 0, 0, 0};
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fr& o) const { return std::memcmp(l, o.l, sizeof l) == 0; }
};
using G1Projective = uzk_g1_jac;
using G1Affine = uzk_g1_affine;

class FpPolynomial {
  public:
    std::vector<Fr> coefs;        // low to high

    // Trims trailing zeros; the zero polynomial keeps one zero coefficient (field_polynomial.rs:86-90).
    static FpPolynomial from_coefs(std::vector<Fr> c) {
        while (c.size() > 1 && c.back().is_zero()) c.pop_back();
        if (c.empty()) c.push_back(Fr{});
        FpPolynomial p;
        p.coefs = std::move(c);
        return p;
    }
    const std::vector<Fr>& get_coefs_ref() const { return coefs; }
    size_t degree() const {
        for (size_t i = coefs.size(); i-- > 0;) if (!coefs[i].is_zero()) return i;
        return 0;
    }
    // field_polynomial.rs:554-567
    static std::optional<uint64_t> evaluation_domain(uint64_t num_coeffs) {
        if (num_coeffs == 0 || (num_coeffs & (num_coeffs - 1)) != 0) throw std::invalid_argument("num_coeffs must be 2^k");
        return uzk_domain_supported(num_coeffs) ? std::optional<uint64_t>(num_coeffs) : std::nullopt;
    }
    static std::optional<uint64_t> quotient_evaluation_domain(uint64_t num_coeffs) {
        const uint64_t m = (num_coeffs % 3 == 0) ? num_coeffs / 3 : num_coeffs;
        if (num_coeffs == 0 || (m & (m - 1)) != 0) throw std::invalid_argument("num_coeffs must be 2^k or 3*2^k");
        return uzk_domain_supported(num_coeffs) ? std::optional<uint64_t>(num_coeffs) : std::nullopt;
    }
    // field_polynomial.rs:570-580
    std::optional<std::vector<Fr>> fft(uint64_t num_coeffs) const {
        if (!(num_coeffs > degree())) throw std::invalid_argument("num_coeffs must exceed the degree");
        auto d = (num_coeffs & (num_coeffs - 1)) == 0 ? evaluation_domain(num_coeffs) : quotient_evaluation_domain(num_coeffs);
        if (!d) return std::nullopt;
        return fft_with_domain(*d);
    }
    // `domain.fft(&self.coefs)`: zero-pad to the domain, natural order (field_polynomial.rs:583-586)
    std::vector<Fr> fft_with_domain(uint64_t domain) const { return transform(domain, coefs, 0, nullptr); }
    // fft of p(kX) (field_polynomial.rs:589-591); the serial mul_var is fused into the device transform
    std::vector<Fr> coset_fft_with_domain(uint64_t domain, const Fr& k) const { return transform(domain, coefs, 0, &k); }
    // `domain.ifft(values)` then from_coefs (field_polynomial.rs:594-597)
    static FpPolynomial ifft_with_domain(uint64_t domain, const std::vector<Fr>& values) {
        return from_coefs(transform(domain, values, 1, nullptr));
    }
    // ifft then mul_var(k_inv) (field_polynomial.rs:601-607)
    static FpPolynomial coset_ifft_with_domain(uint64_t domain, const std::vector<Fr>& values, const Fr& k_inv) {
        return from_coefs(transform(domain, values, 1, &k_inv));
    }
    bool operator==(const FpPolynomial& o) const { return from_coefs(coefs).coefs == from_coefs(o.coefs).coefs; }

  private:
    static std::vector<Fr> transform(uint64_t domain, const std::vector<Fr>& in, int inverse, const Fr* shift) {
        if (in.size() > domain) throw std::invalid_argument("more coefficients than the domain holds");
        std::vector<Fr> buf(domain);
        std::copy(in.begin(), in.end(), buf.begin());
        check(uzk_ntt_fr(reinterpret_cast<uint64_t*>(buf.data()), domain, inverse, shift ? shift->l : nullptr));
        return buf;
    }
};

class KZGCommitmentSchemeBN254 {
  public:
    std::vector<G1Affine> public_parameter_group_1;   // affine, Montgomery coordinates

    explicit KZGCommitmentSchemeBN254(std::vector<G1Affine> g1) : public_parameter_group_1(std::move(g1)) {
        check(uzk_srs_register(public_parameter_group_1.data(), public_parameter_group_1.size(), &handle_));
    }
    KZGCommitmentSchemeBN254(const KZGCommitmentSchemeBN254&) = delete;
    KZGCommitmentSchemeBN254& operator=(const KZGCommitmentSchemeBN254&) = delete;
    ~KZGCommitmentSchemeBN254() { if (handle_) (void)uzk_srs_release(handle_); }

    // u32 len_g1 | u32 len_g2 | len_g1 x (x LE32 || y LE32, flags in the top two bits of the last
    // byte) | G2 (kzg_poly_commitment.rs:228-264).  Coordinates are canonical in the file and are
    // converted to Montgomery form on the device.
    static KZGCommitmentSchemeBN254 from_unchecked_bytes(const std::vector<uint8_t>& bytes) {
        if (bytes.size() < 8) throw UzkgeException(UzkgeError::DeserializationError, "short SRS blob");
        uint32_t len1;
        std::memcpy(&len1, bytes.data(), 4);
        if (bytes.size() < 8 + 64ull * len1) throw UzkgeException(UzkgeError::DeserializationError, "truncated G1 section");
        std::vector<uint64_t> canon(8ull * len1), mont(8ull * len1), dummy(8ull * len1, 0);
        std::vector<bool> inf(len1);
        for (uint32_t i = 0; i < len1; ++i) {
            uint8_t rec[64];
            std::memcpy(rec, bytes.data() + 8 + 64ull * i, 64);
            inf[i] = (rec[63] & 0x40) != 0;
            rec[63] &= 0x3F;
            std::memcpy(&canon[8ull * i], rec, 64);
        }
        check(uzk_field_op_device(/*Fq*/ 0, /*to_mont*/ 7, canon.data(), dummy.data(), mont.data(), 2ull * len1));
        std::vector<G1Affine> g1(len1);
        for (uint32_t i = 0; i < len1; ++i) {
            if (inf[i]) { std::memset(&g1[i], 0, sizeof(G1Affine)); continue; }
            std::memcpy(&g1[i], &mont[8ull * i], 64);
        }
        return KZGCommitmentSchemeBN254(std::move(g1));
    }
    size_t max_degree() const { return public_parameter_group_1.size() - 1; }

    // kzg_poly_commitment.rs:278-293
    G1Projective commit(const FpPolynomial& polynomial) const {
        const size_t degree = polynomial.degree();
        if (degree + 1 > public_parameter_group_1.size()) throw UzkgeException(UzkgeError::DegreeError, "degree exceeds the SRS");
        G1Projective out;
        check(uzk_msm_g1(handle_, 0, reinterpret_cast<const uint64_t*>(polynomial.get_coefs_ref().data()), degree + 1, &out));
        return out;
    }
    // C += sum_i b_i * (SRS[i] - SRS[zeroing_degree + i])  (kzg_poly_commitment.rs:299-313)
    G1Projective apply_blind_factors(const G1Projective& commitment, const std::vector<Fr>& blinds, size_t zeroing_degree) const {
        if (blinds.empty()) return commitment;
        G1Projective parts[3] = {commitment, {}, {}};
        std::vector<Fr> neg = fr_neg(blinds);
        check(uzk_msm_g1(handle_, 0, reinterpret_cast<const uint64_t*>(blinds.data()), blinds.size(), &parts[1]));
        check(uzk_msm_g1(handle_, zeroing_degree, reinterpret_cast<const uint64_t*>(neg.data()), neg.size(), &parts[2]));
        G1Projective out;
        check(uzk_g1_fold(parts, 3, &out));
        return out;
    }
    uint64_t handle() const { return handle_; }

    // element-wise field helpers of the host mirrors (device primitives through the KAT entry point)
    static std::vector<Fr> fr_neg(const std::vector<Fr>& a) { return fr_op(5, a, a); }
    static std::vector<Fr> fr_add(const std::vector<Fr>& a, const std::vector<Fr>& b) { return fr_op(1, a, b); }
    static std::vector<Fr> fr_sub(const std::vector<Fr>& a, const std::vector<Fr>& b) { return fr_op(2, a, b); }

  private:
    static std::vector<Fr> fr_op(int op, const std::vector<Fr>& a, const std::vector<Fr>& b) {
        std::vector<Fr> out(a.size());
        if (!a.empty())
            check(uzk_field_op_device(/*Fr*/ 1, op, reinterpret_cast<const uint64_t*>(a.data()),
                                      reinterpret_cast<const uint64_t*>(b.data()), reinterpret_cast<uint64_t*>(out.data()), a.size()));
        return out;
    }
    uint64_t handle_ = 0;
};

// uzkge/src/gen_params/mod.rs:151-183: the monomial SRS of a size-`size` circuit from the embedded blob -- powers
// 0..2050, the identity up to `size`, then the three padding powers size, size + 1, size + 2.
inline KZGCommitmentSchemeBN254* load_srs_params(const std::vector<uint8_t>& srs_blob, size_t size) {
    if (size > 16384) throw UzkgeException(UzkgeError::ParameterError, "size exceeds the embedded SRS");
    std::vector<G1Affine> g1;
    {
        auto full = KZGCommitmentSchemeBN254::from_unchecked_bytes(srs_blob);
        g1 = full.public_parameter_group_1;
    }
    std::vector<G1Affine> out(std::max<size_t>(size + 3, 2051));
    std::memset(out.data(), 0, out.size() * sizeof(G1Affine));
    std::copy(g1.begin(), g1.begin() + 2051, out.begin());
    const size_t pad = size == 4096 ? 2051 : size == 8192 ? 2054 : size == 16384 ? 2057 : 0;
    if (pad) std::copy(g1.begin() + pad, g1.begin() + pad + 3, out.begin() + size);
    return new KZGCommitmentSchemeBN254(std::move(out));
}

// The reference's `for i in (0..=degree).rev() { if (i & (i - 1)) == 0 { .. break } }` (pcs.rs:139-145,
// helpers.rs:1367-1373): the largest power of two <= degree; 0 for degree 0 (the release-build value of the reference's
// loop: 0 & usize::MAX == 0), which the caller turns into the reference's FFT / PCSProveEval error.
inline size_t max_power_of_2(size_t degree) {
    if (degree == 0) return 0;
    size_t p = 1;
    while (p * 2 <= degree) p *= 2;
    return p;
}

// The tail shared by batch_prove (pcs.rs:137-166, degree = q.degree()) and split_t_and_commit
// (helpers.rs:1366-1394, degree = coefs.len()): fold the coefficients from max_power_of_2 on back onto the low
// ones, fft(N), commit the evaluations over the Lagrange SRS, undo the fold with blind factors.
inline G1Projective commit_folded_lagrange(const KZGCommitmentSchemeBN254& pcs, const KZGCommitmentSchemeBN254& lagrange_pcs,
                                           const std::vector<Fr>& coefs, size_t degree) {
    const size_t N = max_power_of_2(degree);
    if (N == 0) throw UzkgeException(UzkgeError::FFTError, "no evaluation domain for a degree-0 polynomial (max_power_of_2 = 0)");
    std::vector<Fr> hi(coefs.begin() + std::min(N, coefs.size()), coefs.end());
    std::vector<Fr> blinds = KZGCommitmentSchemeBN254::fr_neg(hi);
    std::vector<Fr> new_coefs(coefs.begin(), coefs.begin() + std::min(N, coefs.size()));
    if (!hi.empty()) {
        std::vector<Fr> low(new_coefs.begin(), new_coefs.begin() + hi.size());
        std::vector<Fr> sum = KZGCommitmentSchemeBN254::fr_sub(low, blinds);            // coefs[i] - blinds[i]
        std::copy(sum.begin(), sum.end(), new_coefs.begin());
    }
    auto sub_q = FpPolynomial::from_coefs(new_coefs);
    auto q_eval = sub_q.fft(N);
    if (!q_eval) throw UzkgeException(UzkgeError::FFTError, "no evaluation domain for the folded polynomial");
    auto cm = lagrange_pcs.commit(FpPolynomial::from_coefs(*q_eval));
    return pcs.apply_blind_factors(cm, blinds, N);
}

// The commit closure of prover_with_lagrange (prover.rs:125-149; twin in indexer.rs:284-299).
struct ProverCommit {
    const KZGCommitmentSchemeBN254& pcs;
    const KZGCommitmentSchemeBN254* lagrange_pcs;      // null unless it has exactly n_constraints bases
    size_t n_constraints;
    ProverCommit(const KZGCommitmentSchemeBN254& p, const KZGCommitmentSchemeBN254* l, size_t n)
        : pcs(p), lagrange_pcs((l && l->max_degree() + 1 == n) ? l : nullptr), n_constraints(n) {}
    G1Projective operator()(const std::vector<Fr>& evals, const FpPolynomial& coef_polynomial, const std::vector<Fr>& blinds) const {
        if (lagrange_pcs) {
            auto cm = lagrange_pcs->commit(FpPolynomial::from_coefs(evals));
            return pcs.apply_blind_factors(cm, blinds, n_constraints);
        }
        return pcs.commit(coef_polynomial);
    }
};

// The per-table loop of the preprocessing -- indexer_with_lagrange (uzkge/src/plonk/indexer.rs:316-470) and
// refresh_prover_params_public_key (shuffle/src/gen_params/params.rs:88-121): for each evaluation table
//   coefs = ifft_with_domain(domain_n, evals), coset_evals = coefs.coset_fft_with_domain(domain_m, k[1]),
//   cm = commit(evals, coefs)   (no blinds)
// with all tables in one batched inverse transform, one batched coset transform and one batched MSM.
struct PreprocessedTables {
    std::vector<std::vector<Fr>> coefs;          // k x n   (untrimmed, as the quotient kernel wants them)
    std::vector<std::vector<Fr>> coset_evals;    // k x m
    std::vector<G1Projective> cms;               // k
};
inline PreprocessedTables preprocess_tables(const ProverCommit& commit, const std::vector<std::vector<Fr>>& evals, size_t m, const Fr& k1) {
    PreprocessedTables out;
    const size_t k = evals.size(), n = commit.n_constraints;
    if (k == 0) return out;
    if (m % n != 0) throw UzkgeException(UzkgeError::ParameterError, "the quotient domain must be a multiple of n");
    if (!uzk_domain_supported(n) || !uzk_domain_supported(m)) throw UzkgeException(UzkgeError::FFTError, "no evaluation domain of that size");
    std::vector<Fr> flat(k * n), wide(k * m);
    for (size_t i = 0; i < k; ++i) {
        if (evals[i].size() != n) throw UzkgeException(UzkgeError::ParameterError, "every table has n_constraints rows");
        std::copy(evals[i].begin(), evals[i].end(), flat.begin() + i * n);
    }
    out.cms.resize(k);
    if (commit.lagrange_pcs)
        check(uzk_msm_g1_batch(commit.lagrange_pcs->handle(), 0, reinterpret_cast<const uint64_t*>(flat.data()), n, (uint32_t)k, out.cms.data()));
    check(uzk_ntt_fr_batch(reinterpret_cast<uint64_t*>(flat.data()), n, (uint32_t)k, /*inverse*/ 1, nullptr));
    if (!commit.lagrange_pcs) {
        if (n > commit.pcs.public_parameter_group_1.size()) throw UzkgeException(UzkgeError::DegreeError, "degree exceeds the SRS");
        check(uzk_msm_g1_batch(commit.pcs.handle(), 0, reinterpret_cast<const uint64_t*>(flat.data()), n, (uint32_t)k, out.cms.data()));
    }
    for (size_t i = 0; i < k; ++i) std::copy(flat.begin() + i * n, flat.begin() + (i + 1) * n, wide.begin() + i * m);
    check(uzk_ntt_fr_batch(reinterpret_cast<uint64_t*>(wide.data()), m, (uint32_t)k, /*inverse*/ 0, k1.l));
    out.coefs.resize(k);
    out.coset_evals.resize(k);
    for (size_t i = 0; i < k; ++i) {
        out.coefs[i].assign(flat.begin() + i * n, flat.begin() + (i + 1) * n);
        out.coset_evals[i].assign(wide.begin() + i * m, wide.begin() + (i + 1) * m);
    }
    return out;
}

}  // namespace uzkge
