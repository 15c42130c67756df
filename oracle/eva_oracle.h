/*
 * eva_oracle.h — CPU restatement of the arithmetic behind EVA's execute() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under eva_amd/ may include, link or load this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker — never as the thing measured as the product or shipped.
 *
 * PARITY UNPINNED (vs Microsoft SEAL bits): the arithmetic the reference calls lives in
 * microsoft/SEAL v3.6.x (pinned at /root/reference/README.md:28-30, CMakeLists.txt:24), which
 * is not present in /root/reference nor installable here, and the reference's own tests hold
 * no ciphertext-level golden vectors (tests/common.py:34 is an MSE threshold).  This file
 * restates SEAL 3.6's published algorithms (Evaluator::{add,sub,negate,multiply,square,
 * multiply_plain,add_plain,sub_plain,relinearize,rotate_vector,rescale_to_next,
 * mod_switch_to_next}, CoeffModulus::Create, NTTTables, GaloisTool) at the call sites
 * /root/reference/eva/seal/seal_executor.h:114-243.  Every stored value is a canonical
 * residue in [0,q), so any correct implementation with the same primes / psi / ordering /
 * rounding is bit-identical.  It is pinned by: algebraic known-answer tests (NTT vs O(N^2)
 * evaluation, rescale and key-switch mod-down vs exact big-integer CRT, schoolbook negacyclic
 * product), the prime/psi constants of SURVEY.md Appendix B, the compiler prime_bits KATs and
 * the reference's statistical oracle (MSE < 0.01) on the reference's own programs.
 */
#ifndef EVA_ORACLE_H
#define EVA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct evo_ctx evo_ctx;

/* ---- number theory (SEAL util/numth, modulus) ---- */
int evo_is_prime(uint64_t v);
/* CoeffModulus::Create(N, bit_sizes) — called at /root/reference/eva/seal/seal.cpp:181-182.
 * returns 0 on success; out[n_bits] filled in bit_sizes order. */
int evo_coeff_modulus_create(uint32_t N, const int *bit_sizes, uint32_t n_bits, uint64_t *out);
/* numerically smallest primitive 2N-th root of unity mod q (SEAL try_minimal_primitive_root) */
uint64_t evo_minimal_primitive_root(uint32_t N, uint64_t q);
uint64_t evo_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t evo_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t evo_invmod(uint64_t a, uint64_t q);

/* ---- context: N, the key-level prime chain (special prime last), NTT tables ---- */
evo_ctx *evo_ctx_create(uint32_t N, uint32_t k, const uint64_t *primes);
void evo_ctx_destroy(evo_ctx *c);
uint32_t evo_ctx_n(const evo_ctx *c);
uint32_t evo_ctx_k(const evo_ctx *c);
uint64_t evo_ctx_prime(const evo_ctx *c, uint32_t i);
uint64_t evo_ctx_psi(const evo_ctx *c, uint32_t i);
/* forward table rp[br(i)] = psi^i ; inverse table irp[j] = rp[j]^-1 (same indexing) */
const uint64_t *evo_ctx_root_powers(const evo_ctx *c, uint32_t i);
const uint64_t *evo_ctx_inv_root_powers(const evo_ctx *c, uint32_t i);

/* negacyclic NTT, natural -> bit-reversed (SEAL ntt_negacyclic_harvey); canonical output */
void evo_ntt(const evo_ctx *c, uint32_t prime_idx, uint64_t *x);
/* inverse, bit-reversed -> natural, scaled by N^-1; canonical output */
void evo_intt(const evo_ctx *c, uint32_t prime_idx, uint64_t *x);

/* ---- evaluator.  Ciphertext layout: [size][l][N] uint64, limb i is mod primes[i].
 *      l = number of data limbs at the ciphertext's level (l <= k-1). ---- */
/* evaluator.add / sub (seal_executor.h:124,140).  out has max(sa,sb) polys. */
void evo_add(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, const uint64_t *b,
             uint32_t sb, uint64_t *out);
void evo_sub(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, const uint64_t *b,
             uint32_t sb, uint64_t *out);
/* evaluator.add_plain / sub_plain (seal_executor.h:127,143). pt is [l][N]. */
void evo_add_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                   const uint64_t *pt, uint64_t *out);
void evo_sub_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                   const uint64_t *pt, uint64_t *out);
/* evaluator.negate (seal_executor.h:194) */
void evo_negate(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out);
/* evaluator.multiply 2x2 -> 3 (seal_executor.h:164) */
void evo_multiply(const evo_ctx *c, uint32_t l, const uint64_t *a, const uint64_t *b,
                  uint64_t *out3);
/* evaluator.square 2 -> 3 (seal_executor.h:162) */
void evo_square(const evo_ctx *c, uint32_t l, const uint64_t *a, uint64_t *out3);
/* evaluator.multiply_plain (seal_executor.h:168) */
void evo_multiply_plain(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa,
                        const uint64_t *pt, uint64_t *out);
/* evaluator.rescale_to_next (seal_executor.h:213): in [s][l][N] -> out [s][l-1][N] */
void evo_rescale(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out);
/* evaluator.mod_switch_to_next (seal_executor.h:206): drop the last limb */
void evo_mod_switch(const evo_ctx *c, uint32_t l, const uint64_t *a, uint32_t sa, uint64_t *out);
/* SEAL Evaluator::switch_key_inplace: ct (size 2, [2][l][N]) += keyswitch(target [l][N]).
 * key layout: [l_key_digits >= l][2][k][N] (digit-major; each digit a size-2 ciphertext over
 * all k key primes).  n_digits_stride = k-1 normally. */
void evo_switch_key(const evo_ctx *c, uint32_t l, uint64_t *ct2, const uint64_t *target,
                    const uint64_t *key);
/* evaluator.relinearize 3 -> 2 (seal_executor.h:200) */
void evo_relinearize(const evo_ctx *c, uint32_t l, const uint64_t *a3, const uint64_t *relin_key,
                     uint64_t *out2);
/* Galois element for rotate_vector(steps) (SEAL GaloisTool::get_elt_from_step) */
uint32_t evo_galois_elt_from_step(uint32_t N, int32_t steps);
/* NTT-domain automorphism permutation table: out[i] = in[table[i]] */
void evo_galois_table(uint32_t N, uint32_t galois_elt, uint32_t *table);
/* evaluator.rotate_vector (seal_executor.h:181,188); steps==0 -> copy. size-2 in/out */
void evo_rotate(const evo_ctx *c, uint32_t l, const uint64_t *a2, int32_t steps,
                const uint64_t *galois_key, uint64_t *out2);

/* ---- CKKSEncoder::encode restatement (seal_executor.h:217-243): values[N/2] (already
 * replicated) -> plaintext [l][N] in NTT form.  returns 0 ok, -1 too large. ---- */
int evo_encode(const evo_ctx *c, uint32_t l, const double *values, double scale, uint64_t *pt);
/* coefficient form only (before per-limb NTT); signed coefficients as doubles (rounded) */
void evo_encode_coeffs(uint32_t N, const double *values, double scale, double *coeffs);

/* ---- Decryptor::decrypt + CKKSEncoder::decode restatement (/root/reference/eva/seal/seal.cpp:124-146) ----
 * decrypt: ct [size][l][N], sk_ntt [k][N] -> plaintext [l][N], NTT form (c0 + c1 s + c2 s^2).
 * decode : plaintext [l][N] NTT form at `scale` -> the N/2 slot values, SEAL 3.6's FP64 operation order
 * (CRT composition to base-2^64 words, words to double least significant first with 1/scale folded in,
 * transform_to_rev).  returns 0, -1 for "scale out of bounds". */
void evo_decrypt(const evo_ctx *c, uint32_t l, uint32_t size, const uint64_t *ct, const uint64_t *sk_ntt, uint64_t *pt);
int evo_decode(const evo_ctx *c, uint32_t l, const uint64_t *pt, double scale, double *out);

/* op-triple used by bench.py's cpu_baseline: multiply + relinearize + rescale */
void evo_op_triple(const evo_ctx *c, uint32_t l, const uint64_t *a2, const uint64_t *b2,
                   const uint64_t *relin_key, uint64_t *out2 /* [2][l-1][N] */);

/* ---- eva_oracle_dag.c: walk of a compiled program's encrypted part (the CPU baseline of the DAG
 * configurations).  op codes are the reference's (eva/ir/ops.h:11-25); Input / Constant / Encode
 * entries mark caller-placed slots.  vals[v].kind: 0 empty, 1 ciphertext [size][limbs][N],
 * 2 plaintext [limbs][N].  Slots written by the walk get malloc'ed data (release with
 * evo_dag_free unless `alias` is set: an Output shares its operand's data); intermediates are
 * freed at their last use.  threads <= 1: ProgramTraversal::forwardPass; > 1: the
 * dependency-counting MulticoreProgramTraversal analogue on that many pthreads. */
typedef struct { uint32_t op, dst, src0, src1; int32_t imm; } evo_dag_op;
typedef struct { uint32_t kind, size, limbs, alias; uint64_t *data; } evo_dag_val;
int evo_dag_walk(const evo_ctx *c, const evo_dag_op *ops, uint32_t n_ops, evo_dag_val *vals, uint32_t n_vals,
                 const uint64_t *relin_key, const uint32_t *galois_elts, const uint64_t *const *galois_keys,
                 uint32_t n_galois, int threads);
void evo_dag_free(uint64_t *data);

#ifdef __cplusplus
}
#endif
#endif
