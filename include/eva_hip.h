/*
 * eva_hip.h — C-ABI of libeva_hip.so: the MI355X (gfx950) CKKS evaluation backend that
 * replaces the SEAL Evaluator calls behind EVA's execute() path.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference).  Conventions:
 *   - every function returns 0 on success, non-zero on error; evah_last_error() gives the
 *     message (the reference throws C++ exceptions: seal_executor.h:130,146,171,402 and SEAL's
 *     invalid_argument / logic_error; the host wrapper rethrows std::runtime_error).
 *   - handles are opaque; device memory is owned by the backend; uploads and downloads copy.
 *   - a ciphertext is `size` polynomials x `limbs` RNS limbs x N uint64 residues in NTT form,
 *     poly-major / limb-major / coefficient-contiguous (SEAL's layout); limb i is mod primes[i].
 *     limbs = (k-1) - level, where level is EVA's EncodeAtLevelAttribute / signature level
 *     (seal_executor.h:221-224, seal.cpp:59-62).
 *   - one evah_ctx is driven from one host thread at a time; distinct contexts are independent.
 */
#ifndef EVA_HIP_H
#define EVA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct evah_ctx evah_ctx; /* replaces seal::SEALContext + seal::Evaluator (seal.h:58-66) */
typedef struct evah_ct evah_ct;   /* replaces seal::Ciphertext in SEALExecutor::Objects (seal_executor.h:32-42) */
typedef struct evah_pt evah_pt;   /* replaces seal::Plaintext  in SEALExecutor::Objects */
typedef struct evah_graph evah_graph; /* a captured execute(): replaces re-walking the DAG per call */

/* Last error message of the calling thread ("" if none). */
const char *evah_last_error(void);
/* ABI version of this header. */
int evah_abi_version(void);
/* Number of visible HIP devices. */
int evah_device_count(int *count);

/* ---- context -------------------------------------------------------------------------------
 * Replaces getSEALContext(params) + seal::Evaluator(context) (seal.cpp:148-172, seal.h:52).
 * primes: the key-level chain in CoeffModulus::Create order, special prime last (seal.cpp:181).
 * The backend derives psi (minimal primitive 2N-th root) and all NTT tables itself. */
int evah_ctx_create(uint32_t poly_degree, uint32_t n_primes, const uint64_t *primes, int device,
                    evah_ctx **out);
/* A second issue queue on the same device state: shares the parent's tables and keys (uploaded
 * through either), owns its own HIP stream and buffer pool.  Independent DAG nodes / independent
 * programs issued through different forks overlap on the GPU — the stream-level counterpart of
 * the reference's Galois worker threads (multicore_program_traversal.h:55-78).  A handle may be
 * read by any fork once the producing fork has been synchronised.  Values produced through a
 * fork live in that fork's pool: free them before destroying it (values it only read may outlive it). */
int evah_ctx_fork(evah_ctx *parent, evah_ctx **out);
void evah_ctx_destroy(evah_ctx *ctx);
/* Launch on an external HIP stream (hipStream_t as void*); NULL restores the context's own. */
int evah_ctx_set_stream(evah_ctx *ctx, void *hip_stream);
/* Block until all work issued on the context's stream has finished. */
int evah_ctx_sync(evah_ctx *ctx);
/* *busy = 1 while work enqueued on this queue has not finished (hipStreamQuery), without waiting: how the host side of
 * execute() (seal.cpp:104-122) notices that the caller issued a call while the previous one is still running (r6: twin
 * graph plans) */
int evah_ctx_busy(evah_ctx *ctx, int *busy);
/* Bytes currently held in the context's device pool (in use + cached). */
int evah_ctx_mem_info(evah_ctx *ctx, size_t *in_use, size_t *cached);

/* ---- keys ----------------------------------------------------------------------------------
 * Replaces the seal::RelinKeys / seal::GaloisKeys members of SEALPublic (seal.h:62-63).
 * data: [n_digits][2][n_primes][N] uint64, NTT form (one size-2 key-level ciphertext per digit;
 * SEAL KSwitchKeys layout).  galois_elt is ignored for the relinearization key.
 * The caller always passes the whole key.  On a context that is a limb shard (evah_ctx_set_shard called
 * BEFORE the upload) only the prime rows that shard multiplies into are kept in HBM — its own data limbs
 * and the special prime — and the context then serves the evah_shard_* entry points only. */
#define EVAH_KEY_RELIN 0
#define EVAH_KEY_GALOIS 1
#define EVAH_KEY_PUBLIC 2 /* evah_client_key_upload */
#define EVAH_KEY_SECRET 3
int evah_key_upload(evah_ctx *ctx, int kind, uint32_t galois_elt, uint32_t n_digits,
                    const uint64_t *data);
/* Galois element used by evah_rotate for `steps` (SEAL GaloisTool::get_elt_from_step). */
int evah_galois_elt_from_step(evah_ctx *ctx, int32_t steps, uint32_t *elt);

/* ---- values --------------------------------------------------------------------------------
 * Replace SEALExecutor::setInputs / getOutputs / free (seal_executor.h:264-277,420-435,406-418). */
int evah_ct_upload(evah_ctx *ctx, uint32_t size, uint32_t limbs, double scale,
                   const uint64_t *data /* [size][limbs][N] */, evah_ct **out);
/* ---- pinned host memory for the values setInputs copies in and getOutputs copies out
 * (seal_executor.h:269-270, 420-435; in the reference they live in SEAL's MemoryPool): blocks are
 * page-locked once and recycled by size, so the copies are DMA transfers at PCIe rate.  NULL when
 * no memory can be pinned (no device) — use ordinary memory then.  Thread-safe. */
void *evah_host_alloc(size_t bytes);
void evah_host_free(void *p);

/* ---- batched handles: `batch` (<= 64) independent ciphertexts of one shape in ONE handle
 * ([batch][size][limbs][N]).  Every evaluator entry point above/below accepts them and applies the
 * SEAL call to each instance in one launch set (plaintext operands and keys are shared by the
 * instances; two ciphertext operands must have the same batch).  This is the unit a batch of
 * independent program instances runs on (BASELINE config 4: 256 Sobel DAGs) — the reference runs
 * such a batch as `batch` separate SEALPublic::execute calls (seal.cpp:104-122). */
int evah_ct_upload_batch(evah_ctx *ctx, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                         const uint64_t *data /* [batch][size][limbs][N] */, evah_ct **out);
/* the same from `batch` separate host arrays, each [size][limbs][N] */
int evah_ct_upload_instances(evah_ctx *ctx, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                             const uint64_t *const *data, evah_ct **out);
/* every instance of a batched handle into its own host array out[b] ([size][limbs][N] each) */
int evah_ct_download_instances(evah_ctx *ctx, const evah_ct *ct, uint64_t *const *out);
/* Stream-ordered forms of the two calls above, for pipelining a loop of SEALPublic::execute calls
 * (seal.cpp:104-122) over several queues: the copies are enqueued on ctx's queue and the call returns.
 * Host arrays must stay valid and untouched until evah_ctx_sync(ctx); use evah_host_alloc memory for
 * copies that overlap other queues' kernels.  The handle passed to the download may be freed right away. */
int evah_ct_upload_instances_async(evah_ctx *ctx, uint32_t batch, uint32_t size, uint32_t limbs, double scale,
                                   const uint64_t *const *data, evah_ct **out);
int evah_ct_download_instances_async(evah_ctx *ctx, const evah_ct *ct, uint64_t *const *out);
int evah_ct_batch(const evah_ct *ct, uint32_t *batch);
/* n single ciphertexts (same shape, scale) -> one batched handle; device-side copies */
int evah_ct_stack(evah_ctx *ctx, const evah_ct *const *cts, uint32_t n, evah_ct **out);
/* instance b of a batched handle as a single-ciphertext view (shares the buffer; free separately) */
int evah_ct_unstack(evah_ctx *ctx, const evah_ct *ct, uint32_t b, evah_ct **out);

/* a copy of a value owned by another context: another GPU (peer copy over xGMI — the P2P step at the
 * join of independent sub-DAGs) or another issue queue; asynchronous, ordered
 * after the producer of src on its own queue */
int evah_ct_copy(evah_ctx *dst_ctx, const evah_ct *src, evah_ct **out);
int evah_pt_copy(evah_ctx *dst_ctx, const evah_pt *src, evah_pt **out);
/* overwrite an existing handle's residues (same shape) — refills the input slots of a graph */
int evah_ct_write(evah_ctx *ctx, evah_ct *ct, const uint64_t *data);
int evah_pt_write(evah_ctx *ctx, evah_pt *pt, const uint64_t *data);
/* device-to-device refill of an existing handle from another one of the same shape (the input slot of a
 * captured execute() from a device-resident valuation entry: seal_executor.h:264-277 copies inputs in,
 * here without leaving the device); asynchronous on ctx's queue, ordered after the producer of src */
int evah_ct_assign(evah_ctx *ctx, evah_ct *dst, const evah_ct *src);
/* `waiter`'s queue waits for everything enqueued so far on `signaller`'s queue (both of one device
 * state); no host synchronisation.  Orders work the per-buffer tracking cannot see (graph replays). */
int evah_ctx_wait(evah_ctx *waiter, evah_ctx *signaller);
/* values that crossed the host boundary through this context and its forks since creation:
 * out[0] / out[1] = ciphertexts uploaded (evah_ct_upload*, evah_ct_write) / downloaded, out[2] / out[3] =
 * plaintexts uploaded / downloaded, out[4] / out[5] = bytes host->device / device->host of all of them.
 * The valuation of the reference "may hold device handles" (SURVEY.md 8(b)); tests assert that
 * encrypt -> execute -> decrypt moves no ciphertext across. */
int evah_ctx_transfer_stats(evah_ctx *ctx, uint64_t out[6]);
/* bytes of HBM the evaluation keys (relinearization + Galois) of this context's device state occupy.  After
 * evah_ctx_set_shard(ctx, s, G) an upload keeps only shard s's prime rows of a key (its data limbs and the
 * special prime): (ceil((k-1)/G) + 1) / k of the whole key. */
int evah_ctx_key_bytes(evah_ctx *ctx, uint64_t *bytes);
/* The same keys with the copies the library keeps beside them (the reference keeps one copy per key, seal.h:58-66):
 * out[0] = the key words as uploaded (= evah_ctx_key_bytes), out[1] = the radix-2^30 split copies of whole keys on
 * contexts whose primes all have the top-bit shape (ks_inner_kernel<MAC3>), out[2] = the permuted copies of Galois
 * keys that hoisted rotation sets have used so far.  Up to 3x out[0]; a copy that does not fit the device is simply
 * not made (the affected launches take the form that does not need it). */
int evah_ctx_key_bytes_detail(evah_ctx *ctx, uint64_t out[3]);
int evah_ct_info(const evah_ct *ct, uint32_t *size, uint32_t *limbs, double *scale);
int evah_ct_download(evah_ctx *ctx, const evah_ct *ct, uint64_t *out /* [size][limbs][N] */);
void evah_ct_free(evah_ctx *ctx, evah_ct *ct);
/* data in NTT form */
int evah_pt_upload(evah_ctx *ctx, uint32_t limbs, double scale, const uint64_t *data /* [limbs][N] */,
                   evah_pt **out);
/* data in coefficient form (output of the host FP64 encoder); the backend runs the per-limb
 * forward NTT — the device half of CKKSEncoder::encode (seal_executor.h:242). */
int evah_pt_upload_coeff(evah_ctx *ctx, uint32_t limbs, double scale, const uint64_t *data,
                         evah_pt **out);
/* encoder.encode (seal_executor.h:242) entirely on the device: `n_values` reals (replicated over
 * the N/2 slots as seal_executor.h:226-240 does) -> inverse special FFT in FP64 -> round(x*scale/N)
 * -> residues -> NTT.  Bit-identical to the host encoder followed by evah_pt_upload_coeff; valid when
 * every rounded coefficient is below 2^62 in magnitude (the caller checks a bound), otherwise use
 * the host's multi-precision path */
int evah_pt_encode(evah_ctx *ctx, const double *values, uint32_t n_values, uint32_t limbs, double scale, evah_pt **out);
/* plaintext whose every slot is the same residue per limb: encode of a uniform constant
 * (Program::makeUniformConstant, program.h:58-60) — value[i] = round(c*scale) mod primes[i]. */
int evah_pt_uniform(evah_ctx *ctx, uint32_t limbs, double scale, const uint64_t *value /* [limbs] */,
                    evah_pt **out);
int evah_pt_info(const evah_pt *pt, uint32_t *limbs, double *scale);
int evah_pt_download(evah_ctx *ctx, const evah_pt *pt, uint64_t *out /* [limbs][N] */);
void evah_pt_free(evah_ctx *ctx, evah_pt *pt);

/* ---- evaluator ops: one per SEAL call made by SEALExecutor::operator() ---------------------- */
/* evaluator.add (seal_executor.h:124); sizes may differ (2/3), limbs and scale must match */
int evah_add(evah_ctx *ctx, const evah_ct *a, const evah_ct *b, evah_ct **out);
/* evaluator.sub (seal_executor.h:140) */
int evah_sub(evah_ctx *ctx, const evah_ct *a, const evah_ct *b, evah_ct **out);
/* evaluator.add_plain (seal_executor.h:127) */
int evah_add_plain(evah_ctx *ctx, const evah_ct *a, const evah_pt *b, evah_ct **out);
/* evaluator.sub_plain (seal_executor.h:143) */
int evah_sub_plain(evah_ctx *ctx, const evah_ct *a, const evah_pt *b, evah_ct **out);
/* evaluator.negate (seal_executor.h:194) */
int evah_negate(evah_ctx *ctx, const evah_ct *a, evah_ct **out);
/* evaluator.multiply, both operands size 2 (seal_executor.h:164) */
int evah_multiply(evah_ctx *ctx, const evah_ct *a, const evah_ct *b, evah_ct **out);
/* n (<= 64) independent evaluator.multiply calls at one level (seal_executor.h:164, once per
 * independent Multiply node / per ciphertext of a batch) issued as ONE launch; outs[i] ==
 * evah_multiply(as[i], bs[i]) bit for bit, the outputs share one allocation */
int evah_multiply_many(evah_ctx *ctx, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n, evah_ct **outs);
/* sum_j cts[j] (*) pts[j] in one pass (pts[j] == NULL: cts[j] itself): the value of the
 * evaluator.multiply_plain (seal_executor.h:168) + evaluator.add (:124) chain a convolution or
 * linear-layer row lowers to; n <= 64 terms of one size, level and product scale */
int evah_weighted_sum(evah_ctx *ctx, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **out);
/* evaluator.square, operand size 2 (seal_executor.h:162) */
int evah_square(evah_ctx *ctx, const evah_ct *a, evah_ct **out);
/* evaluator.multiply_plain (seal_executor.h:168) */
int evah_multiply_plain(evah_ctx *ctx, const evah_ct *a, const evah_pt *b, evah_ct **out);
/* n (<= 64) independent multiply_plain calls (seal_executor.h:168) on single ciphertexts of one shape
 * as one launch; outs[i] == evah_multiply_plain(cts[i], pts[i]) */
int evah_multiply_plain_many(evah_ctx *ctx, const evah_ct *const *cts, const evah_pt *const *pts, uint32_t n, evah_ct **outs);
/* evaluator.relinearize, size 3 -> 2 (seal_executor.h:200); needs the relin key */
int evah_relinearize(evah_ctx *ctx, const evah_ct *a, evah_ct **out);
/* evaluator.relinearize immediately followed by evaluator.rescale_to_next (+ scale fix-up) on the
 * result (seal_executor.h:200 then :213-214), evaluated together: identical ciphertext, ~13% fewer
 * transforms.  The host executor uses it when a Relinearize term's only use is a Rescale. */
int evah_relinearize_rescale(evah_ctx *ctx, const evah_ct *a, uint32_t divisor_bits, evah_ct **out);
/* the same for n (<= 64) independent ciphertexts at one level (a batch of multiplications to be
 * relinearized and rescaled): outs[b] == evah_relinearize_rescale(as[b]); one wide launch set,
 * the shared relinearization key is streamed once per XCD for the whole batch. */
int evah_relinearize_rescale_many(evah_ctx *ctx, const evah_ct *const *as, uint32_t n, uint32_t divisor_bits,
                                  evah_ct **outs);
/* evaluator.multiply (seal_executor.h:164), evaluator.relinearize (:200) and
 * evaluator.rescale_to_next + scale fix-up (:213-214) on one pair of size-2 ciphertexts — the
 * Mul -> Relinearize -> Rescale chain every ciphertext product of a compiled CKKS program goes
 * through — evaluated together: the size-3 product is never written to HBM (its polynomials are
 * formed where the key switch and the combine consume them).  Identical ciphertext to the three calls. */
int evah_multiply_relinearize_rescale(evah_ctx *ctx, const evah_ct *a, const evah_ct *b, uint32_t divisor_bits, evah_ct **out);
/* the same for n (<= 64) independent pairs at one level as one launch set; BASELINE.json's
 * op-triple (multiply + relinearize + rescale) is exactly one unit of this call */
int evah_multiply_relinearize_rescale_many(evah_ctx *ctx, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n,
                                           uint32_t divisor_bits, evah_ct **outs);
/* evaluator.multiply (seal_executor.h:164; a == b: evaluator.square, :162), evaluator.rescale_to_next + scale fix-up
 * (:213-214) on the size-3 product and evaluator.relinearize (:200) — the order lazy relinearization
 * (eva/ckks/lazy_relinearizer.h:73-80) gives a product under the waterline rescalers — evaluated together: the
 * size-3 product is never written; the rescaled d2 is formed in coefficient form where the digit decomposition reads it, and
 * the rescale of d0 / d1 shares the forward transform of the key switch's mod-down (six dependent launches).
 * Identical ciphertext to the three calls.  Batched handles: a and b hold the same number of instances, so does *out. */
int evah_multiply_rescale_relinearize(evah_ctx *ctx, const evah_ct *a, const evah_ct *b, uint32_t divisor_bits, evah_ct **out);
/* the same for n independent pairs at one level as one launch set (n x instances per handle <= 64) */
int evah_multiply_rescale_relinearize_many(evah_ctx *ctx, const evah_ct *const *as, const evah_ct *const *bs, uint32_t n,
                                           uint32_t divisor_bits, evah_ct **outs);
/* evaluator.rescale_to_next + scale fix-up (seal_executor.h:213-214) of a size-3 ciphertext followed by evaluator.relinearize
 * (:200): what lazy relinearization (eva/ckks/lazy_relinearizer.h:73-80) leaves after a SUM of products.  Evaluated together
 * like evah_multiply_rescale_relinearize, on the stored polynomials.  Identical ciphertext to the two calls.  Batched handles
 * keep their instance count. */
int evah_rescale_relinearize(evah_ctx *ctx, const evah_ct *a, uint32_t divisor_bits, evah_ct **out);
/* the same for n size-3 ciphertexts at one level as one launch set (n x instances per handle <= 64) */
int evah_rescale_relinearize_many(evah_ctx *ctx, const evah_ct *const *as, uint32_t n, uint32_t divisor_bits, evah_ct **outs);
/* evaluator.rotate_vector(a, steps) (seal_executor.h:181; rightRotate passes -steps, :188);
 * steps == 0 copies; needs the Galois key for exactly this step's element */
int evah_rotate(evah_ctx *ctx, const evah_ct *a, int32_t steps, evah_ct **out);
/* n (<= 64) non-zero rotations of the SAME ciphertext — n evaluator.rotate_vector calls
 * (seal_executor.h:181) issued as one set of n-times-wider launches; outs[r] == evah_rotate(a, steps[r]).
 * The host executor groups the sibling rotations of a term (convolution windows) into one call.
 * Throughput-sized sets are hoisted: c1 is decomposed once and every rotation's key inner product is
 * formed from the permuted digits plus a per-(element, level) constant — the residues are exactly
 * those of rotating first and decomposing after, at 1/n of the transforms.
 * EVAH_HOIST=0 disables, EVAH_HOIST_MIN_TILES sets the size from which it is used. */
int evah_rotate_many(evah_ctx *ctx, const evah_ct *a, const int32_t *steps, uint32_t n, evah_ct **outs);
/* n (<= 64) rotations of SEVERAL ciphertexts of one level, pair i = (cts[i], steps[i] != 0): the
 * sibling rotations of independent sub-expressions (seal_executor.h:181/188 called once per node)
 * as one launch set; outs[i] == evah_rotate(cts[i], steps[i]) bit for bit.  A ciphertext that occurs
 * in several pairs is decomposed once for all of them when the set is large enough (hoisting, as above). */
int evah_rotate_pairs(evah_ctx *ctx, const evah_ct *const *cts, const int32_t *steps, uint32_t n, evah_ct **outs);
/* Sums of plaintext-weighted rotations, the convolution pattern of EVA programs
 * (/root/reference/examples/image_processing.py:22-34 convolutionXY: rotated = image << (i*w + j);
 *  Ix += rotated * filter[i][j]; Iy += rotated * filter[j][i]) — the rotate_vector (seal_executor.h:181/188),
 * multiply_plain (:168) and add (:124) calls of every term in one launch set.
 *   window w has win_terms[w] terms (cts / steps, consecutive; steps == 0: the ciphertext itself) and
 *   win_sums[w] sums over those terms; pts holds, window after window, win_sums[w] rows of win_terms[w]
 *   weights (NULL = 1); outs receives the sums of all windows in order:
 *     out = sum_t pts[row][t] (*) rotate(cts[t], steps[t])        — bit for bit the op-by-op result.
 * All ciphertexts: size 2, one level, one batch count; every sum has one product scale (checked as
 * evah_weighted_sum checks it).  Throughput-sized sets whose windows have one or two sums and at most one
 * unrotated term are hoisted AND fused: the rotated ciphertexts are never written — the mod-down's last pass
 * multiplies by the weights and accumulates the sums in registers; other shapes run as
 * evah_rotate_pairs / evah_rotate_many followed by evah_weighted_sum.  EVAH_WIN_FUSE=0 forces the latter. */
int evah_rotate_weighted_sums(evah_ctx *ctx, const evah_ct *const *cts, const int32_t *steps, const uint32_t *win_terms,
                              const uint32_t *win_sums, uint32_t n_windows, const evah_pt *const *pts, evah_ct **outs);
/* n independent evaluator.rescale_to_next calls (seal_executor.h:213) of one size and level,
 * n * size <= 128, as one launch set */
int evah_rescale_many(evah_ctx *ctx, const evah_ct *const *cts, uint32_t n, uint32_t divisor_bits, evah_ct **outs);
/* n (<= 64) independent evaluator.relinearize calls (seal_executor.h:200) of one level as one launch set */
int evah_relinearize_many(evah_ctx *ctx, const evah_ct *const *cts, uint32_t n, evah_ct **outs);
/* evaluator.rescale_to_next + scale fix-up out.scale = a.scale / 2^divisor_bits
 * (seal_executor.h:213-214) */
int evah_rescale(evah_ctx *ctx, const evah_ct *a, uint32_t divisor_bits, evah_ct **out);
/* evaluator.mod_switch_to_next (seal_executor.h:206) */
int evah_mod_switch(evah_ctx *ctx, const evah_ct *a, evah_ct **out);

/* ---- the neighbours of execute() on the device --------------------------
 * SEALPublic::encrypt (seal.cpp:24-102: encoder.encode + encryptor.encrypt) and SEALSecret::decrypt
 * (seal.cpp:124-146: decryptor.decrypt + encoder.decode).  Randomness is the caller's (a host CSPRNG):
 * `small` holds the ternary u and the two error polynomials as int8 [3][N].  Keys: the public key
 * [2][k][N] and the secret key in NTT form [k][N], both uploaded with evah_client_key_upload. */
int evah_client_key_upload(evah_ctx *ctx, int kind /* EVAH_KEY_PUBLIC | EVAH_KEY_SECRET */, const uint64_t *data);
/* (pk0 u + e0, pk1 u + e1) one level above pt, divided-and-rounded by the extra prime, + pt on c0 */
int evah_encrypt(evah_ctx *ctx, const evah_pt *pt, const int8_t *small, evah_ct **out);
/* m = c0 + c1 s (+ c2 s^2) -> inverse transform -> exact recomposition -> double (1/scale folded in) ->
 * special FFT, in the operation order of SEAL 3.6's CKKSEncoder::decode_internal: the first n_out slot
 * values, the same doubles as the host decoder and the CPU oracle produce */
int evah_decrypt_decode(evah_ctx *ctx, const evah_ct *ct, uint32_t n_out, double *out);

/* ---- whole-DAG submit: a topologically sorted flat op list over a value table.
 * One call replaces the per-node loop ProgramTraversal::forwardPass + SEALExecutor::operator()
 * (program_traversal.h:36-93, seal_executor.h:279-404) for the encrypted part of a program.
 *   op    : the reference's Op codes (eva/ir/ops.h:11-25): Negate 10, Add 11, Sub 12, Mul 13,
 *           RotateLeftConst 14, RotateRightConst 15, Relinearize 20, ModSwitch 21, Rescale 22,
 *           Output 2 (dst names the same ciphertext as src0); Input 1 / Constant 3 / Encode 23 are
 *           accepted as markers of caller-placed slots
 *   imm   : rotation steps (14/15), rescale divisor bits (22)
 *   flags : EVAH_OPF_FREE_SRC0/1 = the caller does not need that operand afterwards: it is released
 *           once its last reader in the list has run (the reference frees at last use under
 *           Galois, multicore_program_traversal.h:62-67)
 * Single assignment: every dst slot starts empty and is written by exactly one op; inputs and
 * encoded plaintexts (Encode / Constant nodes) are placed in the table by the caller; on return
 * the slots written by ops hold new handles owned by the caller (unless released by a flag).
 * Operand kinds select the evaluator call exactly as seal_executor.h:114-175 does: Add/Mul swap a
 * plaintext first operand behind the ciphertext, Mul with src0 == src1 is square, Sub needs a
 * ciphertext first.
 * Scheduling inside (same ciphertexts as running the list op by op): ops are taken level by level
 * (depth from the caller-placed values — what MulticoreProgramTraversal's ready set holds), the
 * independent rotations / rescales / relinearizations / ciphertext products of a level go out
 * through the batched entry points, a Relinearize read only by a Rescale is evaluated with it,
 * a Mul read only by such a Relinearize joins them (evah_multiply_relinearize_rescale_many),
 * multiply_plain / add chains without other readers become one evah_weighted_sum, and a Rotate whose
 * readers are all such products (the taps of a convolution window) is not evaluated on its own: the
 * sums that end at one level and share their rotations go out as one evah_rotate_weighted_sums.
 * r5: what is left of the elementwise ops (Negate / Add / Sub / Mul on intermediates: freeable, read by somebody,
 * every check of their entry point passing) is recorded, not run; when a rescale, key switch, rotation or output needs
 * one of them, every unevaluated op below it goes out as one evah_elementwise_program per (level, batch size), storing
 * only what an op outside the program reads (EVAH_EW_FUSE=0: one launch per op). */
enum { EVAH_VAL_NONE = 0, EVAH_VAL_CT = 1, EVAH_VAL_PT = 2 };
enum { EVAH_OPF_FREE_SRC0 = 1, EVAH_OPF_FREE_SRC1 = 2 };
typedef struct evah_val { uint32_t kind; void *h; } evah_val;
typedef struct evah_op { uint32_t op, dst, src0, src1; int32_t imm; uint32_t flags; } evah_op;
int evah_execute(evah_ctx *ctx, const evah_op *ops, uint32_t n_ops, evah_val *table, uint32_t n_vals);

/* ---- a straight-line program of elementwise evaluator calls in ONE launch (eva_amd/csrc/ewprogram.hip) ----------
 * Replaces a run of SEALExecutor's elementwise dispatches — add / sub (+ plain) seal_executor.h:114-150, multiply /
 * square / multiply_plain :152-175, negate :191-195 — on values nobody else reads: Harris' response
 * det - k trace^2 is seven such calls, Sobel's magnitude and polynomial a dozen (examples/image_processing.py:39-100).
 * Values 0 .. n_in-1 are the inputs (ciphertexts or plaintexts, as in evah_execute's table), value n_in + j is the
 * result of ops[j]; `op` is the Op code of /root/reference/eva/ir/ops.h:11-25 (10 Negate, 11 Add, 12 Sub, 13 Mul) with
 * SEALExecutor's dispatch rules: a plaintext first operand of Add / Mul goes behind the ciphertext, Mul(a, a) is
 * square, Sub keeps its order (plain - cipher is "Unsupported operation encountered", as in the reference).  Checks,
 * error messages and results are those of the separate entry points called in program order; intermediates that are not
 * listed in out_vals never reach HBM.  All ciphertexts of a program are of one level and batch size.  A program beyond
 * one launch's registers runs as the separate calls.  evah_execute builds these programs itself (EVAH_EW_FUSE). */
typedef struct evah_ew_op { uint32_t op, a, b; } evah_ew_op;
int evah_elementwise_program(evah_ctx *ctx, const evah_val *in, uint32_t n_in, const evah_ew_op *ops, uint32_t n_ops,
                             const uint32_t *out_vals, uint32_t n_out, evah_ct **outs);

/* ---- limb-sharded execution (BASELINE config 5) --------------------------
 * The RNS limbs of every value are dealt over G shards — limb i on shard i mod G — one shard per
 * GPU (one process per GPU, RCCL between them) or several shards in one process.  A shard is an
 * evah_ctx with a limb -> prime map: after evah_ctx_set_shard(ctx, s, G) the context's values hold
 * only the limbs s, s+G, ... (`limbs` of a handle is then the LOCAL count) and every per-limb entry
 * point above works on them unchanged; evah_mod_switch is called on the owner of the dropped limb
 * only.  The reference has no counterpart (its parallelism is node-level,
 * multicore_program_traversal.h:55-78): these entry points split the SEAL calls that mix limbs —
 * switch_key_inplace behind relinearize / rotate_vector (seal_executor.h:200, :181/:188) and
 * rescale_to_next (:213) — into phases with ONE exchange step between phases, done by the caller:
 *   key switch   ks_digits -> all-gather of the coefficient-form digits -> ks_products ->
 *                broadcast of r from the owner of the special limb (shard l mod G) -> ks_finish
 *   rescale      rescale_last on the owner of limb l-1 -> broadcast of r -> rescale_finish
 * `l` is the GLOBAL limb count of the level.  evah_buf is device memory for the exchange steps;
 * evah_buf_ptr exposes the device address (for torch / RCCL), evah_buf_copy is the in-process exchange. */
typedef struct evah_buf evah_buf;
int evah_ctx_set_shard(evah_ctx *ctx, uint32_t shard, uint32_t n_shards);
int evah_ctx_shard_info(evah_ctx *ctx, uint32_t *shard, uint32_t *n_shards);
int evah_buf_alloc(evah_ctx *ctx, size_t words, evah_buf **out);
void evah_buf_free(evah_ctx *ctx, evah_buf *buf);
void *evah_buf_ptr(evah_buf *buf);
size_t evah_buf_words(const evah_buf *buf);
int evah_buf_copy(evah_ctx *dst_ctx, evah_buf *dst, size_t dst_off, const evah_buf *src, size_t src_off, size_t words);
/* one launch on dst's queue that pulls n (<= 64) chunks of `words` words: srcs[j][src_offs[j]..) -> dst[dst_offs[j]..)
 * (offsets and words even); sources on other devices are read as peers — the all-gather / broadcast of the phases above
 * as ONE kernel per receiving shard.  Stands where the reference's workers share memory
 * (multicore_program_traversal.h:55-78). */
int evah_buf_gather(evah_ctx *dst_ctx, evah_buf *dst, uint32_t n, const evah_buf *const *srcs, const size_t *src_offs,
                    const size_t *dst_offs, size_t words);
/* peer access between the devices of two contexts, both directions (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess);
 * a refusal is an error (evah_last_error names the pair), never a silent staging through the host.  Every multi-device
 * group calls it for each pair of its members before the first cross-device copy. */
int evah_ctx_enable_peer(evah_ctx *a, evah_ctx *b);
int evah_buf_download(evah_ctx *ctx, const evah_buf *buf, size_t off, size_t words, uint64_t *host);
int evah_buf_upload(evah_ctx *ctx, evah_buf *buf, size_t off, size_t words, const uint64_t *host);
/* (perm(c0), perm(c1)) on the local limbs: the NTT-domain Galois automorphism of rotate_vector */
int evah_shard_galois_perm(evah_ctx *ctx, const evah_ct *a, uint32_t galois_elt, evah_ct **out);
/* digits: [G][rows][N] words, rows >= ceil(l / G); row [s][j] = INTT of local limb j of polynomial `poly` */
int evah_shard_ks_digits(evah_ctx *ctx, const evah_ct *a, uint32_t poly, uint32_t l, evah_buf *digits, uint32_t rows);
/* prod: [2][ni][N] with ni = local data limbs (+1 on the owner of the special limb); r: [2][N] */
int evah_shard_ks_products(evah_ctx *ctx, const evah_ct *a, uint32_t poly, uint32_t l, const evah_buf *digits, uint32_t rows,
                           int key_kind, uint32_t galois_elt, evah_buf *prod, evah_buf *r);
/* out = (add's first add_polys polynomials, or nothing) + mod-down of prod, size 2, at `scale` */
int evah_shard_ks_finish(evah_ctx *ctx, uint32_t l, const evah_buf *prod, const evah_buf *r, const evah_ct *add,
                         uint32_t add_polys, double scale, evah_ct **out);
/* r: [size][N] */
int evah_shard_rescale_last(evah_ctx *ctx, const evah_ct *a, uint32_t l, evah_buf *r);
int evah_shard_rescale_finish(evah_ctx *ctx, const evah_ct *a, uint32_t l, const evah_buf *r, uint32_t divisor_bits,
                              evah_ct **out);

/* ---- whole-DAG capture ------------------------------------------------------------------------
 * Replaces the per-call DAG walk of SEALPublic::execute (seal.cpp:104-122) for repeated
 * executions of one compiled program: every evaluator call issued on `q0` and `others` (forks of
 * one context) between begin and end is recorded into a hipGraph, including the cross-queue
 * ordering; evah_graph_launch replays it on q0's stream.  Handles created before the capture
 * (inputs, pre-encoded plaintexts) and handles still alive at the end (outputs) keep their device
 * addresses; calls that synchronise with the host (uploads, downloads) are rejected while
 * capturing.  The temporaries of the captured walk are taken out of the queues' pools for the graph's
 * lifetime (every replay writes them again), so the queues stay usable for other work — e.g. copies of
 * the outputs enqueued on q0 right behind a replay — and evah_graph_free returns the blocks. */
int evah_capture_begin(evah_ctx *q0, evah_ctx **others, uint32_t n_others);
int evah_capture_end(evah_ctx *q0, evah_ctx **others, uint32_t n_others, evah_graph **out);
int evah_graph_launch(evah_ctx *q0, evah_graph *g);
void evah_graph_free(evah_graph *g);

/* ---- test / measurement hooks -------------------------------------------------------------- */
/* in-place negacyclic NTT (inverse=0) or INTT (inverse=1) of one host polynomial mod primes[i] */
int evah_test_ntt(evah_ctx *ctx, uint32_t prime_idx, int inverse, uint64_t *host_inout);
/* Per-launch HIP-event profile by kernel class (events are recorded on the launch stream around
 * every kernel launch while enabled); used by bench.py for the roofline of the dominant kernel. */
int evah_profile_enable(evah_ctx *ctx, int on);
int evah_profile_reset(evah_ctx *ctx);
int evah_profile_classes(void);
const char *evah_profile_class_name(int cls);
int evah_profile_get(evah_ctx *ctx, int cls, uint64_t *launches, double *total_ms);
/* HIP-event timing on the context's stream: start, stop -> elapsed milliseconds */
int evah_timer_start(evah_ctx *ctx);
int evah_timer_stop(evah_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif
