/* Plain-C use of libeva_hip.so: the op-triple (multiply -> relinearize -> rescale_to_next) on random
 * residues, once through the per-op entry points and once through the one-call submit
 * evah_execute; both must give the same words.  Nothing here but <stdint.h> types and the C-ABI
 * of include/eva_hip.h — what a cgo / JNI / ctypes / C++ caller binds.
 *
 *   gcc -O2 -Iinclude examples/c_abi_triple.c -Leva_amd/lib -leva_hip -Wl,-rpath,$PWD/eva_amd/lib -o c_abi_triple
 *   ./c_abi_triple            (needs an MI355X; prints a checksum and "match")
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "eva_hip.h"

#define N 8192u
/* CoeffModulus::Create(8192, {60, 60, 60, 60}) — SURVEY.md Appendix B */
static const uint64_t PRIMES[4] = {0xFFFFFFFFFFC4001ull, 0xFFFFFFFFFFD8001ull, 0xFFFFFFFFFFE8001ull, 0xFFFFFFFFFFFC001ull};
#define K 4u
#define L (K - 1u)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64(void) { /* splitmix64 */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint64_t *random_residues(size_t polys, uint32_t limbs) {
  uint64_t *p = malloc(sizeof(uint64_t) * polys * limbs * N);
  for (size_t a = 0; a < polys; a++)
    for (uint32_t i = 0; i < limbs; i++)
      for (uint32_t n = 0; n < N; n++) p[(a * limbs + i) * N + n] = next_u64() % PRIMES[i];
  return p;
}
#define CHECK(call)                                                            \
  do {                                                                         \
    if (call) {                                                                \
      fprintf(stderr, "%s failed: %s\n", #call, evah_last_error());            \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(void) {
  evah_ctx *ctx = NULL;
  CHECK(evah_ctx_create(N, K, PRIMES, 0, &ctx));
  uint64_t *key = random_residues((size_t)L * 2, K); /* [digit][2][k][N] */
  CHECK(evah_key_upload(ctx, EVAH_KEY_RELIN, 0, L, key));
  uint64_t *ha = random_residues(2, L), *hb = random_residues(2, L);
  evah_ct *a = NULL, *b = NULL;
  CHECK(evah_ct_upload(ctx, 2, L, 1099511627776.0 /* 2^40 */, ha, &a));
  CHECK(evah_ct_upload(ctx, 2, L, 1099511627776.0, hb, &b));

  /* per-op entry points (seal_executor.h:164, :200, :213) */
  evah_ct *m = NULL, *r = NULL, *o1 = NULL;
  CHECK(evah_multiply(ctx, a, b, &m));
  CHECK(evah_relinearize(ctx, m, &r));
  CHECK(evah_rescale(ctx, r, 60, &o1));

  /* the same as one submitted op list: slots 0,1 inputs; 2 = Mul; 3 = Relinearize; 4 = Rescale */
  evah_val table[5] = {{EVAH_VAL_CT, a}, {EVAH_VAL_CT, b}, {EVAH_VAL_NONE, NULL}, {EVAH_VAL_NONE, NULL}, {EVAH_VAL_NONE, NULL}};
  const evah_op ops[3] = {
      {13, 2, 0, 1, 0, 0},
      {20, 3, 2, 0, 0, EVAH_OPF_FREE_SRC0},
      {22, 4, 3, 0, 60, EVAH_OPF_FREE_SRC0},
  };
  CHECK(evah_execute(ctx, ops, 3, table, 5));
  evah_ct *o2 = (evah_ct *)table[4].h;

  uint32_t size = 0, limbs = 0;
  double scale = 0;
  CHECK(evah_ct_info(o2, &size, &limbs, &scale));
  const size_t words = (size_t)size * limbs * N;
  uint64_t *w1 = malloc(sizeof(uint64_t) * words), *w2 = malloc(sizeof(uint64_t) * words);
  CHECK(evah_ct_download(ctx, o1, w1));
  CHECK(evah_ct_download(ctx, o2, w2));
  uint64_t sum = 0;
  int same = 1;
  for (size_t i = 0; i < words; i++) {
    sum = sum * 1099511628211ull + w1[i];
    same &= (w1[i] == w2[i]);
  }
  printf("op-triple: size %u, limbs %u, scale 2^%.0f, checksum %016llx, evah_execute %s\n", size, limbs,
         __builtin_log2(scale), (unsigned long long)sum, same ? "match" : "MISMATCH");
  evah_ct_free(ctx, o2);
  evah_ct_free(ctx, o1);
  evah_ct_free(ctx, r);
  evah_ct_free(ctx, m);
  evah_ct_free(ctx, b);
  evah_ct_free(ctx, a);
  evah_ctx_destroy(ctx);
  free(w1); free(w2); free(ha); free(hb); free(key);
  return same ? 0 : 2;
}
