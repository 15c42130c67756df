/*
 * eva_oracle_dag.c — CPU walk of a compiled program's encrypted part over the oracle's evaluator:
 * the reported CPU baseline of the DAG configurations.  TEST INFRASTRUCTURE ONLY (see
 * eva_oracle.h); nothing under eva_amd/ links or loads it.
 *
 * Restates, over a flat op list, the two traversals the reference offers:
 *   threads == 1 : ProgramTraversal::forwardPass (/root/reference/eva/common/program_traversal.h:36-93)
 *                  — the list in topological order, one node after the other;
 *   threads  > 1 : MulticoreProgramTraversal::forwardPass
 *                  (/root/reference/eva/common/multicore_program_traversal.h:24-83) — dependency
 *                  counting: a node becomes ready when its last operand has been evaluated, ready
 *                  nodes are taken by a pool of worker threads (Galois do_all over a worklist
 *                  there, pthreads here), operands are freed at their last use (:62-67).
 * Node dispatch follows SEALExecutor::operator() (/root/reference/eva/seal/seal_executor.h:279-404):
 * Add / Mul put the ciphertext first, Mul of a term with itself is square, right rotation is a
 * negative step count (:188).
 */
#include "eva_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;

/* op codes: /root/reference/eva/ir/ops.h:11-25 */
enum { OP_INPUT = 1, OP_OUTPUT = 2, OP_CONSTANT = 3, OP_NEGATE = 10, OP_ADD = 11, OP_SUB = 12, OP_MUL = 13,
       OP_ROTL = 14, OP_ROTR = 15, OP_RELIN = 20, OP_MODSWITCH = 21, OP_RESCALE = 22, OP_ENCODE = 23 };

typedef struct {
  const evo_ctx *c;
  const evo_dag_op *ops;
  uint32_t n_ops, n_vals;
  evo_dag_val *vals;
  const u64 *relin;
  const uint32_t *gal_elts;
  const u64 *const *gal_keys;
  uint32_t n_gal;
  /* scheduling state */
  int *waiting;        /* per op: operands not yet produced */
  uint32_t *readers;   /* per value: reads left (eager free) */
  char *keep;          /* per value: caller-placed or an output: never freed here */
  int *producer;       /* per value: op index or -1 */
  uint32_t **users;    /* per op: ops that read its dst */
  uint32_t *n_users;
  uint32_t *ready;
  uint32_t ready_head, ready_tail, done, failed;
  pthread_mutex_t mu;
  pthread_cond_t cv;
} walk_t;

static int arity(uint32_t op) {
  if (op == OP_ADD || op == OP_SUB || op == OP_MUL) return 2;
  if (op == OP_INPUT || op == OP_CONSTANT || op == OP_ENCODE) return 0;
  return 1;
}

static const u64 *galois_key(const walk_t *w, int32_t steps) {
  uint32_t elt = evo_galois_elt_from_step(evo_ctx_n(w->c), steps);
  for (uint32_t i = 0; i < w->n_gal; i++)
    if (w->gal_elts[i] == elt) return w->gal_keys[i];
  return NULL;
}

/* one node; returns 0 on success */
static int eval_node(walk_t *w, const evo_dag_op *o) {
  const evo_ctx *c = w->c;
  const size_t N = evo_ctx_n(c);
  evo_dag_val *out = &w->vals[o->dst];
  const evo_dag_val *a = &w->vals[o->src0], *b = arity(o->op) == 2 ? &w->vals[o->src1] : NULL;
  if (arity(o->op) == 2 && a->kind != 1) { /* ciphertext first (seal_executor.h:116-119,155-158) */
    if (o->op == OP_SUB || b->kind != 1) return -1;
    const evo_dag_val *t = a; a = b; b = t;
  }
  if (a->kind != 1 && o->op != OP_OUTPUT) return -1;
  uint32_t size = a->size, limbs = a->limbs;
  switch (o->op) {
  case OP_ADD: case OP_SUB:
    if (b->kind == 1 && b->size > size) size = b->size;
    break;
  case OP_MUL:
    if (b->kind == 1) size = 3;
    break;
  case OP_RELIN: size = 2; break;
  case OP_MODSWITCH: case OP_RESCALE: limbs = a->limbs - 1; break;
  default: break;
  }
  if (o->op == OP_OUTPUT) { /* the same value under another name: shares the data, never freed here */
    *out = *a;
    out->alias = 1;
    return 0;
  }
  u64 *d = (u64 *)malloc(sizeof(u64) * size * limbs * N);
  if (!d) return -2;
  switch (o->op) {
  case OP_NEGATE: evo_negate(c, a->limbs, a->data, a->size, d); break;
  case OP_ADD:
    if (b->kind == 1) evo_add(c, a->limbs, a->data, a->size, b->data, b->size, d);
    else evo_add_plain(c, a->limbs, a->data, a->size, b->data, d);
    break;
  case OP_SUB:
    if (b->kind == 1) evo_sub(c, a->limbs, a->data, a->size, b->data, b->size, d);
    else evo_sub_plain(c, a->limbs, a->data, a->size, b->data, d);
    break;
  case OP_MUL:
    if (b->kind == 1) {
      if (a->size != 2 || b->size != 2) { free(d); return -3; }
      if (o->src0 == o->src1) evo_square(c, a->limbs, a->data, d);
      else evo_multiply(c, a->limbs, a->data, b->data, d);
    } else {
      evo_multiply_plain(c, a->limbs, a->data, a->size, b->data, d);
    }
    break;
  case OP_ROTL: case OP_ROTR: {
    int32_t steps = o->op == OP_ROTL ? o->imm : -o->imm;
    const u64 *key = steps ? galois_key(w, steps) : NULL;
    if (steps && !key) { free(d); return -4; }
    evo_rotate(c, a->limbs, a->data, steps, key, d);
    break;
  }
  case OP_RELIN: evo_relinearize(c, a->limbs, a->data, w->relin, d); break;
  case OP_MODSWITCH: evo_mod_switch(c, a->limbs, a->data, a->size, d); break;
  case OP_RESCALE: evo_rescale(c, a->limbs, a->data, a->size, d); break;
  default: free(d); return -5;
  }
  out->kind = 1;
  out->size = size;
  out->limbs = limbs;
  out->data = d;
  out->alias = 0;
  return 0;
}

static void release_operands(walk_t *w, const evo_dag_op *o) { /* caller holds w->mu */
  const uint32_t srcs[2] = {o->src0, o->src1};
  for (int k = 0; k < arity(o->op); k++) {
    uint32_t v = srcs[k];
    if (k == 1 && o->src0 == o->src1) { w->readers[v]--; continue; }
    if (--w->readers[v] == 0 && !w->keep[v] && w->vals[v].kind == 1 && !w->vals[v].alias) {
      free(w->vals[v].data);
      w->vals[v].data = NULL;
      w->vals[v].kind = 0;
    }
  }
}

static void *worker(void *arg) {
  walk_t *w = (walk_t *)arg;
  for (;;) {
    pthread_mutex_lock(&w->mu);
    while (w->ready_head == w->ready_tail && w->done < w->n_ops && !w->failed) pthread_cond_wait(&w->cv, &w->mu);
    if (w->done >= w->n_ops || w->failed) {
      pthread_mutex_unlock(&w->mu);
      return NULL;
    }
    uint32_t i = w->ready[w->ready_head++];
    pthread_mutex_unlock(&w->mu);
    int rc = arity(w->ops[i].op) ? eval_node(w, &w->ops[i]) : 0;
    pthread_mutex_lock(&w->mu);
    if (rc) w->failed = (uint32_t)(-rc);
    if (arity(w->ops[i].op)) release_operands(w, &w->ops[i]);
    for (uint32_t u = 0; u < w->n_users[i]; u++)
      if (--w->waiting[w->users[i][u]] == 0) w->ready[w->ready_tail++] = w->users[i][u];
    w->done++;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
  }
}

int evo_dag_walk(const evo_ctx *c, const evo_dag_op *ops, uint32_t n_ops, evo_dag_val *vals, uint32_t n_vals,
                 const uint64_t *relin_key, const uint32_t *galois_elts, const uint64_t *const *galois_keys,
                 uint32_t n_galois, int threads) {
  walk_t w;
  memset(&w, 0, sizeof w);
  w.c = c; w.ops = ops; w.n_ops = n_ops; w.vals = vals; w.n_vals = n_vals;
  w.relin = relin_key; w.gal_elts = galois_elts; w.gal_keys = galois_keys; w.n_gal = n_galois;
  w.waiting = (int *)calloc(n_ops, sizeof(int));
  w.readers = (uint32_t *)calloc(n_vals, sizeof(uint32_t));
  w.keep = (char *)calloc(n_vals, 1);
  w.producer = (int *)malloc(sizeof(int) * n_vals);
  w.users = (uint32_t **)calloc(n_ops, sizeof(uint32_t *));
  w.n_users = (uint32_t *)calloc(n_ops, sizeof(uint32_t));
  w.ready = (uint32_t *)malloc(sizeof(uint32_t) * (n_ops + 1));
  for (uint32_t v = 0; v < n_vals; v++) { w.producer[v] = -1; w.keep[v] = vals[v].kind != 0; }
  for (uint32_t i = 0; i < n_ops; i++) {
    if (ops[i].dst >= n_vals) return -10;
    if (arity(ops[i].op)) w.producer[ops[i].dst] = (int)i;
    if (ops[i].op == OP_OUTPUT) { w.keep[ops[i].dst] = 1; w.keep[ops[i].src0] = 1; }
  }
  /* users / waiting: count first, then fill */
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t i = 0; i < n_ops; i++) {
      const uint32_t srcs[2] = {ops[i].src0, ops[i].src1};
      for (int k = 0; k < arity(ops[i].op); k++) {
        uint32_t v = srcs[k];
        if (v >= n_vals) return -10;
        if (pass == 0) w.readers[v]++;
        int p = w.producer[v];
        if (p < 0) continue;
        if (k == 1 && ops[i].src0 == ops[i].src1) continue;
        if (pass == 0) { w.n_users[p]++; w.waiting[i]++; }
        else w.users[p][w.n_users[p]++] = i;
      }
    }
    if (pass == 0)
      for (uint32_t i = 0; i < n_ops; i++) {
        w.users[i] = (uint32_t *)malloc(sizeof(uint32_t) * (w.n_users[i] + 1));
        w.n_users[i] = 0;
      }
  }
  for (uint32_t i = 0; i < n_ops; i++)
    if (w.waiting[i] == 0) w.ready[w.ready_tail++] = i;
  pthread_mutex_init(&w.mu, NULL);
  pthread_cond_init(&w.cv, NULL);
  if (threads <= 1) {
    /* serial forwardPass: list order (topological), same frees */
    for (uint32_t i = 0; i < n_ops && !w.failed; i++) {
      if (!arity(ops[i].op)) continue;
      int rc = eval_node(&w, &ops[i]);
      if (rc) w.failed = (uint32_t)(-rc);
      release_operands(&w, &ops[i]);
    }
  } else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &w);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
  }
  pthread_mutex_destroy(&w.mu);
  pthread_cond_destroy(&w.cv);
  for (uint32_t i = 0; i < n_ops; i++) free(w.users[i]);
  free(w.users); free(w.n_users); free(w.waiting); free(w.readers); free(w.keep); free(w.producer); free(w.ready);
  return w.failed ? -(int)w.failed : 0;
}

void evo_dag_free(uint64_t *data) { free(data); }
