/*
 * infera_oracle.c -- CPU ORACLE (test infrastructure, never linked into the product).
 * See infera_oracle.h for scope and parity-pin status.
 *
 * Layout of this file:
 *   1. protobuf wire reader + ONNX ModelProto subset (what tract-onnx's model_for_path parses;
 *      reference call site engine.rs:49-51)
 *   2. fp32 operator executor following the ONNX operator spec (what SimplePlan::run computes;
 *      reference call site engine.rs:142-145, 246-249)
 *   3. engine mirror: shape_rows_cols / run_inference_impl / run_inference_blob_impl
 *      (engine.rs:19-29, 111-164, 199-263) with error.rs:13-61 message texts
 *   4. synthetic table generator + multi-threaded CPU scan baseline
 *
 * Arithmetic convention (stated so the HIP kernels can be compared meaningfully): every
 * dot product is a k-ordered chain of fused multiply-adds in f32 starting from 0,
 * acc = fmaf(a[k], b[k], acc), one rounding per step; bias / residual adds are separate
 * f32 additions.  Summation order is unspecified by ONNX and by the reference; the
 * north-star tolerance (1e-4 relative) absorbs order differences.
 */
#define _GNU_SOURCE
#include "infera_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* 0. small utilities                                                                          */
/* ------------------------------------------------------------------------------------------ */

static void set_err(char *err, size_t errlen, const char *fmt, ...) {
  if (!err || errlen == 0) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err, errlen, fmt, ap);
  va_end(ap);
}

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}
static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz ? sz : 1);
  if (!p) {
    fprintf(stderr, "oracle: out of memory\n");
    abort();
  }
  return p;
}
static char *xstrndup(const uint8_t *s, size_t n) {
  char *p = (char *)xmalloc(n + 1);
  memcpy(p, s, n);
  p[n] = 0;
  return p;
}

/* ------------------------------------------------------------------------------------------ */
/* 1. protobuf wire reader + ONNX subset                                                       */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  const uint8_t *p, *end;
  int bad;
} Pb;

static uint64_t pb_varint(Pb *b) {
  uint64_t v = 0;
  int shift = 0;
  while (b->p < b->end && shift < 70) {
    uint8_t c = *b->p++;
    v |= (uint64_t)(c & 0x7f) << shift;
    if (!(c & 0x80)) return v;
    shift += 7;
  }
  b->bad = 1;
  return 0;
}
/* Reads a tag; returns 0 at end. wire type in *wt. */
static uint32_t pb_tag(Pb *b, int *wt) {
  if (b->p >= b->end || b->bad) return 0;
  uint64_t t = pb_varint(b);
  *wt = (int)(t & 7);
  if ((t >> 3) == 0) b->bad = 1;
  return (uint32_t)(t >> 3);
}
static Pb pb_sub(Pb *b) {
  Pb s = {0, 0, 0};
  uint64_t n = pb_varint(b);
  if (b->bad || n > (uint64_t)(b->end - b->p)) {
    b->bad = 1;
    s.bad = 1;
    return s;
  }
  s.p = b->p;
  s.end = b->p + n;
  b->p += n;
  return s;
}
static void pb_skip(Pb *b, int wt) {
  switch (wt) {
    case 0: (void)pb_varint(b); break;
    case 1:
      if (b->end - b->p < 8) b->bad = 1; else b->p += 8;
      break;
    case 2: (void)pb_sub(b); break;
    case 5:
      if (b->end - b->p < 4) b->bad = 1; else b->p += 4;
      break;
    default: b->bad = 1;
  }
}

enum { DT_FLOAT = 1, DT_INT32 = 6, DT_INT64 = 7, DT_DOUBLE = 11 };

#define MAXRANK 8

typedef struct {
  char *name;
  int dtype; /* DT_FLOAT or DT_INT64 */
  int rank;
  int64_t dims[MAXRANK];
  size_t n;
  float *f;     /* dtype FLOAT */
  int64_t *i64; /* dtype INT64 */
  int arena;    /* payload lives in the thread's bump arena (not freed individually) */
} Tensor;

typedef struct {
  char *name;
  int type; /* 1 f, 2 i, 3 s, 4 t, 6 floats, 7 ints */
  float f;
  int64_t i;
  char *s;
  int64_t *ints;
  size_t nints;
  float *floats;
  size_t nfloats;
  Tensor *t;
} Attr;

typedef struct {
  char *op;
  char *name;
  char **in;
  size_t nin;
  char **out;
  size_t nout;
  Attr *attrs;
  size_t nattrs;
} Node;

typedef struct {
  char *name;
  int has_type;
  int elem_type;
  int rank; /* -1 unknown */
  int64_t dims[MAXRANK]; /* -1 symbolic / unknown */
} ValueInfo;

struct OrcModel {
  int64_t ir_version;
  int64_t opset; /* default-domain opset version */
  Node *nodes;
  size_t nnodes;
  Tensor *inits;
  size_t ninits;
  ValueInfo *inputs; /* graph inputs that are NOT initializers */
  size_t ninputs;
  ValueInfo *outputs;
  size_t noutputs;
  int in_rank, out_rank;
  int64_t in_shape[MAXRANK], out_shape[MAXRANK];
};

/* Optional per-thread bump arena for intermediate tensors.  The multi-threaded scan baseline turns
 * it on so that hundreds of threads do not serialise on mmap/munmap of MB-sized per-chunk tensors
 * (an allocator artefact, not part of the algorithm); everything else uses plain malloc. */
static __thread struct { char *base; size_t cap, off; int active; } t_arena;

static void *arena_alloc(size_t n) {
  n = (n + 63) & ~(size_t)63;
  if (t_arena.off + n > t_arena.cap) return NULL;
  void *p = t_arena.base + t_arena.off;
  t_arena.off += n;
  return p;
}

static void tensor_free_payload(Tensor *t) {
  free(t->name);
  if (!t->arena) {
    free(t->f);
    free(t->i64);
  }
}

static size_t dims_count(const int64_t *d, int rank) {
  size_t n = 1;
  for (int i = 0; i < rank; i++) n *= (size_t)(d[i] < 0 ? 0 : d[i]);
  return n;
}

/* TensorProto: dims=1, data_type=2, float_data=4, int32_data=5, int64_data=7, name=8,
 * raw_data=9, double_data=10. */
static int parse_tensor(Pb b, Tensor *t, char *err, size_t errlen) {
  memset(t, 0, sizeof *t);
  const uint8_t *raw = NULL;
  size_t rawlen = 0;
  float *fd = NULL;
  size_t nfd = 0, capfd = 0;
  int64_t *id = NULL;
  size_t nid = 0, capid = 0;
  int dtype = 0, wt;
  uint32_t f;
  while ((f = pb_tag(&b, &wt))) {
    if (f == 1) { /* dims: packed or not */
      if (wt == 2) {
        Pb s = pb_sub(&b);
        while (s.p < s.end && !s.bad) {
          if (t->rank >= MAXRANK) { set_err(err, errlen, "tensor rank > %d", MAXRANK); goto fail; }
          t->dims[t->rank++] = (int64_t)pb_varint(&s);
        }
      } else {
        if (t->rank >= MAXRANK) { set_err(err, errlen, "tensor rank > %d", MAXRANK); goto fail; }
        t->dims[t->rank++] = (int64_t)pb_varint(&b);
      }
    } else if (f == 2 && wt == 0) {
      dtype = (int)pb_varint(&b);
    } else if (f == 4) { /* float_data */
      if (wt == 2) {
        Pb s = pb_sub(&b);
        size_t k = (size_t)(s.end - s.p) / 4;
        if (nfd + k > capfd) { capfd = (nfd + k) * 2; fd = realloc(fd, capfd * 4); }
        memcpy(fd + nfd, s.p, k * 4);
        nfd += k;
      } else if (wt == 5) {
        if (nfd + 1 > capfd) { capfd = (nfd + 1) * 2; fd = realloc(fd, capfd * 4); }
        if (b.end - b.p < 4) { b.bad = 1; break; }
        memcpy(fd + nfd, b.p, 4);
        b.p += 4;
        nfd++;
      } else pb_skip(&b, wt);
    } else if (f == 7 || f == 5) { /* int64_data / int32_data (varints) */
      if (wt == 2) {
        Pb s = pb_sub(&b);
        while (s.p < s.end && !s.bad) {
          if (nid + 1 > capid) { capid = (nid + 1) * 2; id = realloc(id, capid * 8); }
          id[nid++] = (int64_t)pb_varint(&s);
        }
      } else {
        if (nid + 1 > capid) { capid = (nid + 1) * 2; id = realloc(id, capid * 8); }
        id[nid++] = (int64_t)pb_varint(&b);
      }
    } else if (f == 8 && wt == 2) {
      Pb s = pb_sub(&b);
      free(t->name);
      t->name = xstrndup(s.p, (size_t)(s.end - s.p));
    } else if (f == 9 && wt == 2) {
      Pb s = pb_sub(&b);
      raw = s.p;
      rawlen = (size_t)(s.end - s.p);
    } else {
      pb_skip(&b, wt);
    }
  }
  if (b.bad) { set_err(err, errlen, "malformed TensorProto"); goto fail; }
  t->n = dims_count(t->dims, t->rank);
  if (dtype == DT_FLOAT) {
    t->dtype = DT_FLOAT;
    t->f = (float *)xmalloc(t->n * 4);
    if (raw) {
      if (rawlen != t->n * 4) { set_err(err, errlen, "tensor '%s': raw_data size mismatch", t->name ? t->name : ""); goto fail; }
      memcpy(t->f, raw, rawlen);
    } else {
      if (nfd != t->n) { set_err(err, errlen, "tensor '%s': float_data size mismatch", t->name ? t->name : ""); goto fail; }
      memcpy(t->f, fd, nfd * 4);
    }
  } else if (dtype == DT_INT64 || dtype == DT_INT32) {
    t->dtype = DT_INT64;
    t->i64 = (int64_t *)xmalloc(t->n * 8);
    if (raw) {
      size_t es = dtype == DT_INT64 ? 8 : 4;
      if (rawlen != t->n * es) { set_err(err, errlen, "tensor '%s': raw_data size mismatch", t->name ? t->name : ""); goto fail; }
      for (size_t i = 0; i < t->n; i++) {
        if (es == 8) { int64_t v; memcpy(&v, raw + i * 8, 8); t->i64[i] = v; }
        else { int32_t v; memcpy(&v, raw + i * 4, 4); t->i64[i] = v; }
      }
    } else {
      if (nid != t->n) { set_err(err, errlen, "tensor '%s': int data size mismatch", t->name ? t->name : ""); goto fail; }
      memcpy(t->i64, id, nid * 8);
    }
  } else {
    set_err(err, errlen, "tensor '%s': unsupported data_type %d", t->name ? t->name : "", dtype);
    goto fail;
  }
  free(fd);
  free(id);
  return 0;
fail:
  free(fd);
  free(id);
  tensor_free_payload(t);
  memset(t, 0, sizeof *t);
  return -1;
}

/* AttributeProto: name=1, f=2, i=3, s=4, t=5, floats=7, ints=8, type=20 */
static int parse_attr(Pb b, Attr *a, char *err, size_t errlen) {
  memset(a, 0, sizeof *a);
  int wt, saw_f = 0, saw_i = 0;
  uint32_t f;
  size_t capi = 0, capf = 0;
  while ((f = pb_tag(&b, &wt))) {
    if (f == 1 && wt == 2) {
      Pb s = pb_sub(&b);
      a->name = xstrndup(s.p, (size_t)(s.end - s.p));
    } else if (f == 2 && wt == 5) {
      if (b.end - b.p < 4) { b.bad = 1; break; }
      memcpy(&a->f, b.p, 4);
      b.p += 4;
      saw_f = 1;
    } else if (f == 3 && wt == 0) {
      a->i = (int64_t)pb_varint(&b);
      saw_i = 1;
    } else if (f == 4 && wt == 2) {
      Pb s = pb_sub(&b);
      a->s = xstrndup(s.p, (size_t)(s.end - s.p));
    } else if (f == 5 && wt == 2) {
      Pb s = pb_sub(&b);
      a->t = (Tensor *)xcalloc(1, sizeof(Tensor));
      if (parse_tensor(s, a->t, err, errlen)) return -1;
    } else if (f == 7) {
      if (wt == 2) {
        Pb s = pb_sub(&b);
        size_t k = (size_t)(s.end - s.p) / 4;
        if (a->nfloats + k > capf) { capf = (a->nfloats + k) * 2; a->floats = realloc(a->floats, capf * 4); }
        memcpy(a->floats + a->nfloats, s.p, k * 4);
        a->nfloats += k;
      } else if (wt == 5) {
        if (a->nfloats + 1 > capf) { capf = (a->nfloats + 1) * 2; a->floats = realloc(a->floats, capf * 4); }
        if (b.end - b.p < 4) { b.bad = 1; break; }
        memcpy(a->floats + a->nfloats++, b.p, 4);
        b.p += 4;
      } else pb_skip(&b, wt);
    } else if (f == 8) {
      if (wt == 2) {
        Pb s = pb_sub(&b);
        while (s.p < s.end && !s.bad) {
          if (a->nints + 1 > capi) { capi = (a->nints + 1) * 2; a->ints = realloc(a->ints, capi * 8); }
          a->ints[a->nints++] = (int64_t)pb_varint(&s);
        }
      } else {
        if (a->nints + 1 > capi) { capi = (a->nints + 1) * 2; a->ints = realloc(a->ints, capi * 8); }
        a->ints[a->nints++] = (int64_t)pb_varint(&b);
      }
    } else if (f == 20 && wt == 0) {
      a->type = (int)pb_varint(&b);
    } else {
      pb_skip(&b, wt);
    }
  }
  if (b.bad) { set_err(err, errlen, "malformed AttributeProto"); return -1; }
  if (a->type == 0) { /* older writers omit `type` */
    if (a->nints) a->type = 7;
    else if (a->nfloats) a->type = 6;
    else if (a->t) a->type = 4;
    else if (a->s) a->type = 3;
    else if (saw_f) a->type = 1;
    else if (saw_i) a->type = 2;
  }
  return 0;
}

static void push_str(char ***arr, size_t *n, char *s) {
  *arr = realloc(*arr, (*n + 1) * sizeof(char *));
  (*arr)[(*n)++] = s;
}

/* NodeProto: input=1, output=2, name=3, op_type=4, attribute=5 */
static int parse_node(Pb b, Node *nd, char *err, size_t errlen) {
  memset(nd, 0, sizeof *nd);
  int wt;
  uint32_t f;
  while ((f = pb_tag(&b, &wt))) {
    if (wt == 2 && (f == 1 || f == 2 || f == 3 || f == 4)) {
      Pb s = pb_sub(&b);
      char *str = xstrndup(s.p, (size_t)(s.end - s.p));
      if (f == 1) push_str(&nd->in, &nd->nin, str);
      else if (f == 2) push_str(&nd->out, &nd->nout, str);
      else if (f == 3) { free(nd->name); nd->name = str; }
      else { free(nd->op); nd->op = str; }
    } else if (f == 5 && wt == 2) {
      Pb s = pb_sub(&b);
      nd->attrs = realloc(nd->attrs, (nd->nattrs + 1) * sizeof(Attr));
      if (parse_attr(s, &nd->attrs[nd->nattrs], err, errlen)) return -1;
      nd->nattrs++;
    } else {
      pb_skip(&b, wt);
    }
  }
  if (b.bad || !nd->op) { set_err(err, errlen, "malformed NodeProto"); return -1; }
  return 0;
}

/* ValueInfoProto: name=1, type=2{tensor_type=1{elem_type=1, shape=2{dim=1{dim_value=1,dim_param=2}}}} */
static int parse_value_info(Pb b, ValueInfo *vi) {
  memset(vi, 0, sizeof *vi);
  vi->rank = -1;
  int wt;
  uint32_t f;
  while ((f = pb_tag(&b, &wt))) {
    if (f == 1 && wt == 2) {
      Pb s = pb_sub(&b);
      vi->name = xstrndup(s.p, (size_t)(s.end - s.p));
    } else if (f == 2 && wt == 2) {
      Pb ty = pb_sub(&b);
      int wt2;
      uint32_t f2;
      while ((f2 = pb_tag(&ty, &wt2))) {
        if (f2 == 1 && wt2 == 2) { /* tensor_type */
          Pb tt = pb_sub(&ty);
          vi->has_type = 1;
          int wt3;
          uint32_t f3;
          while ((f3 = pb_tag(&tt, &wt3))) {
            if (f3 == 1 && wt3 == 0) vi->elem_type = (int)pb_varint(&tt);
            else if (f3 == 2 && wt3 == 2) {
              Pb sh = pb_sub(&tt);
              vi->rank = 0;
              int wt4;
              uint32_t f4;
              while ((f4 = pb_tag(&sh, &wt4))) {
                if (f4 == 1 && wt4 == 2) {
                  Pb dm = pb_sub(&sh);
                  int64_t val = -1;
                  int wt5;
                  uint32_t f5;
                  while ((f5 = pb_tag(&dm, &wt5))) {
                    if (f5 == 1 && wt5 == 0) val = (int64_t)pb_varint(&dm);
                    else pb_skip(&dm, wt5);
                  }
                  if (vi->rank < MAXRANK) vi->dims[vi->rank++] = val;
                  if (dm.bad) b.bad = 1;
                } else pb_skip(&sh, wt4);
              }
              if (sh.bad) b.bad = 1;
            } else pb_skip(&tt, wt3);
          }
          if (tt.bad) b.bad = 1;
        } else pb_skip(&ty, wt2);
      }
      if (ty.bad) b.bad = 1;
    } else {
      pb_skip(&b, wt);
    }
  }
  return (b.bad || !vi->name) ? -1 : 0;
}

static const Tensor *find_init(const OrcModel *m, const char *name) {
  for (size_t i = 0; i < m->ninits; i++)
    if (m->inits[i].name && strcmp(m->inits[i].name, name) == 0) return &m->inits[i];
  return NULL;
}

/* GraphProto: node=1, name=2, initializer=5, input=11, output=12 */
static int parse_graph(Pb b, OrcModel *m, char *err, size_t errlen) {
  int wt;
  uint32_t f;
  ValueInfo *allin = NULL;
  size_t nallin = 0;
  while ((f = pb_tag(&b, &wt))) {
    if (f == 1 && wt == 2) {
      Pb s = pb_sub(&b);
      m->nodes = realloc(m->nodes, (m->nnodes + 1) * sizeof(Node));
      if (parse_node(s, &m->nodes[m->nnodes], err, errlen)) { free(allin); return -1; }
      m->nnodes++;
    } else if (f == 5 && wt == 2) {
      Pb s = pb_sub(&b);
      m->inits = realloc(m->inits, (m->ninits + 1) * sizeof(Tensor));
      if (parse_tensor(s, &m->inits[m->ninits], err, errlen)) { free(allin); return -1; }
      m->ninits++;
    } else if ((f == 11 || f == 12) && wt == 2) {
      Pb s = pb_sub(&b);
      ValueInfo vi;
      if (parse_value_info(s, &vi)) { set_err(err, errlen, "malformed ValueInfoProto"); free(allin); return -1; }
      if (f == 11) {
        allin = realloc(allin, (nallin + 1) * sizeof(ValueInfo));
        allin[nallin++] = vi;
      } else {
        m->outputs = realloc(m->outputs, (m->noutputs + 1) * sizeof(ValueInfo));
        m->outputs[m->noutputs++] = vi;
      }
    } else {
      pb_skip(&b, wt);
    }
  }
  if (b.bad) { set_err(err, errlen, "malformed GraphProto"); free(allin); return -1; }
  /* real inputs = graph.input minus initializers (IR < 4 lists initializers as inputs too) */
  for (size_t i = 0; i < nallin; i++) {
    if (find_init(m, allin[i].name)) { free(allin[i].name); continue; }
    m->inputs = realloc(m->inputs, (m->ninputs + 1) * sizeof(ValueInfo));
    m->inputs[m->ninputs++] = allin[i];
  }
  free(allin);
  return 0;
}

static const Attr *find_attr(const Node *nd, const char *name) {
  for (size_t i = 0; i < nd->nattrs; i++)
    if (nd->attrs[i].name && strcmp(nd->attrs[i].name, name) == 0) return &nd->attrs[i];
  return NULL;
}
static int64_t attr_i(const Node *nd, const char *name, int64_t dflt) {
  const Attr *a = find_attr(nd, name);
  return a ? a->i : dflt;
}
static float attr_f(const Node *nd, const char *name, float dflt) {
  const Attr *a = find_attr(nd, name);
  return a ? a->f : dflt;
}

/* ------------------------------------------------------------------------------------------ */
/* 2. fp32 executor                                                                             */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  Tensor *v;
  size_t n, cap;
} Env;

static Tensor *env_new(Env *e, const char *name, int dtype, int rank, const int64_t *dims_in) {
  /* callers pass an input tensor's own dims, which live in e->v: copy them before the array may move */
  int64_t dims[MAXRANK];
  memcpy(dims, dims_in, sizeof(int64_t) * (size_t)rank);
  if (e->n == e->cap) {
    e->cap = e->cap ? e->cap * 2 : 64;
    e->v = realloc(e->v, e->cap * sizeof(Tensor));
  }
  Tensor *t = &e->v[e->n++];
  memset(t, 0, sizeof *t);
  t->name = strdup(name);
  t->dtype = dtype;
  t->rank = rank;
  memcpy(t->dims, dims, sizeof(int64_t) * (size_t)rank);
  t->n = dims_count(dims, rank);
  void *mem = t_arena.active ? arena_alloc(t->n * (dtype == DT_FLOAT ? 4 : 8)) : NULL;
  t->arena = mem != NULL;
  if (dtype == DT_FLOAT) t->f = mem ? (float *)mem : (float *)xmalloc(t->n * 4);
  else t->i64 = mem ? (int64_t *)mem : (int64_t *)xmalloc(t->n * 8);
  return t;
}
static void env_free(Env *e) {
  for (size_t i = 0; i < e->n; i++) tensor_free_payload(&e->v[i]);
  free(e->v);
}
/* NOTE: returned pointers into e->v are invalidated by env_new (realloc); callers look inputs up
 * AFTER creating outputs, or copy what they need first.  To keep it simple every op below fetches
 * its inputs through an index and re-resolves after env_new. */
static long env_find(const Env *e, const char *name) {
  for (size_t i = e->n; i-- > 0;)
    if (strcmp(e->v[i].name, name) == 0) return (long)i;
  return -1;
}

typedef struct {
  const OrcModel *m;
  Env env;
  char *err;
  size_t errlen;
} Exec;

static const Tensor *get_in(Exec *x, const Node *nd, size_t k) {
  if (k >= nd->nin || nd->in[k][0] == 0) return NULL;
  long i = env_find(&x->env, nd->in[k]);
  if (i >= 0) return &x->env.v[i];
  return find_init(x->m, nd->in[k]);
}

#define FAIL(...)                                  \
  do {                                             \
    set_err(x->err, x->errlen, __VA_ARGS__);       \
    return -1;                                     \
  } while (0)

/* ---- "best CPU" GEMM for bench.py's cpu_baseline.best_cpu leg ------------------------
 * A register-blocked micro-kernel: 6 rows x 32 columns (AVX-512: 12 zmm accumulators, 2 B loads and one broadcast per k) or
 * 6 x 16 (AVX2), the B panel of one column block walked top to bottom for every row block (K x 32 floats: L1-resident).  Every
 * C element is still ONE k-ordered fmaf chain from 0, so the results are bit-identical to gemm_nn below (tests/test_oracle_
 * blocked_gemm.py) -- only the instruction schedule is what a packed SIMD matmul such as Tract's runs.  Selected per thread
 * (t_gemm_blocked) by the bench scan only; every parity test runs the plain loop. */
static __thread int t_gemm_blocked;

__attribute__((target("avx512f"))) static void gemm_block_avx512(const float *A, const float *B, float *C, size_t N, size_t K, size_t M) {
  size_t j = 0;
  for (; j + 32 <= M; j += 32) {
    size_t n = 0;
    for (; n + 6 <= N; n += 6) {
      __m512 c00 = _mm512_setzero_ps(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00, c40 = c00, c41 = c00, c50 = c00, c51 = c00;
      const float *a0 = A + n * K, *b = B + j;
      for (size_t k = 0; k < K; k++, b += M) {
        const __m512 b0 = _mm512_loadu_ps(b), b1 = _mm512_loadu_ps(b + 16);
        __m512 a;
        a = _mm512_set1_ps(a0[k]);         c00 = _mm512_fmadd_ps(a, b0, c00); c01 = _mm512_fmadd_ps(a, b1, c01);
        a = _mm512_set1_ps(a0[K + k]);     c10 = _mm512_fmadd_ps(a, b0, c10); c11 = _mm512_fmadd_ps(a, b1, c11);
        a = _mm512_set1_ps(a0[2 * K + k]); c20 = _mm512_fmadd_ps(a, b0, c20); c21 = _mm512_fmadd_ps(a, b1, c21);
        a = _mm512_set1_ps(a0[3 * K + k]); c30 = _mm512_fmadd_ps(a, b0, c30); c31 = _mm512_fmadd_ps(a, b1, c31);
        a = _mm512_set1_ps(a0[4 * K + k]); c40 = _mm512_fmadd_ps(a, b0, c40); c41 = _mm512_fmadd_ps(a, b1, c41);
        a = _mm512_set1_ps(a0[5 * K + k]); c50 = _mm512_fmadd_ps(a, b0, c50); c51 = _mm512_fmadd_ps(a, b1, c51);
      }
      float *c = C + n * M + j;
      _mm512_storeu_ps(c, c00);         _mm512_storeu_ps(c + 16, c01);
      _mm512_storeu_ps(c + M, c10);     _mm512_storeu_ps(c + M + 16, c11);
      _mm512_storeu_ps(c + 2 * M, c20); _mm512_storeu_ps(c + 2 * M + 16, c21);
      _mm512_storeu_ps(c + 3 * M, c30); _mm512_storeu_ps(c + 3 * M + 16, c31);
      _mm512_storeu_ps(c + 4 * M, c40); _mm512_storeu_ps(c + 4 * M + 16, c41);
      _mm512_storeu_ps(c + 5 * M, c50); _mm512_storeu_ps(c + 5 * M + 16, c51);
    }
    for (; n < N; n++) {  /* row tail: one row x 32 columns */
      __m512 c0 = _mm512_setzero_ps(), c1 = c0;
      const float *a0 = A + n * K, *b = B + j;
      for (size_t k = 0; k < K; k++, b += M) {
        const __m512 a = _mm512_set1_ps(a0[k]);
        c0 = _mm512_fmadd_ps(a, _mm512_loadu_ps(b), c0);
        c1 = _mm512_fmadd_ps(a, _mm512_loadu_ps(b + 16), c1);
      }
      _mm512_storeu_ps(C + n * M + j, c0);
      _mm512_storeu_ps(C + n * M + j + 16, c1);
    }
  }
  if (j < M) { /* column tail (< 32 columns, e.g. the MLP's 64 -> 1 head): eight independent row chains in flight */
    size_t n = 0;
    for (; n + 8 <= N; n += 8)
      for (size_t jj = j; jj < M; jj++) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float *a0 = A + n * K;
        for (size_t k = 0; k < K; k++) {
          const float bk = B[k * M + jj];
          for (int r = 0; r < 8; r++) acc[r] = fmaf(a0[(size_t)r * K + k], bk, acc[r]);
        }
        for (int r = 0; r < 8; r++) C[(n + (size_t)r) * M + jj] = acc[r];
      }
    for (; n < N; n++)
      for (size_t jj = j; jj < M; jj++) {
        float acc = 0.0f;
        for (size_t k = 0; k < K; k++) acc = fmaf(A[n * K + k], B[k * M + jj], acc);
        C[n * M + jj] = acc;
      }
  }
}

__attribute__((target("avx2,fma"))) static void gemm_block_avx2(const float *A, const float *B, float *C, size_t N, size_t K, size_t M) {
  size_t j = 0;
  for (; j + 16 <= M; j += 16) {
    size_t n = 0;
    for (; n + 6 <= N; n += 6) {
      __m256 c00 = _mm256_setzero_ps(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00, c40 = c00, c41 = c00, c50 = c00, c51 = c00;
      const float *a0 = A + n * K, *b = B + j;
      for (size_t k = 0; k < K; k++, b += M) {
        const __m256 b0 = _mm256_loadu_ps(b), b1 = _mm256_loadu_ps(b + 8);
        __m256 a;
        a = _mm256_broadcast_ss(a0 + k);         c00 = _mm256_fmadd_ps(a, b0, c00); c01 = _mm256_fmadd_ps(a, b1, c01);
        a = _mm256_broadcast_ss(a0 + K + k);     c10 = _mm256_fmadd_ps(a, b0, c10); c11 = _mm256_fmadd_ps(a, b1, c11);
        a = _mm256_broadcast_ss(a0 + 2 * K + k); c20 = _mm256_fmadd_ps(a, b0, c20); c21 = _mm256_fmadd_ps(a, b1, c21);
        a = _mm256_broadcast_ss(a0 + 3 * K + k); c30 = _mm256_fmadd_ps(a, b0, c30); c31 = _mm256_fmadd_ps(a, b1, c31);
        a = _mm256_broadcast_ss(a0 + 4 * K + k); c40 = _mm256_fmadd_ps(a, b0, c40); c41 = _mm256_fmadd_ps(a, b1, c41);
        a = _mm256_broadcast_ss(a0 + 5 * K + k); c50 = _mm256_fmadd_ps(a, b0, c50); c51 = _mm256_fmadd_ps(a, b1, c51);
      }
      float *c = C + n * M + j;
      _mm256_storeu_ps(c, c00);         _mm256_storeu_ps(c + 8, c01);
      _mm256_storeu_ps(c + M, c10);     _mm256_storeu_ps(c + M + 8, c11);
      _mm256_storeu_ps(c + 2 * M, c20); _mm256_storeu_ps(c + 2 * M + 8, c21);
      _mm256_storeu_ps(c + 3 * M, c30); _mm256_storeu_ps(c + 3 * M + 8, c31);
      _mm256_storeu_ps(c + 4 * M, c40); _mm256_storeu_ps(c + 4 * M + 8, c41);
      _mm256_storeu_ps(c + 5 * M, c50); _mm256_storeu_ps(c + 5 * M + 8, c51);
    }
    for (; n < N; n++) {
      __m256 c0 = _mm256_setzero_ps(), c1 = c0;
      const float *a0 = A + n * K, *b = B + j;
      for (size_t k = 0; k < K; k++, b += M) {
        const __m256 a = _mm256_broadcast_ss(a0 + k);
        c0 = _mm256_fmadd_ps(a, _mm256_loadu_ps(b), c0);
        c1 = _mm256_fmadd_ps(a, _mm256_loadu_ps(b + 8), c1);
      }
      _mm256_storeu_ps(C + n * M + j, c0);
      _mm256_storeu_ps(C + n * M + j + 8, c1);
    }
  }
  for (size_t n = 0; j < M && n < N; n++)
    for (size_t jj = j; jj < M; jj++) {
      float acc = 0.0f;
      for (size_t k = 0; k < K; k++) acc = fmaf(A[n * K + k], B[k * M + jj], acc);
      C[n * M + jj] = acc;
    }
}

/* C[n][m] = sum_k A[n][k] * B[k][m], k-ordered fmaf chain from 0, vectorisable over m. */
static void gemm_nn(const float *A, const float *B, float *C, size_t N, size_t K, size_t M) {
  if (t_gemm_blocked) {
    if (__builtin_cpu_supports("avx512f")) { gemm_block_avx512(A, B, C, N, K, M); return; }
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) { gemm_block_avx2(A, B, C, N, K, M); return; }
  }
  for (size_t n = 0; n < N; n++) {
    float *c = C + n * M;
    for (size_t j = 0; j < M; j++) c[j] = 0.0f;
    const float *a = A + n * K;
    for (size_t k = 0; k < K; k++) {
      const float ak = a[k];
      const float *b = B + k * M;
      for (size_t j = 0; j < M; j++) c[j] = fmaf(ak, b[j], c[j]);
    }
  }
}

/* test hook: the calling thread's later orc_predict calls use the blocked GEMM (1) or the plain loop (0) */
void orc_set_blocked_gemm(int on) { t_gemm_blocked = on; }

/* numpy-style broadcast of two shapes; returns rank or -1 */
static int bcast_shape(const Tensor *a, const Tensor *b, int64_t *out) {
  int r = a->rank > b->rank ? a->rank : b->rank;
  for (int i = 0; i < r; i++) {
    int ia = i - (r - a->rank), ib = i - (r - b->rank);
    int64_t da = ia >= 0 ? a->dims[ia] : 1, db = ib >= 0 ? b->dims[ib] : 1;
    if (da != db && da != 1 && db != 1) return -1;
    out[i] = da == 1 ? db : da;
  }
  return r;
}

static size_t bcast_index(const Tensor *t, int r, const int64_t *odims, size_t flat) {
  /* map flat index in the broadcast result to flat index in t */
  size_t idx = 0, stride = 1;
  for (int i = r - 1; i >= 0; i--) {
    size_t coord = flat % (size_t)odims[i];
    flat /= (size_t)odims[i];
    int it = i - (r - t->rank);
    if (it >= 0) {
      if (t->dims[it] != 1) idx += coord * stride;
      stride *= (size_t)t->dims[it];
    }
  }
  return idx;
}

static int op_binary(Exec *x, const Node *nd, char kind) {
  const Tensor *a = get_in(x, nd, 0), *b = get_in(x, nd, 1);
  if (!a || !b) FAIL("%s: missing input", nd->op);
  if (a->dtype != b->dtype) FAIL("%s: operand types differ", nd->op);
  int64_t od[MAXRANK];
  int r = bcast_shape(a, b, od);
  if (r < 0) FAIL("%s: shapes not broadcastable", nd->op);
  const int dt = a->dtype;
  Tensor *o = env_new(&x->env, nd->out[0], dt, r, od);
  a = get_in(x, nd, 0);
  b = get_in(x, nd, 1);
  int same = (a->n == o->n && b->n == o->n);
  if (dt == DT_INT64) { /* integer shape arithmetic (exporter sub-graphs around Reshape) */
    for (size_t i = 0; i < o->n; i++) {
      int64_t va = a->i64[same ? i : bcast_index(a, r, od, i)], vb = b->i64[same ? i : bcast_index(b, r, od, i)], v;
      switch (kind) {
        case '+': v = va + vb; break;
        case '-': v = va - vb; break;
        case '*': v = va * vb; break;
        case '/': if (vb == 0) FAIL("%s: integer division by zero", nd->op); v = va / vb; break;
        case 'm': v = va < vb ? va : vb; break;
        case 'M': v = va > vb ? va : vb; break;
        default: FAIL("%s: not defined on integers", nd->op);
      }
      o->i64[i] = v;
    }
    return 0;
  }
  for (size_t i = 0; i < o->n; i++) {
    float va = a->f[same ? i : bcast_index(a, r, od, i)];
    float vb = b->f[same ? i : bcast_index(b, r, od, i)];
    float v;
    switch (kind) {
      case '+': v = va + vb; break;
      case '-': v = va - vb; break;
      case '*': v = va * vb; break;
      case '/': v = va / vb; break;
      case 'm': v = fminf(va, vb); break;
      case 'M': v = fmaxf(va, vb); break;
      case '^': v = powf(va, vb); break;
      default: v = va >= 0.0f ? va : vb * va; break; /* PRelu: b = slope */
    }
    o->f[i] = v;
  }
  return 0;
}

static int op_unary(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("%s: missing input", nd->op);
  if (a->dtype != DT_FLOAT) FAIL("%s: only f32 supported", nd->op);
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, a->rank, a->dims);
  a = get_in(x, nd, 0);
  const char *op = nd->op;
  if (!strcmp(op, "Relu")) {
    for (size_t i = 0; i < o->n; i++) o->f[i] = a->f[i] > 0.0f ? a->f[i] : 0.0f;
  } else if (!strcmp(op, "Sigmoid")) {
    for (size_t i = 0; i < o->n; i++) o->f[i] = 1.0f / (1.0f + expf(-a->f[i]));
  } else if (!strcmp(op, "Tanh")) {
    for (size_t i = 0; i < o->n; i++) o->f[i] = tanhf(a->f[i]);
  } else if (!strcmp(op, "LeakyRelu")) {
    float alpha = attr_f(nd, "alpha", 0.01f);
    for (size_t i = 0; i < o->n; i++) o->f[i] = a->f[i] >= 0.0f ? a->f[i] : alpha * a->f[i];
  } else if (!strcmp(op, "Identity") || !strcmp(op, "Dropout")) {
    memcpy(o->f, a->f, o->n * 4);
  } else if (!strcmp(op, "Clip")) {
    float lo = -INFINITY, hi = INFINITY;
    const Attr *amin = find_attr(nd, "min"), *amax = find_attr(nd, "max");
    if (amin) lo = amin->f;
    if (amax) hi = amax->f;
    const Tensor *tmin = get_in(x, nd, 1), *tmax = get_in(x, nd, 2);
    if (tmin && tmin->n == 1) lo = tmin->f[0];
    if (tmax && tmax->n == 1) hi = tmax->f[0];
    for (size_t i = 0; i < o->n; i++) {
      float v = a->f[i];
      o->f[i] = v < lo ? lo : (v > hi ? hi : v);
    }
  } else if (!strcmp(op, "Elu")) {
    float alpha = attr_f(nd, "alpha", 1.0f);
    for (size_t i = 0; i < o->n; i++) o->f[i] = a->f[i] >= 0.0f ? a->f[i] : alpha * (expf(a->f[i]) - 1.0f);
  } else if (!strcmp(op, "Selu")) {
    float alpha = attr_f(nd, "alpha", 1.67326319217681884765625f), gamma = attr_f(nd, "gamma", 1.05070102214813232421875f);
    for (size_t i = 0; i < o->n; i++) o->f[i] = a->f[i] > 0.0f ? gamma * a->f[i] : gamma * (alpha * expf(a->f[i]) - alpha);
  } else if (!strcmp(op, "HardSigmoid")) {
    float alpha = attr_f(nd, "alpha", 0.2f), beta = attr_f(nd, "beta", 0.5f);
    for (size_t i = 0; i < o->n; i++) o->f[i] = fmaxf(0.0f, fminf(1.0f, alpha * a->f[i] + beta));
  } else if (!strcmp(op, "Gelu")) {
    const Attr *ap = find_attr(nd, "approximate");
    if (ap && ap->s && strcmp(ap->s, "none")) FAIL("Gelu: only the exact form");
    for (size_t i = 0; i < o->n; i++) o->f[i] = 0.5f * a->f[i] * (1.0f + erff(a->f[i] * 0.707106781186547524f));
  } else {
#define ORC_UN(NAME, EXPR)                                            \
  if (!strcmp(op, NAME)) {                                            \
    for (size_t i = 0; i < o->n; i++) { const float v = a->f[i]; o->f[i] = (EXPR); } \
    return 0;                                                         \
  }
    ORC_UN("Exp", expf(v))
    ORC_UN("Log", logf(v))
    ORC_UN("Sqrt", sqrtf(v))
    ORC_UN("Neg", -v)
    ORC_UN("Abs", fabsf(v))
    ORC_UN("Softplus", logf(expf(v) + 1.0f))
    ORC_UN("HardSwish", v * fmaxf(0.0f, fminf(1.0f, v * (1.0f / 6.0f) + 0.5f)))
    ORC_UN("Erf", erff(v))
    ORC_UN("Reciprocal", 1.0f / v)
    ORC_UN("Floor", floorf(v))
    ORC_UN("Ceil", ceilf(v))
    ORC_UN("Softsign", v / (1.0f + fabsf(v)))
    ORC_UN("Round", rintf(v))
#undef ORC_UN
    FAIL("unsupported unary op %s", op);
  }
  return 0;
}

/* ---- integer / shape sub-graph operators and the few data-movement ops exported models use ---- */
static int op_shape(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("Shape: missing input");
  int64_t r = a->rank, st = attr_i(nd, "start", 0), en = attr_i(nd, "end", r);
  if (st < 0) st += r;
  if (en < 0) en += r;
  if (st < 0) st = 0;
  if (st > r) st = r;
  if (en < st) en = st;
  if (en > r) en = r;
  int64_t cnt = en - st, dims[MAXRANK];
  for (int64_t i = 0; i < cnt; i++) dims[i] = a->dims[st + i];
  Tensor *o = env_new(&x->env, nd->out[0], DT_INT64, 1, &cnt);
  memcpy(o->i64, dims, (size_t)cnt * 8);
  return 0;
}

/* Gather along `axis`: out.shape = data.shape[:axis] + indices.shape + data.shape[axis+1:] */
static int op_gather(Exec *x, const Node *nd) {
  const Tensor *d = get_in(x, nd, 0), *ix = get_in(x, nd, 1);
  if (!d || !ix || ix->dtype != DT_INT64) FAIL("Gather: needs data and int64 indices");
  int64_t axis = attr_i(nd, "axis", 0);
  if (axis < 0) axis += d->rank;
  if (d->rank < 1 || axis < 0 || axis >= d->rank) FAIL("Gather: axis out of range");
  if (d->rank - 1 + ix->rank > MAXRANK) FAIL("Gather: rank too high");
  int64_t od[MAXRANK];
  int r = 0;
  size_t outer = 1, inner = 1;
  for (int i = 0; i < axis; i++) { od[r++] = d->dims[i]; outer *= (size_t)d->dims[i]; }
  for (int i = 0; i < ix->rank; i++) od[r++] = ix->dims[i];
  for (int i = (int)axis + 1; i < d->rank; i++) { od[r++] = d->dims[i]; inner *= (size_t)d->dims[i]; }
  const size_t len = (size_t)d->dims[axis], ni = ix->n;
  const int dtype = d->dtype;
  Tensor *o = env_new(&x->env, nd->out[0], dtype, r, od);
  d = get_in(x, nd, 0);
  ix = get_in(x, nd, 1);
  for (size_t u = 0; u < outer; u++)
    for (size_t j = 0; j < ni; j++) {
      int64_t k = ix->i64[j];
      if (k < 0) k += (int64_t)len;
      if (k < 0 || (size_t)k >= len) FAIL("Gather: index out of range");
      for (size_t v = 0; v < inner; v++) {
        const size_t src = (u * len + (size_t)k) * inner + v, dst = (u * ni + j) * inner + v;
        if (dtype == DT_FLOAT) o->f[dst] = d->f[src];
        else o->i64[dst] = d->i64[src];
      }
    }
  return 0;
}

/* copies dims[axis] in [b, e) of `a` into a new tensor named `name` */
static int slice_axis(Exec *x, const char *name, const char *in_name, int axis, int64_t b, int64_t e) {
  long ai = env_find(&x->env, in_name);
  const Tensor *a = ai >= 0 ? &x->env.v[ai] : find_init(x->m, in_name);
  if (!a) FAIL("slice: input '%s' not found", in_name);
  int64_t od[MAXRANK];
  memcpy(od, a->dims, sizeof od);
  od[axis] = e - b;
  const int dt = a->dtype, rank = a->rank;
  Tensor *o = env_new(&x->env, name, dt, rank, od);
  ai = env_find(&x->env, in_name);
  a = ai >= 0 ? &x->env.v[ai] : find_init(x->m, in_name);
  size_t outer = 1, inner = 1, esz = dt == DT_FLOAT ? 4 : 8;
  for (int i = 0; i < axis; i++) outer *= (size_t)a->dims[i];
  for (int i = axis + 1; i < rank; i++) inner *= (size_t)a->dims[i];
  const char *src = dt == DT_FLOAT ? (const char *)a->f : (const char *)a->i64;
  char *dst = dt == DT_FLOAT ? (char *)o->f : (char *)o->i64;
  for (size_t u = 0; u < outer; u++)
    memcpy(dst + u * (size_t)(e - b) * inner * esz, src + (u * (size_t)a->dims[axis] + (size_t)b) * inner * esz, (size_t)(e - b) * inner * esz);
  return 0;
}

static int op_slice(Exec *x, const Node *nd) {
  const Tensor *d = get_in(x, nd, 0);
  if (!d || d->rank < 1) FAIL("Slice: bad data");
  int64_t st, en, step = 1, axis = 0;
  const Tensor *ts = get_in(x, nd, 1), *te = get_in(x, nd, 2), *ta = get_in(x, nd, 3), *tp = get_in(x, nd, 4);
  if (ts && te) {
    if (ts->n != 1 || te->n != 1) FAIL("Slice: one axis only");
    st = ts->i64[0];
    en = te->i64[0];
    if (ta) axis = ta->i64[0];
    if (tp) step = tp->i64[0];
  } else {
    const Attr *as = find_attr(nd, "starts"), *ae = find_attr(nd, "ends"), *aa = find_attr(nd, "axes");
    if (!as || !ae || as->nints != 1 || ae->nints != 1) FAIL("Slice: starts/ends");
    st = as->ints[0];
    en = ae->ints[0];
    if (aa && aa->nints == 1) axis = aa->ints[0];
  }
  if (step != 1) FAIL("Slice: step must be 1");
  if (axis < 0) axis += d->rank;
  if (axis < 0 || axis >= d->rank) FAIL("Slice: axis out of range");
  int64_t len = d->dims[axis];
  if (st < 0) st += len;
  if (en < 0) en += len;
  if (st < 0) st = 0;
  if (st > len) st = len;
  if (en < st) en = st;
  if (en > len) en = len;
  return slice_axis(x, nd->out[0], nd->in[0], (int)axis, st, en);
}

static int op_split(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("Split: missing input");
  int64_t axis = attr_i(nd, "axis", 0);
  if (axis < 0) axis += a->rank;
  if (axis < 0 || axis >= a->rank) FAIL("Split: axis out of range");
  const int64_t K = a->dims[axis];
  int64_t sizes[64];
  size_t ns = 0;
  const Tensor *tsz = get_in(x, nd, 1);
  const Attr *asz = find_attr(nd, "split");
  if (tsz && tsz->dtype == DT_INT64) { ns = tsz->n; if (ns > 64) FAIL("Split: too many pieces"); for (size_t i = 0; i < ns; i++) sizes[i] = tsz->i64[i]; }
  else if (asz) { ns = asz->nints; if (ns > 64) FAIL("Split: too many pieces"); for (size_t i = 0; i < ns; i++) sizes[i] = asz->ints[i]; }
  else {
    ns = (size_t)attr_i(nd, "num_outputs", (int64_t)nd->nout);
    if (ns < 1 || ns > 64) FAIL("Split: bad piece count");
    int64_t each = (K + (int64_t)ns - 1) / (int64_t)ns;
    for (size_t i = 0; i < ns; i++) { int64_t left = K - (int64_t)i * each; sizes[i] = left < each ? left : each; }
  }
  if (ns != nd->nout) FAIL("Split: %zu sizes for %zu outputs", ns, nd->nout);
  int64_t off = 0;
  for (size_t i = 0; i < ns; i++) {
    if (sizes[i] < 0 || off + sizes[i] > K) FAIL("Split: sizes exceed the axis");
    if (nd->out[i][0] && slice_axis(x, nd->out[i], nd->in[0], (int)axis, off, off + sizes[i])) return -1;
    off += sizes[i];
  }
  if (off != K) FAIL("Split: sizes do not cover the axis");
  return 0;
}

static int op_cast(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("Cast: missing input");
  int64_t to = attr_i(nd, "to", DT_FLOAT);
  int to_int = to == DT_INT64 || to == DT_INT32, to_f = to == DT_FLOAT || to == DT_DOUBLE;
  if (!to_int && !to_f) FAIL("Cast: only f32/f64/int32/int64 targets");
  Tensor *o = env_new(&x->env, nd->out[0], to_int ? DT_INT64 : DT_FLOAT, a->rank, a->dims);
  a = get_in(x, nd, 0);
  for (size_t i = 0; i < o->n; i++) {
    if (to_int) o->i64[i] = a->dtype == DT_INT64 ? a->i64[i] : (int64_t)a->f[i]; /* C conversion truncates toward zero */
    else o->f[i] = a->dtype == DT_INT64 ? (float)a->i64[i] : a->f[i];
  }
  return 0;
}

static int op_concat(Exec *x, const Node *nd) {
  if (nd->nin < 1) FAIL("Concat: no inputs");
  const Tensor *a0 = get_in(x, nd, 0);
  if (!a0) FAIL("Concat: missing input");
  int rank = a0->rank, dt = a0->dtype;
  int64_t axis = attr_i(nd, "axis", 0);
  if (axis < 0) axis += rank;
  if (rank == 0 || axis < 0 || axis >= rank) FAIL("Concat: bad axis");
  int64_t od[MAXRANK];
  memcpy(od, a0->dims, sizeof od);
  od[axis] = 0;
  for (size_t k = 0; k < nd->nin; k++) {
    const Tensor *t = get_in(x, nd, k);
    if (!t || t->rank != rank || t->dtype != dt) FAIL("Concat: operand mismatch");
    for (int i = 0; i < rank; i++)
      if (i != axis && t->dims[i] != a0->dims[i]) FAIL("Concat: shape mismatch");
    od[axis] += t->dims[axis];
  }
  Tensor *o = env_new(&x->env, nd->out[0], dt, rank, od);
  size_t outer = 1, inner = 1;
  for (int i = 0; i < axis; i++) outer *= (size_t)od[i];
  for (int i = (int)axis + 1; i < rank; i++) inner *= (size_t)od[i];
  size_t off = 0, esz = dt == DT_FLOAT ? 4 : 8;
  char *dst = dt == DT_FLOAT ? (char *)o->f : (char *)o->i64;
  for (size_t k = 0; k < nd->nin; k++) {
    const Tensor *t = get_in(x, nd, k);
    const char *src = dt == DT_FLOAT ? (const char *)t->f : (const char *)t->i64;
    size_t blk = (size_t)t->dims[axis] * inner;
    for (size_t u = 0; u < outer; u++)
      memcpy(dst + (u * (size_t)od[axis] * inner + off) * esz, src + u * blk * esz, blk * esz);
    off += blk;
  }
  return 0;
}

static int op_reduce_mean(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT) FAIL("ReduceMean: bad input");
  int red[MAXRANK] = {0};
  const Tensor *ax = get_in(x, nd, 1);
  const Attr *aax = find_attr(nd, "axes");
  const int64_t *av = NULL;
  size_t na = 0;
  if (ax && ax->dtype == DT_INT64) { av = ax->i64; na = ax->n; }
  else if (aax) { av = aax->ints; na = aax->nints; }
  if (na == 0) for (int i = 0; i < a->rank; i++) red[i] = 1;
  for (size_t k = 0; k < na; k++) {
    int64_t v = av[k] < 0 ? av[k] + a->rank : av[k];
    if (v < 0 || v >= a->rank) FAIL("ReduceMean: axis out of range");
    red[v] = 1;
  }
  int keep = attr_i(nd, "keepdims", 1) != 0;
  int64_t od[MAXRANK], full[MAXRANK];
  int r = 0;
  size_t cnt = 1;
  for (int i = 0; i < a->rank; i++) {
    full[i] = red[i] ? 1 : a->dims[i];
    if (red[i]) cnt *= (size_t)a->dims[i];
    if (!red[i] || keep) od[r++] = full[i];
  }
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, r, od);
  a = get_in(x, nd, 0);
  for (size_t i = 0; i < o->n; i++) o->f[i] = 0.0f;
  /* sequential sum in input order per output element, then one division (as GlobalAveragePool above) */
  for (size_t flat = 0; flat < a->n; flat++) {
    size_t rem = flat, oi = 0, stride = 1;
    for (int i = a->rank - 1; i >= 0; i--) {
      size_t c = rem % (size_t)a->dims[i];
      rem /= (size_t)a->dims[i];
      if (!red[i]) { oi += c * stride; stride *= (size_t)a->dims[i]; }
    }
    o->f[oi] += a->f[flat];
  }
  for (size_t i = 0; i < o->n; i++) o->f[i] = o->f[i] / (float)cnt;
  return 0;
}

static int op_argmax(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT) FAIL("ArgMax: bad input");
  int64_t axis = attr_i(nd, "axis", 0);
  if (axis < 0) axis += a->rank;
  if (axis < 0 || axis >= a->rank) FAIL("ArgMax: axis out of range");
  if (attr_i(nd, "select_last_index", 0)) FAIL("ArgMax: select_last_index=1");
  int keep = attr_i(nd, "keepdims", 1) != 0;
  int64_t od[MAXRANK] = {0};
  int r = 0;
  size_t outer = 1, inner = 1, len = (size_t)a->dims[axis];
  for (int i = 0; i < a->rank; i++) {
    if (i < axis) outer *= (size_t)a->dims[i];
    if (i > axis) inner *= (size_t)a->dims[i];
    if (i != axis) od[r++] = a->dims[i];
    else if (keep) od[r++] = 1;
  }
  Tensor *o = env_new(&x->env, nd->out[0], DT_INT64, r, od);
  a = get_in(x, nd, 0);
  for (size_t u = 0; u < outer; u++)
    for (size_t v = 0; v < inner; v++) {
      const float *src = a->f + u * len * inner + v;
      float best = src[0];
      int64_t bi = 0;
      for (size_t j = 1; j < len; j++)
        if (src[j * inner] > best) { best = src[j * inner]; bi = (int64_t)j; }
      o->i64[u * inner + v] = bi;
    }
  return 0;
}

static int op_matmul(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0), *b = get_in(x, nd, 1);
  if (!a || !b) FAIL("MatMul: missing input");
  if (a->dtype != DT_FLOAT || b->dtype != DT_FLOAT) FAIL("MatMul: only f32 supported");
  if (a->rank < 2 || b->rank != 2) FAIL("MatMul: only [..,K] x [K,M] supported");
  size_t K = (size_t)a->dims[a->rank - 1], M = (size_t)b->dims[1];
  if ((size_t)b->dims[0] != K) FAIL("MatMul: inner dimensions differ (%zu vs %lld)", K, (long long)b->dims[0]);
  int64_t od[MAXRANK];
  memcpy(od, a->dims, sizeof(int64_t) * (size_t)a->rank);
  od[a->rank - 1] = (int64_t)M;
  size_t N = a->n / (K ? K : 1);
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, a->rank, od);
  a = get_in(x, nd, 0);
  b = get_in(x, nd, 1);
  gemm_nn(a->f, b->f, o->f, N, K, M);
  return 0;
}

/* Gemm: Y = alpha * A' * B' + beta * C  (A' = transA ? A^T : A) */
static int op_gemm(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0), *b = get_in(x, nd, 1), *c = get_in(x, nd, 2);
  if (!a || !b) FAIL("Gemm: missing input");
  if (a->rank != 2 || b->rank != 2) FAIL("Gemm: inputs must be rank 2");
  int tA = (int)attr_i(nd, "transA", 0), tB = (int)attr_i(nd, "transB", 0);
  float alpha = attr_f(nd, "alpha", 1.0f), beta = attr_f(nd, "beta", 1.0f);
  size_t N = (size_t)(tA ? a->dims[1] : a->dims[0]), K = (size_t)(tA ? a->dims[0] : a->dims[1]);
  size_t Kb = (size_t)(tB ? b->dims[1] : b->dims[0]), M = (size_t)(tB ? b->dims[0] : b->dims[1]);
  if (K != Kb) FAIL("Gemm: inner dimensions differ (%zu vs %zu)", K, Kb);
  int64_t od[2] = {(int64_t)N, (int64_t)M};
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, 2, od);
  a = get_in(x, nd, 0);
  b = get_in(x, nd, 1);
  c = get_in(x, nd, 2);
  float *At = NULL, *Bt = NULL;
  const float *A = a->f, *B = b->f;
  if (tA) {
    At = (float *)xmalloc(N * K * 4);
    for (size_t n = 0; n < N; n++)
      for (size_t k = 0; k < K; k++) At[n * K + k] = a->f[k * N + n];
    A = At;
  }
  if (tB) {
    Bt = (float *)xmalloc(K * M * 4);
    for (size_t k = 0; k < K; k++)
      for (size_t j = 0; j < M; j++) Bt[k * M + j] = b->f[j * K + k];
    B = Bt;
  }
  gemm_nn(A, B, o->f, N, K, M);
  free(At);
  free(Bt);
  if (alpha != 1.0f)
    for (size_t i = 0; i < o->n; i++) o->f[i] *= alpha;
  if (c) {
    int64_t bd[MAXRANK];
    if (bcast_shape(o, c, bd) != 2 || bd[0] != od[0] || bd[1] != od[1]) FAIL("Gemm: C not broadcastable to [N,M]");
    for (size_t i = 0; i < o->n; i++) {
      float cv = c->f[bcast_index(c, 2, od, i)];
      o->f[i] = o->f[i] + (beta == 1.0f ? cv : beta * cv);
    }
  }
  return 0;
}

static int op_softmax(Exec *x, const Node *nd, int logsm) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("Softmax: missing input");
  int64_t opset = x->m->opset;
  int64_t axis = attr_i(nd, "axis", opset >= 13 ? -1 : 1);
  if (axis < 0) axis += a->rank;
  if (axis < 0 || axis >= a->rank) FAIL("Softmax: axis out of range");
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, a->rank, a->dims);
  a = get_in(x, nd, 0);
  size_t outer = 1, len, inner = 1;
  if (opset >= 13) { /* single axis */
    for (int i = 0; i < axis; i++) outer *= (size_t)a->dims[i];
    len = (size_t)a->dims[axis];
    for (int i = (int)axis + 1; i < a->rank; i++) inner *= (size_t)a->dims[i];
  } else { /* coerce to 2D: [prod(d[:axis]), prod(d[axis:])] */
    for (int i = 0; i < axis; i++) outer *= (size_t)a->dims[i];
    len = 1;
    for (int i = (int)axis; i < a->rank; i++) len *= (size_t)a->dims[i];
  }
  for (size_t ou = 0; ou < outer; ou++)
    for (size_t in = 0; in < inner; in++) {
      const float *src = a->f + ou * len * inner + in;
      float *dst = o->f + ou * len * inner + in;
      float mx = -INFINITY;
      for (size_t j = 0; j < len; j++) mx = src[j * inner] > mx ? src[j * inner] : mx;
      float sum = 0.0f;
      for (size_t j = 0; j < len; j++) {
        float e = expf(src[j * inner] - mx);
        dst[j * inner] = e;
        sum += e;
      }
      if (logsm) {
        float ls = logf(sum);
        for (size_t j = 0; j < len; j++) dst[j * inner] = (src[j * inner] - mx) - ls;
      } else {
        for (size_t j = 0; j < len; j++) dst[j * inner] = dst[j * inner] / sum;
      }
    }
  return 0;
}

/* ---- ai.onnx.ml operators (the classical-ML nodes sklearn exporters emit; tract-onnx 0.22 ops/ml).
 * tract's sources are not in /root/reference; these follow the published ONNX-ML operator specification
 * (onnx/docs/Operators-ml.md) and the ONNX reference evaluator's arithmetic.  Parity for them is pinned on
 * the spec text only -- stated in DESIGN.md. */
static const char *attr_str(const Node *nd, const char *name, const char *dflt) {
  const Attr *a = find_attr(nd, name);
  return a && a->s ? a->s : dflt;
}

/* Scaler: Y = (X - offset) * scale, per feature (either attribute may hold one value for all features) */
static int op_ml_scaler(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT || a->rank != 2) FAIL("Scaler: expects an f32 [N,F] input");
  const Attr *off = find_attr(nd, "offset"), *sc = find_attr(nd, "scale");
  size_t F = (size_t)a->dims[1], N = (size_t)a->dims[0];
  if (off && off->nfloats != F && off->nfloats != 1) FAIL("Scaler: offset has %zu values for %zu features", off->nfloats, F);
  if (sc && sc->nfloats != F && sc->nfloats != 1) FAIL("Scaler: scale has %zu values for %zu features", sc->nfloats, F);
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, 2, a->dims);
  a = get_in(x, nd, 0);
  for (size_t n = 0; n < N; n++)
    for (size_t k = 0; k < F; k++) {
      float of = off && off->nfloats ? off->floats[off->nfloats == 1 ? 0 : k] : 0.0f;
      float s = sc && sc->nfloats ? sc->floats[sc->nfloats == 1 ? 0 : k] : 1.0f;
      o->f[n * F + k] = (a->f[n * F + k] - of) * s;
    }
  return 0;
}

/* post_transform over a [N,E] score matrix, in place */
static int ml_post_transform(Exec *x, const char *pt, float *s, size_t N, size_t E, const char *who) {
  if (!strcmp(pt, "NONE")) return 0;
  if (!strcmp(pt, "LOGISTIC")) {
    for (size_t i = 0; i < N * E; i++) s[i] = 1.0f / (1.0f + expf(-s[i]));
    return 0;
  }
  if (!strcmp(pt, "SOFTMAX")) {
    for (size_t n = 0; n < N; n++) {
      float *r = s + n * E, mx = -INFINITY, sum = 0.0f;
      for (size_t j = 0; j < E; j++) mx = r[j] > mx ? r[j] : mx;
      for (size_t j = 0; j < E; j++) { r[j] = expf(r[j] - mx); sum += r[j]; }
      for (size_t j = 0; j < E; j++) r[j] = r[j] / sum;
    }
    return 0;
  }
  FAIL("%s: post_transform %s is not supported", who, pt);
}

/* scores[n][e] = sum_k X[n][k] * coef[e][k] + intercept[e]  (coefficients are [E, F] row-major) */
static void ml_linear_scores(const float *X, const float *coef, const Attr *icpt, float *S, size_t N, size_t F, size_t E) {
  for (size_t n = 0; n < N; n++)
    for (size_t e = 0; e < E; e++) {
      float acc = 0.0f;
      for (size_t k = 0; k < F; k++) acc = fmaf(X[n * F + k], coef[e * F + k], acc);
      S[n * E + e] = icpt && icpt->nfloats ? acc + icpt->floats[e] : acc;
    }
}

/* LinearRegressor: Y = post_transform(X . coefficients^T + intercepts), `targets` rows of coefficients */
static int op_ml_linear_regressor(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT || a->rank != 2) FAIL("LinearRegressor: expects an f32 [N,F] input");
  const Attr *co = find_attr(nd, "coefficients"), *ic = find_attr(nd, "intercepts");
  size_t N = (size_t)a->dims[0], F = (size_t)a->dims[1], E = (size_t)attr_i(nd, "targets", 1);
  if (!co || E == 0 || co->nfloats != E * F) FAIL("LinearRegressor: coefficients must hold targets x features = %zu values", E * F);
  if (ic && ic->nfloats && ic->nfloats != E) FAIL("LinearRegressor: intercepts must hold %zu values", E);
  int64_t od[2] = {(int64_t)N, (int64_t)E};
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, 2, od);
  a = get_in(x, nd, 0);
  ml_linear_scores(a->f, co->floats, ic, o->f, N, F, E);
  return ml_post_transform(x, attr_str(nd, "post_transform", "NONE"), o->f, N, E, "LinearRegressor");
}

/* LinearClassifier: outputs (label [N] int64, scores [N,E] f32).  label = classlabels_ints[argmax of the raw
 * scores] (first maximum wins); scores = post_transform(raw).  One coefficient row per class label. */
static int op_ml_linear_classifier(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT || a->rank != 2) FAIL("LinearClassifier: expects an f32 [N,F] input");
  const Attr *co = find_attr(nd, "coefficients"), *ic = find_attr(nd, "intercepts"), *li = find_attr(nd, "classlabels_ints");
  if (find_attr(nd, "classlabels_strings")) FAIL("LinearClassifier: string class labels cannot be returned as numbers");
  if (!li || li->nints < 2) FAIL("LinearClassifier: needs at least two classlabels_ints");
  size_t N = (size_t)a->dims[0], F = (size_t)a->dims[1], E = li->nints;
  if (!co || co->nfloats != E * F) FAIL("LinearClassifier: coefficients must hold classes x features = %zu values", E * F);
  if (ic && ic->nfloats && ic->nfloats != E) FAIL("LinearClassifier: intercepts must hold %zu values", E);
  int64_t ld[1] = {(int64_t)N}, sd[2] = {(int64_t)N, (int64_t)E};
  env_new(&x->env, nd->out[0], DT_INT64, 1, ld);
  size_t li_idx = x->env.n - 1;
  const char *sname = nd->nout > 1 && nd->out[1][0] ? nd->out[1] : "\x01ml_scores";
  env_new(&x->env, sname, DT_FLOAT, 2, sd);
  Tensor *lab = &x->env.v[li_idx], *sc = &x->env.v[x->env.n - 1];
  a = get_in(x, nd, 0);
  ml_linear_scores(a->f, co->floats, ic, sc->f, N, F, E);
  for (size_t n = 0; n < N; n++) {
    size_t best = 0;
    for (size_t e = 1; e < E; e++)
      if (sc->f[n * E + e] > sc->f[n * E + best]) best = e;
    lab->i64[n] = li->ints[best];
  }
  return ml_post_transform(x, attr_str(nd, "post_transform", "NONE"), sc->f, N, E, "LinearClassifier");
}

/* Normalizer: each row divided by its max |x| (MAX, the default), sum |x| (L1) or sqrt(sum x^2) (L2); the divisor
 * is floored at 1e-30 so an all-zero row stays zero (ONNX reference evaluator). */
static int op_ml_normalizer(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT || a->rank != 2) FAIL("Normalizer: expects an f32 [N,F] input");
  const char *norm = attr_str(nd, "norm", "MAX");
  int mode = !strcmp(norm, "MAX") ? 0 : !strcmp(norm, "L1") ? 1 : !strcmp(norm, "L2") ? 2 : -1;
  if (mode < 0) FAIL("Normalizer: norm %s is not supported", norm);
  size_t N = (size_t)a->dims[0], F = (size_t)a->dims[1];
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, 2, a->dims);
  a = get_in(x, nd, 0);
  for (size_t n = 0; n < N; n++) {
    const float *r = a->f + n * F;
    float d = 0.0f;
    for (size_t k = 0; k < F; k++) {
      float v = fabsf(r[k]);
      d = mode == 0 ? (v > d ? v : d) : mode == 1 ? d + v : fmaf(r[k], r[k], d);
    }
    if (mode == 2) d = sqrtf(d);
    if (d < 1e-30f) d = 1e-30f;
    for (size_t k = 0; k < F; k++) o->f[n * F + k] = r[k] / d;
  }
  return 0;
}

/* ArrayFeatureExtractor: Y = X[..., indices] (gather along the last axis; indices int64 or whole-number f32).
 * The output takes the indices' shape behind X's leading axes; a 1-D X indexed by [N] gives [N]. */
static int op_ml_array_feature_extractor(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0), *ix = get_in(x, nd, 1);
  if (!a || !ix || a->rank < 1) FAIL("ArrayFeatureExtractor: missing input");
  size_t last = (size_t)a->dims[a->rank - 1], outer = a->n / (last ? last : 1), ni = ix->n;
  int a_vec = a->rank == 1;
  int64_t od[MAXRANK];
  int r = 0;
  if (a_vec) {
    for (int i = 0; i < ix->rank; i++) od[r++] = ix->dims[i];
    outer = 1;
  } else {
    for (int i = 0; i + 1 < a->rank; i++) od[r++] = a->dims[i];
    od[r++] = (int64_t)ni;
  }
  Tensor *o = env_new(&x->env, nd->out[0], a->dtype, r, od);
  a = get_in(x, nd, 0);
  ix = get_in(x, nd, 1);
  for (size_t u = 0; u < outer; u++)
    for (size_t j = 0; j < ni; j++) {
      int64_t k = ix->dtype == DT_INT64 ? ix->i64[j] : (int64_t)ix->f[j];
      if (k < 0 || (size_t)k >= last) FAIL("ArrayFeatureExtractor: index %lld out of range", (long long)k);
      if (a->dtype == DT_FLOAT) o->f[u * ni + j] = a->f[u * last + (size_t)k];
      else o->i64[u * ni + j] = a->i64[u * last + (size_t)k];
    }
  return 0;
}

static int op_reshape_like(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a) FAIL("%s: missing input", nd->op);
  int64_t od[MAXRANK];
  int r = 0;
  if (!strcmp(nd->op, "Flatten")) {
    int64_t axis = attr_i(nd, "axis", 1);
    if (axis < 0) axis += a->rank;
    if (axis < 0 || axis > a->rank) FAIL("Flatten: axis out of range");
    int64_t d0 = 1, d1 = 1;
    for (int i = 0; i < a->rank; i++) {
      if (i < axis) d0 *= a->dims[i];
      else d1 *= a->dims[i];
    }
    od[0] = d0;
    od[1] = d1;
    r = 2;
  } else if (!strcmp(nd->op, "Reshape")) {
    const Tensor *sh = get_in(x, nd, 1);
    const Attr *ash = find_attr(nd, "shape"); /* opset < 5 */
    const int64_t *sv;
    size_t ns;
    if (sh && sh->dtype == DT_INT64) { sv = sh->i64; ns = sh->n; }
    else if (ash) { sv = ash->ints; ns = ash->nints; }
    else FAIL("Reshape: shape must be a constant int64 tensor");
    if (ns > MAXRANK) FAIL("Reshape: rank too large");
    int64_t known = 1;
    int neg = -1;
    for (size_t i = 0; i < ns; i++) {
      int64_t d = sv[i];
      if (d == 0) {
        if (i >= (size_t)a->rank) FAIL("Reshape: 0-dim out of range");
        d = a->dims[i];
      }
      if (d == -1) { if (neg >= 0) FAIL("Reshape: more than one -1"); neg = (int)i; od[i] = 1; continue; }
      od[i] = d;
      known *= d;
    }
    r = (int)ns;
    if (neg >= 0) {
      if (known == 0 || a->n % (size_t)known) FAIL("Reshape: cannot infer -1");
      od[neg] = (int64_t)(a->n / (size_t)known);
    }
    if (dims_count(od, r) != a->n) FAIL("Reshape: element count mismatch");
  } else if (!strcmp(nd->op, "Squeeze") || !strcmp(nd->op, "Unsqueeze")) {
    const Tensor *ax = get_in(x, nd, 1);
    const Attr *aax = find_attr(nd, "axes");
    const int64_t *av = NULL;
    size_t na = 0;
    if (ax && ax->dtype == DT_INT64) { av = ax->i64; na = ax->n; }
    else if (aax) { av = aax->ints; na = aax->nints; }
    if (!strcmp(nd->op, "Squeeze")) {
      for (int i = 0; i < a->rank; i++) {
        int drop = 0;
        if (na == 0) drop = a->dims[i] == 1;
        for (size_t k = 0; k < na; k++) {
          int64_t v = av[k] < 0 ? av[k] + a->rank : av[k];
          if (v == i) drop = 1;
        }
        if (!drop) od[r++] = a->dims[i];
      }
    } else {
      int nr = a->rank + (int)na;
      if (nr > MAXRANK) FAIL("Unsqueeze: rank too large");
      int src = 0;
      for (int i = 0; i < nr; i++) {
        int ins = 0;
        for (size_t k = 0; k < na; k++) {
          int64_t v = av[k] < 0 ? av[k] + nr : av[k];
          if (v == i) ins = 1;
        }
        od[i] = ins ? 1 : a->dims[src++];
      }
      r = nr;
    }
  } else {
    FAIL("unsupported reshape-like op %s", nd->op);
  }
  Tensor *o = env_new(&x->env, nd->out[0], a->dtype, r, od);
  a = get_in(x, nd, 0);
  if (a->dtype == DT_FLOAT) memcpy(o->f, a->f, a->n * 4);
  else memcpy(o->i64, a->i64, a->n * 8);
  return 0;
}

/* spatial helper: resolves kernel/stride/pad/dilation attrs for 2-D ops */
typedef struct {
  int64_t kh, kw, sh, sw, pt, pl, pb, pr, dh, dw;
} Spatial;

static int spatial_attrs(Exec *x, const Node *nd, int64_t kh, int64_t kw, int64_t H, int64_t W, Spatial *s) {
  s->kh = kh; s->kw = kw;
  s->sh = s->sw = 1; s->dh = s->dw = 1;
  s->pt = s->pl = s->pb = s->pr = 0;
  const Attr *a;
  if ((a = find_attr(nd, "strides")) && a->nints == 2) { s->sh = a->ints[0]; s->sw = a->ints[1]; }
  if ((a = find_attr(nd, "dilations")) && a->nints == 2) { s->dh = a->ints[0]; s->dw = a->ints[1]; }
  if ((a = find_attr(nd, "pads")) && a->nints == 4) { s->pt = a->ints[0]; s->pl = a->ints[1]; s->pb = a->ints[2]; s->pr = a->ints[3]; }
  /* 1-D operators ([N,C,L] tensors run as [N,C,1,L]): one entry per attribute, two pads */
  if ((a = find_attr(nd, "strides")) && a->nints == 1) s->sw = a->ints[0];
  if ((a = find_attr(nd, "dilations")) && a->nints == 1) s->dw = a->ints[0];
  if ((a = find_attr(nd, "pads")) && a->nints == 2) { s->pl = a->ints[0]; s->pr = a->ints[1]; }
  if ((a = find_attr(nd, "auto_pad")) && a->s && strcmp(a->s, "NOTSET") != 0) {
    if (!strcmp(a->s, "VALID")) { s->pt = s->pl = s->pb = s->pr = 0; }
    else if (!strcmp(a->s, "SAME_UPPER") || !strcmp(a->s, "SAME_LOWER")) {
      int64_t oh = (H + s->sh - 1) / s->sh, ow = (W + s->sw - 1) / s->sw;
      int64_t ph = (oh - 1) * s->sh + (kh - 1) * s->dh + 1 - H, pw = (ow - 1) * s->sw + (kw - 1) * s->dw + 1 - W;
      if (ph < 0) ph = 0;
      if (pw < 0) pw = 0;
      int upper = !strcmp(a->s, "SAME_UPPER");
      s->pt = upper ? ph / 2 : ph - ph / 2; s->pb = ph - s->pt;
      s->pl = upper ? pw / 2 : pw - pw / 2; s->pr = pw - s->pl;
    } else FAIL("%s: unsupported auto_pad %s", nd->op, a->s);
  }
  return 0;
}

/* Pooling output extent (ONNX MaxPool/AveragePool): floor or, with ceil_mode=1, ceil -- and a last window that
 * would START beyond the input plus its leading pad is dropped. */
static int64_t pool_extent(int64_t in, int64_t p0, int64_t p1, int64_t k, int64_t d, int64_t s, int ceil_mode) {
  int64_t num = in + p0 + p1 - (d * (k - 1) + 1);
  int64_t o = (ceil_mode ? (num + s - 1) / s : num / s) + 1;
  if (ceil_mode && (o - 1) * s >= in + p0) o--;
  return o;
}

/* Conv (NCHW, 2-D): im2col per image then W[M x CKK] * col[CKK x P]; the dot product over
 * (c,kh,kw) is a k-ordered fmaf chain from 0; bias added afterwards. */
static int op_conv(Exec *x, const Node *nd) {
  const Tensor *in = get_in(x, nd, 0), *w = get_in(x, nd, 1), *bias = get_in(x, nd, 2);
  if (!in || !w) FAIL("Conv: missing input");
  const int one_d = in->rank == 3 && w->rank == 3; /* Conv1d: [N,C,L] as [N,C,1,L], kernel [M,C/g,k] as [M,C/g,1,k] */
  if (!one_d && (in->rank != 4 || w->rank != 4)) FAIL("Conv: only 1-D / 2-D convolutions supported");
  int64_t N = in->dims[0], C = in->dims[1], H = one_d ? 1 : in->dims[2], W = one_d ? in->dims[2] : in->dims[3];
  int64_t M = w->dims[0], Cg = w->dims[1], kh = one_d ? 1 : w->dims[2], kw = one_d ? w->dims[2] : w->dims[3];
  int64_t G = attr_i(nd, "group", 1);
  if (G < 1 || C != Cg * G || M % G) FAIL("Conv: channel/group mismatch");
  Spatial s;
  if (spatial_attrs(x, nd, kh, kw, H, W, &s)) return -1;
  int64_t OH = (H + s.pt + s.pb - (s.dh * (kh - 1) + 1)) / s.sh + 1;
  int64_t OW = (W + s.pl + s.pr - (s.dw * (kw - 1) + 1)) / s.sw + 1;
  if (OH <= 0 || OW <= 0) FAIL("Conv: empty output");
  int64_t od[4] = {N, M, OH, OW};
  if (one_d) od[2] = OW;
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, one_d ? 3 : 4, od);
  in = get_in(x, nd, 0);
  w = get_in(x, nd, 1);
  bias = get_in(x, nd, 2);
  size_t P = (size_t)(OH * OW), CKK = (size_t)(Cg * kh * kw), Mg = (size_t)(M / G);
  float *col = (float *)xmalloc(CKK * P * 4);
  for (int64_t n = 0; n < N; n++)
    for (int64_t g = 0; g < G; g++) {
      for (int64_t c = 0; c < Cg; c++)
        for (int64_t i = 0; i < kh; i++)
          for (int64_t j = 0; j < kw; j++) {
            float *dst = col + (size_t)((c * kh + i) * kw + j) * P;
            const float *src = in->f + (size_t)((n * C + g * Cg + c) * H) * (size_t)W;
            for (int64_t oy = 0; oy < OH; oy++) {
              int64_t iy = oy * s.sh - s.pt + i * s.dh;
              for (int64_t ox = 0; ox < OW; ox++) {
                int64_t ix = ox * s.sw - s.pl + j * s.dw;
                dst[oy * OW + ox] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? src[iy * W + ix] : 0.0f;
              }
            }
          }
      float *out = o->f + (size_t)((n * M + g * (int64_t)Mg) * OH * OW);
      gemm_nn(w->f + (size_t)g * Mg * CKK, col, out, Mg, CKK, P);
      if (bias)
        for (size_t m = 0; m < Mg; m++) {
          float bv = bias->f[(size_t)g * Mg + m];
          for (size_t p = 0; p < P; p++) out[m * P + p] += bv;
        }
    }
  free(col);
  return 0;
}

/* BatchNormalization (inference): y = (x - mean) / sqrt(var + eps) * scale + B, evaluated in that
 * operator order in f32. */
static int op_batchnorm(Exec *x, const Node *nd) {
  const Tensor *in = get_in(x, nd, 0);
  if (!in || in->rank < 2) FAIL("BatchNormalization: bad input");
  float eps = attr_f(nd, "epsilon", 1e-5f);
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, in->rank, in->dims);
  in = get_in(x, nd, 0);
  const Tensor *sc = get_in(x, nd, 1), *bi = get_in(x, nd, 2), *mu = get_in(x, nd, 3), *var = get_in(x, nd, 4);
  if (!sc || !bi || !mu || !var) FAIL("BatchNormalization: missing parameter");
  size_t N = (size_t)in->dims[0], C = (size_t)in->dims[1], S = in->n / (N * C);
  if (sc->n != C || bi->n != C || mu->n != C || var->n != C) FAIL("BatchNormalization: parameter size mismatch");
  for (size_t n = 0; n < N; n++)
    for (size_t c = 0; c < C; c++) {
      float inv = 1.0f / sqrtf(var->f[c] + eps);
      const float *src = in->f + (n * C + c) * S;
      float *dst = o->f + (n * C + c) * S;
      for (size_t i = 0; i < S; i++) dst[i] = (src[i] - mu->f[c]) * inv * sc->f[c] + bi->f[c];
    }
  return 0;
}

static int op_pool(Exec *x, const Node *nd, int is_max) {
  const Tensor *in = get_in(x, nd, 0);
  const int one_d = in && in->rank == 3;
  if (!in || (in->rank != 4 && !one_d)) FAIL("%s: only 1-D / 2-D pooling supported", nd->op);
  const Attr *ks = find_attr(nd, "kernel_shape");
  if (!ks || ks->nints != (one_d ? 1u : 2u)) FAIL("%s: kernel_shape required", nd->op);
  int64_t N = in->dims[0], C = in->dims[1], H = one_d ? 1 : in->dims[2], W = one_d ? in->dims[2] : in->dims[3];
  Spatial s;
  if (spatial_attrs(x, nd, one_d ? 1 : ks->ints[0], one_d ? ks->ints[0] : ks->ints[1], H, W, &s)) return -1;
  int ceil_mode = attr_i(nd, "ceil_mode", 0) != 0;
  int count_pad = (int)attr_i(nd, "count_include_pad", 0);
  if (ceil_mode && count_pad && !is_max) FAIL("%s: ceil_mode=1 with count_include_pad=1 unsupported", nd->op);
  int64_t OH = pool_extent(H, s.pt, s.pb, s.kh, s.dh, s.sh, ceil_mode);
  int64_t OW = pool_extent(W, s.pl, s.pr, s.kw, s.dw, s.sw, ceil_mode);
  int64_t od[4] = {N, C, OH, OW};
  if (one_d) od[2] = OW;
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, one_d ? 3 : 4, od);
  in = get_in(x, nd, 0);
  for (int64_t nc = 0; nc < N * C; nc++) {
    const float *src = in->f + (size_t)nc * (size_t)(H * W);
    float *dst = o->f + (size_t)nc * (size_t)(OH * OW);
    for (int64_t oy = 0; oy < OH; oy++)
      for (int64_t ox = 0; ox < OW; ox++) {
        float acc = is_max ? -INFINITY : 0.0f;
        int cnt = 0;
        for (int64_t i = 0; i < s.kh; i++)
          for (int64_t j = 0; j < s.kw; j++) {
            int64_t iy = oy * s.sh - s.pt + i * s.dh, ix = ox * s.sw - s.pl + j * s.dw;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            float v = src[iy * W + ix];
            if (is_max) acc = v > acc ? v : acc;
            else acc += v;
            cnt++;
          }
        if (!is_max) acc = acc / (float)(count_pad ? s.kh * s.kw : (cnt ? cnt : 1));
        dst[oy * OW + ox] = acc;
      }
  }
  return 0;
}

static int op_global_avgpool(Exec *x, const Node *nd) {
  const Tensor *in = get_in(x, nd, 0);
  if (!in || in->rank < 3) FAIL("GlobalAveragePool: bad input");
  int64_t od[MAXRANK];
  for (int i = 0; i < in->rank; i++) od[i] = i < 2 ? in->dims[i] : 1;
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, in->rank, od);
  in = get_in(x, nd, 0);
  size_t NC = (size_t)(in->dims[0] * in->dims[1]), S = in->n / NC;
  for (size_t i = 0; i < NC; i++) {
    float acc = 0.0f;
    for (size_t j = 0; j < S; j++) acc += in->f[i * S + j];
    o->f[i] = acc / (float)S;
  }
  return 0;
}

static int op_global_maxpool(Exec *x, const Node *nd) {
  const Tensor *in = get_in(x, nd, 0);
  if (!in || in->rank < 3) FAIL("GlobalMaxPool: bad input");
  int64_t od[MAXRANK];
  for (int i = 0; i < in->rank; i++) od[i] = i < 2 ? in->dims[i] : 1;
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, in->rank, od);
  in = get_in(x, nd, 0);
  size_t NC = (size_t)(in->dims[0] * in->dims[1]), S = in->n / NC;
  for (size_t i = 0; i < NC; i++) {
    float mx = in->f[i * S];
    for (size_t j = 1; j < S; j++) mx = in->f[i * S + j] > mx ? in->f[i * S + j] : mx;
    o->f[i] = mx;
  }
  return 0;
}

/* Pad (mode constant): pads = [begin_0..begin_{r-1}, end_0..end_{r-1}], attribute (opset < 11) or input 1 */
static int op_pad(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0), *tp = get_in(x, nd, 1), *tv = get_in(x, nd, 2);
  if (!a || a->dtype != DT_FLOAT) FAIL("Pad: bad input");
  const Attr *am = find_attr(nd, "mode");
  if (am && am->s && strcmp(am->s, "constant")) FAIL("Pad: only constant mode");
  int64_t pads[2 * MAXRANK];
  const Attr *ap = find_attr(nd, "pads");
  size_t np = tp ? tp->n : (ap ? ap->nints : 0);
  if (np != (size_t)(2 * a->rank)) FAIL("Pad: pads must hold 2*rank entries");
  for (size_t i = 0; i < np; i++) {
    pads[i] = tp ? tp->i64[i] : ap->ints[i];
    if (pads[i] < 0) FAIL("Pad: negative pads");
  }
  float value = tv ? tv->f[0] : attr_f(nd, "value", 0.0f);
  int64_t od[MAXRANK];
  const int rank = a->rank;
  for (int i = 0; i < rank; i++) od[i] = a->dims[i] + pads[i] + pads[rank + i];
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, rank, od);
  a = get_in(x, nd, 0);
  for (size_t i = 0; i < o->n; i++) o->f[i] = value;
  int64_t idx[MAXRANK] = {0};
  for (size_t flat = 0; flat < a->n; flat++) {
    size_t dst = 0;
    for (int i = 0; i < rank; i++) dst = dst * (size_t)od[i] + (size_t)(idx[i] + pads[i]);
    o->f[dst] = a->f[flat];
    for (int i = rank - 1; i >= 0; i--) {
      if (++idx[i] < a->dims[i]) break;
      idx[i] = 0;
    }
  }
  return 0;
}

/* Sum: elementwise sum of equal-shaped inputs, left to right */
static int op_sum(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT) FAIL("Sum: bad input");
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, a->rank, a->dims);
  a = get_in(x, nd, 0);
  memcpy(o->f, a->f, a->n * 4);
  for (size_t k = 1; k < nd->nin; k++) {
    const Tensor *b = get_in(x, nd, k);
    if (!b || b->n != o->n) FAIL("Sum: inputs must have equal shapes");
    for (size_t i = 0; i < o->n; i++) o->f[i] = o->f[i] + b->f[i];
  }
  return 0;
}

/* LRN over the channel axis: y = x / (bias + alpha/size * sum_{c' in window} x[c']^2)^beta,
 * window = [c - floor((size-1)/2), c + ceil((size-1)/2)] clipped to the channels */
static int op_lrn(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT || a->rank < 3) FAIL("LRN: expects [N,C,...]");
  const int64_t size = attr_i(nd, "size", 0);
  if (size < 1) FAIL("LRN: size attribute required");
  const float alpha = attr_f(nd, "alpha", 1e-4f), beta = attr_f(nd, "beta", 0.75f), bias = attr_f(nd, "bias", 1.0f);
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, a->rank, a->dims);
  a = get_in(x, nd, 0);
  const size_t N = (size_t)a->dims[0], C = (size_t)a->dims[1], S = a->n / (N * C);
  const int64_t lo = (size - 1) / 2, hi = size - 1 - lo;
  for (size_t n = 0; n < N; n++)
    for (size_t c = 0; c < C; c++)
      for (size_t p = 0; p < S; p++) {
        int64_t c0 = (int64_t)c - lo, c1 = (int64_t)c + hi;
        if (c0 < 0) c0 = 0;
        if (c1 > (int64_t)C - 1) c1 = (int64_t)C - 1;
        float sq = 0.0f;
        for (int64_t k = c0; k <= c1; k++) {
          const float v = a->f[(n * C + (size_t)k) * S + p];
          sq = fmaf(v, v, sq);
        }
        o->f[(n * C + c) * S + p] = a->f[(n * C + c) * S + p] / powf(bias + alpha / (float)size * sq, beta);
      }
  return 0;
}

/* Transpose: out.dims[i] = in.dims[perm[i]] (default: reversed axes) */
static int op_transpose(Exec *x, const Node *nd) {
  const Tensor *a = get_in(x, nd, 0);
  if (!a || a->dtype != DT_FLOAT) FAIL("Transpose: bad input");
  const Attr *ap = find_attr(nd, "perm");
  const int rank = a->rank;
  int64_t perm[MAXRANK], od[MAXRANK], istr[MAXRANK];
  if (ap && ap->nints != (size_t)rank) FAIL("Transpose: perm length");
  for (int i = 0; i < rank; i++) perm[i] = ap ? ap->ints[i] : rank - 1 - i;
  for (int i = 0; i < rank; i++) {
    if (perm[i] < 0 || perm[i] >= rank) FAIL("Transpose: bad perm");
    od[i] = a->dims[perm[i]];
  }
  Tensor *o = env_new(&x->env, nd->out[0], DT_FLOAT, rank, od);
  a = get_in(x, nd, 0);
  int64_t st = 1;
  for (int i = rank - 1; i >= 0; i--) { istr[i] = st; st *= a->dims[i]; }
  int64_t idx[MAXRANK] = {0};
  for (size_t flat = 0; flat < o->n; flat++) {
    size_t src = 0;
    for (int i = 0; i < rank; i++) src += (size_t)(idx[i] * istr[perm[i]]);
    o->f[flat] = a->f[src];
    for (int i = rank - 1; i >= 0; i--) {
      if (++idx[i] < od[i]) break;
      idx[i] = 0;
    }
  }
  return 0;
}

static int op_constant(Exec *x, const Node *nd) {
  const Attr *a = find_attr(nd, "value");
  if (a && a->t) {
    Tensor *o = env_new(&x->env, nd->out[0], a->t->dtype, a->t->rank, a->t->dims);
    if (a->t->dtype == DT_FLOAT) memcpy(o->f, a->t->f, o->n * 4);
    else memcpy(o->i64, a->t->i64, o->n * 8);
    return 0;
  }
  int64_t one = 1;
  if ((a = find_attr(nd, "value_float"))) { env_new(&x->env, nd->out[0], DT_FLOAT, 0, &one)->f[0] = a->f; return 0; }
  if ((a = find_attr(nd, "value_int"))) { env_new(&x->env, nd->out[0], DT_INT64, 0, &one)->i64[0] = a->i; return 0; }
  if ((a = find_attr(nd, "value_ints"))) {
    int64_t cnt = (int64_t)a->nints;
    Tensor *o = env_new(&x->env, nd->out[0], DT_INT64, 1, &cnt);
    memcpy(o->i64, a->ints, (size_t)cnt * 8);
    return 0;
  }
  FAIL("Constant: only the value / value_float / value_int / value_ints forms are supported");
}

static int run_node(Exec *x, const Node *nd) {
  const char *op = nd->op;
  if (nd->nout < 1) FAIL("node %s has no output", op);
  if (!strcmp(op, "MatMul")) return op_matmul(x, nd);
  if (!strcmp(op, "Gemm")) return op_gemm(x, nd);
  if (!strcmp(op, "Add")) return op_binary(x, nd, '+');
  if (!strcmp(op, "Sub")) return op_binary(x, nd, '-');
  if (!strcmp(op, "Mul")) return op_binary(x, nd, '*');
  if (!strcmp(op, "Div")) return op_binary(x, nd, '/');
  if (!strcmp(op, "Min")) return op_binary(x, nd, 'm');
  if (!strcmp(op, "Max")) return op_binary(x, nd, 'M');
  if (!strcmp(op, "Pow")) return op_binary(x, nd, '^');
  if (!strcmp(op, "PRelu")) return op_binary(x, nd, 'p');
  {
    static const char *const unary_ops[] = {"Relu", "Sigmoid", "Tanh", "LeakyRelu", "Identity", "Dropout", "Clip", "Exp", "Log", "Sqrt",
                                            "Neg", "Abs", "Elu", "Selu", "Softplus", "HardSigmoid", "HardSwish", "Erf", "Gelu",
                                            "Reciprocal", "Floor", "Ceil", "Softsign", "Round", NULL};
    for (int i = 0; unary_ops[i]; i++)
      if (!strcmp(op, unary_ops[i])) return op_unary(x, nd);
  }
  if (!strcmp(op, "Shape")) return op_shape(x, nd);
  if (!strcmp(op, "Gather")) return op_gather(x, nd);
  if (!strcmp(op, "Slice")) return op_slice(x, nd);
  if (!strcmp(op, "Split")) return op_split(x, nd);
  if (!strcmp(op, "Cast")) return op_cast(x, nd);
  if (!strcmp(op, "Concat")) return op_concat(x, nd);
  if (!strcmp(op, "ReduceMean")) return op_reduce_mean(x, nd);
  if (!strcmp(op, "ArgMax")) return op_argmax(x, nd);
  if (!strcmp(op, "Softmax")) return op_softmax(x, nd, 0);
  if (!strcmp(op, "LogSoftmax")) return op_softmax(x, nd, 1);
  if (!strcmp(op, "Flatten") || !strcmp(op, "Reshape") || !strcmp(op, "Squeeze") || !strcmp(op, "Unsqueeze"))
    return op_reshape_like(x, nd);
  if (!strcmp(op, "Conv")) return op_conv(x, nd);
  if (!strcmp(op, "BatchNormalization")) return op_batchnorm(x, nd);
  if (!strcmp(op, "MaxPool")) return op_pool(x, nd, 1);
  if (!strcmp(op, "AveragePool")) return op_pool(x, nd, 0);
  if (!strcmp(op, "GlobalAveragePool")) return op_global_avgpool(x, nd);
  if (!strcmp(op, "GlobalMaxPool")) return op_global_maxpool(x, nd);
  if (!strcmp(op, "Pad")) return op_pad(x, nd);
  if (!strcmp(op, "Sum")) return op_sum(x, nd);
  if (!strcmp(op, "LRN")) return op_lrn(x, nd);
  if (!strcmp(op, "Transpose")) return op_transpose(x, nd);
  if (!strcmp(op, "Constant")) return op_constant(x, nd);
  if (!strcmp(op, "Scaler")) return op_ml_scaler(x, nd);
  if (!strcmp(op, "LinearRegressor")) return op_ml_linear_regressor(x, nd);
  if (!strcmp(op, "LinearClassifier")) return op_ml_linear_classifier(x, nd);
  if (!strcmp(op, "Normalizer")) return op_ml_normalizer(x, nd);
  if (!strcmp(op, "ArrayFeatureExtractor")) return op_ml_array_feature_extractor(x, nd);
  FAIL("unsupported operator: %s", op);
}

/* Runs the graph on one f32 input; returns the first output (engine.rs:146-149). */
static int run_graph(const OrcModel *m, const float *data, int rank, const int64_t *dims, Tensor *result,
                     char *err, size_t errlen) {
  Exec xs = {m, {0, 0, 0}, err, errlen};
  Exec *x = &xs;
  if (m->ninputs < 1) { set_err(err, errlen, "model has no input"); return -1; }
  const ValueInfo *vi = &m->inputs[0];
  if (m->ninputs > 1) {
    /* Build extension mirrored from the product (the reference feeds input 0 only, engine.rs:139-145): several f32
     * [rows, k_i] inputs take consecutive column ranges of the fed [rows, sum k_i] matrix, in declaration order. */
    if (rank != 2) { set_err(err, errlen, "input rank mismatch: multi-input models take a rank-2 feature matrix"); return -1; }
    int64_t total = 0;
    for (size_t k = 0; k < m->ninputs; k++) {
      const ValueInfo *v = &m->inputs[k];
      if (v->rank != 2 || v->dims[1] <= 0 || (v->has_type && v->elem_type != DT_FLOAT)) { set_err(err, errlen, "multi-input models need f32 [rows, k] inputs"); return -1; }
      if (v->dims[0] >= 0 && v->dims[0] != dims[0]) { set_err(err, errlen, "input shape mismatch at axis 0: model expects %lld, got %lld", (long long)v->dims[0], (long long)dims[0]); return -1; }
      total += v->dims[1];
    }
    if (total != dims[1]) { set_err(err, errlen, "input shape mismatch at axis 1: model expects %lld, got %lld", (long long)total, (long long)dims[1]); return -1; }
    int64_t off = 0;
    for (size_t k = 0; k < m->ninputs; k++) {
      int64_t d2[2] = {dims[0], m->inputs[k].dims[1]};
      Tensor *t = env_new(&x->env, m->inputs[k].name, DT_FLOAT, 2, d2);
      for (int64_t r = 0; r < dims[0]; r++) memcpy(t->f + r * d2[1], data + r * dims[1] + off, (size_t)d2[1] * 4);
      off += d2[1];
    }
  } else {
  /* The backend checks the fed tensor against the declared input fact (rank + fixed dims). */
  if (vi->rank >= 0) {
    if (vi->rank != rank) { set_err(err, errlen, "input rank mismatch: model expects rank %d, got rank %d", vi->rank, rank); return -1; }
    for (int i = 0; i < rank; i++)
      if (vi->dims[i] >= 0 && vi->dims[i] != dims[i]) {
        set_err(err, errlen, "input shape mismatch at axis %d: model expects %lld, got %lld", i, (long long)vi->dims[i], (long long)dims[i]);
        return -1;
      }
  }
  Tensor *in = env_new(&x->env, vi->name, DT_FLOAT, rank, dims);
  memcpy(in->f, data, in->n * 4);
  }
  /* only nodes that feed the FIRST output run (engine.rs:146-149 reads outputs[0]); liveness by a backward sweep
   * over the topologically sorted node list */
  char *live = (char *)xcalloc(m->nnodes ? m->nnodes : 1, 1);
  {
    size_t nneed = 1, cap = 64;
    const char **need = (const char **)xmalloc(cap * sizeof(char *));
    need[0] = m->noutputs ? m->outputs[0].name : "";
    for (size_t i = m->nnodes; i-- > 0;) {
      const Node *nd = &m->nodes[i];
      int hit = 0;
      for (size_t o = 0; o < nd->nout && !hit; o++)
        for (size_t k = 0; k < nneed && !hit; k++) hit = strcmp(nd->out[o], need[k]) == 0;
      if (!hit) continue;
      live[i] = 1;
      for (size_t k = 0; k < nd->nin; k++) {
        if (nneed == cap) { cap *= 2; need = (const char **)realloc(need, cap * sizeof(char *)); }
        need[nneed++] = nd->in[k];
      }
    }
    free(need);
  }
  for (size_t i = 0; i < m->nnodes; i++)
    if (live[i] && run_node(x, &m->nodes[i])) { free(live); env_free(&x->env); return -1; }
  free(live);
  if (m->noutputs < 1) { set_err(err, errlen, "No output tensor"); env_free(&x->env); return -1; }
  long oi = env_find(&x->env, m->outputs[0].name);
  const Tensor *ot = oi >= 0 ? &x->env.v[oi] : find_init(m, m->outputs[0].name);
  if (!ot) { set_err(err, errlen, "output '%s' was never produced", m->outputs[0].name); env_free(&x->env); return -1; }
  /* The reference rejects non-f32 outputs (engine.rs:150-152).  The build's documented extension:
   * integer outputs (ArgMax labels, Cast to int) come back as f32 VALUES, since the C ABI carries f32 only. */
  memset(result, 0, sizeof *result);
  result->dtype = DT_FLOAT;
  result->rank = ot->rank;
  memcpy(result->dims, ot->dims, sizeof(ot->dims));
  result->n = ot->n;
  result->f = (float *)xmalloc(ot->n * 4);
  if (ot->dtype == DT_FLOAT) memcpy(result->f, ot->f, ot->n * 4);
  else for (size_t i = 0; i < ot->n; i++) result->f[i] = (float)ot->i64[i];
  env_free(&x->env);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* 3. engine mirror                                                                             */
/* ------------------------------------------------------------------------------------------ */

static void free_node(Node *nd) {
  free(nd->op);
  free(nd->name);
  for (size_t i = 0; i < nd->nin; i++) free(nd->in[i]);
  for (size_t i = 0; i < nd->nout; i++) free(nd->out[i]);
  free(nd->in);
  free(nd->out);
  for (size_t i = 0; i < nd->nattrs; i++) {
    Attr *a = &nd->attrs[i];
    free(a->name);
    free(a->s);
    free(a->ints);
    free(a->floats);
    if (a->t) { tensor_free_payload(a->t); free(a->t); }
  }
  free(nd->attrs);
}

void orc_free_model(OrcModel *m) {
  if (!m) return;
  for (size_t i = 0; i < m->nnodes; i++) free_node(&m->nodes[i]);
  free(m->nodes);
  for (size_t i = 0; i < m->ninits; i++) tensor_free_payload(&m->inits[i]);
  free(m->inits);
  for (size_t i = 0; i < m->ninputs; i++) free(m->inputs[i].name);
  free(m->inputs);
  for (size_t i = 0; i < m->noutputs; i++) free(m->outputs[i].name);
  free(m->outputs);
  free(m);
}

/* Output fact when graph.output carries no (or partial) shape: run the graph with every symbolic
 * input dim set to 1 and then to 2; axes whose extent changes are symbolic (-1). */
static int infer_output_shape(OrcModel *m, char *err, size_t errlen) {
  const ValueInfo *vi = &m->inputs[0];
  Tensor r[2];
  for (int pass = 0; pass < 2; pass++) {
    int64_t d[MAXRANK];
    for (int i = 0; i < vi->rank; i++) d[i] = m->in_shape[i] >= 0 ? m->in_shape[i] : pass + 1;
    size_t n = dims_count(d, vi->rank);
    float *z = (float *)xcalloc(n, 4);
    int rc = run_graph(m, z, vi->rank, d, &r[pass], err, errlen);
    free(z);
    if (rc) {
      if (pass) free(r[0].f);
      return -1;
    }
  }
  m->out_rank = r[0].rank;
  for (int i = 0; i < r[0].rank; i++) m->out_shape[i] = r[0].dims[i] == r[1].dims[i] ? r[0].dims[i] : -1;
  free(r[0].f);
  free(r[1].f);
  return 0;
}

OrcModel *orc_load(const char *path, char *err, size_t errlen) {
  FILE *fp = fopen(path, "rb");
  if (!fp) { set_err(err, errlen, "ONNX error: cannot open %s", path); return NULL; }
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)xmalloc((size_t)sz);
  if (fread(buf, 1, (size_t)sz, fp) != (size_t)sz) { fclose(fp); free(buf); set_err(err, errlen, "ONNX error: short read"); return NULL; }
  fclose(fp);
  OrcModel *m = (OrcModel *)xcalloc(1, sizeof *m);
  char e2[256] = {0};
  Pb b = {buf, buf + sz, 0};
  int wt, saw_graph = 0;
  uint32_t f;
  while ((f = pb_tag(&b, &wt))) {
    if (f == 1 && wt == 0) m->ir_version = (int64_t)pb_varint(&b);
    else if (f == 7 && wt == 2) {
      Pb g = pb_sub(&b);
      if (b.bad) break;
      if (parse_graph(g, m, e2, sizeof e2)) goto fail;
      saw_graph = 1;
    } else if (f == 8 && wt == 2) { /* opset_import {domain=1, version=2} */
      Pb s = pb_sub(&b);
      int wt2, dflt = 1;
      uint32_t f2;
      int64_t ver = 0;
      while ((f2 = pb_tag(&s, &wt2))) {
        if (f2 == 1 && wt2 == 2) { Pb d = pb_sub(&s); if (d.end - d.p > 0 && !(d.end - d.p == 7 && !memcmp(d.p, "ai.onnx", 7))) dflt = 0; }
        else if (f2 == 2 && wt2 == 0) ver = (int64_t)pb_varint(&s);
        else pb_skip(&s, wt2);
      }
      if (dflt) m->opset = ver;
    } else pb_skip(&b, wt);
  }
  if (b.bad || !saw_graph) { snprintf(e2, sizeof e2, "not a valid ONNX ModelProto"); goto fail; }
  if (m->opset == 0) m->opset = 1;
  if (m->ninputs < 1 || m->noutputs < 1) { snprintf(e2, sizeof e2, "model needs at least one input and one output"); goto fail; }
  if (m->inputs[0].rank < 0) { snprintf(e2, sizeof e2, "input 0 has no shape"); goto fail; }
  m->in_rank = m->inputs[0].rank;
  for (int i = 0; i < m->in_rank; i++) m->in_shape[i] = m->inputs[0].dims[i] >= 0 ? m->inputs[0].dims[i] : -1;
  if (m->ninputs > 1 && m->in_rank == 2) { /* the call's feature matrix spans all inputs (see run_graph) */
    int64_t total = 0;
    for (size_t k = 0; k < m->ninputs; k++) total += m->inputs[k].rank == 2 && m->inputs[k].dims[1] > 0 ? m->inputs[k].dims[1] : 0;
    m->in_shape[1] = total;
  }
  if (infer_output_shape(m, e2, sizeof e2)) goto fail;
  free(buf);
  return m;
fail:
  set_err(err, errlen, "ONNX error: %s", e2);
  free(buf);
  orc_free_model(m);
  return NULL;
}

int orc_input_rank(const OrcModel *m) { return m->in_rank; }
int orc_output_rank(const OrcModel *m) { return m->out_rank; }
const int64_t *orc_input_shape(const OrcModel *m) { return m->in_shape; }
const int64_t *orc_output_shape(const OrcModel *m) { return m->out_shape; }

/* engine.rs:19-29 */
void orc_shape_rows_cols(const size_t *shape, int rank, size_t *rows, size_t *cols) {
  if (rank == 0) { *rows = 1; *cols = 1; return; }
  if (rank == 1) { *rows = shape[0]; *cols = 1; return; }
  size_t c = 1;
  for (int i = 1; i < rank; i++) c *= shape[i];
  *rows = shape[0];
  *cols = c < 1 ? 1 : c;
}

static void fill_result(const Tensor *t, OrcResult *out) {
  size_t shp[MAXRANK];
  for (int i = 0; i < t->rank; i++) shp[i] = (size_t)t->dims[i];
  orc_shape_rows_cols(shp, t->rank, &out->rows, &out->cols); /* engine.rs:153 */
  out->data = t->f;                                           /* engine.rs:154-156 */
  out->len = t->n;
}

/* Rust `{:?}` of &[i64], e.g. "[3]" or "[3, 224, 224]" (engine.rs:132) */
static void fmt_i64_debug(const int64_t *v, int n, char *buf, size_t len) {
  size_t o = 0;
  o += (size_t)snprintf(buf + o, len - o, "[");
  for (int i = 0; i < n && o < len; i++) o += (size_t)snprintf(buf + o, len - o, "%s%lld", i ? ", " : "", (long long)v[i]);
  if (o < len) snprintf(buf + o, len - o, "]");
}

int orc_predict(const OrcModel *m, const float *data, size_t rows, size_t cols, OrcResult *out, char *err,
                size_t errlen) {
  memset(out, 0, sizeof *out);
  /* engine.rs:126-137: if all inner dims known, cols must equal their product */
  if (m->in_rank > 0) {
    int all = 1;
    size_t expect = 1;
    for (int i = 1; i < m->in_rank; i++) {
      if (m->in_shape[i] <= 0) all = 0;
      else expect *= (size_t)m->in_shape[i];
    }
    if (all && cols != expect) {
      char dbg[128];
      fmt_i64_debug(m->in_shape + 1, m->in_rank - 1, dbg, sizeof dbg);
      set_err(err, errlen, "Invalid input shape: expected batch x %s, got %zu x %zu", dbg, rows, cols);
      return -1;
    }
  }
  /* engine.rs:139-145: always a rank-2 [rows, cols] tensor */
  int64_t d[2] = {(int64_t)rows, (int64_t)cols};
  Tensor t;
  char e2[256] = {0};
  if (run_graph(m, data, 2, d, &t, e2, sizeof e2)) { set_err(err, errlen, "ONNX error: %s", e2); return -1; }
  fill_result(&t, out);
  return 0;
}

int orc_predict_blob(const OrcModel *m, const uint8_t *blob, size_t len, OrcResult *out, char *err, size_t errlen) {
  memset(out, 0, sizeof *out);
  if (len % 4 != 0) { set_err(err, errlen, "Invalid BLOB size: length must be a multiple of 4"); return -1; } /* :209-211 */
  size_t n = len / 4;
  size_t expected = 1; /* :221-226: product of dims > 0 (empty product = 1) */
  for (int i = 0; i < m->in_rank; i++)
    if (m->in_shape[i] > 0) expected *= (size_t)m->in_shape[i];
  if (expected == 0 || n % expected != 0) { /* :227-232 */
    set_err(err, errlen, "BLOB data does not match model's expected input shape. Expected %zu elements, but BLOB contained %zu.", expected, n);
    return -1;
  }
  size_t batch = n / expected; /* :233 */
  int64_t d[MAXRANK];
  size_t prod = 1;
  for (int i = 0; i < m->in_rank; i++) { /* :234-238 every -1 -> batch */
    d[i] = m->in_shape[i] == -1 ? (int64_t)batch : m->in_shape[i];
    prod *= (size_t)d[i];
  }
  if (prod != n) { /* Tensor::from_shape fails (:239-240) */
    set_err(err, errlen, "ONNX error: shape/data length mismatch: shape holds %zu elements, data holds %zu", prod, n);
    return -1;
  }
  float *f = (float *)xmalloc(len);
  memcpy(f, blob, len); /* native-endian bytes -> f32 (:212-220) */
  Tensor t;
  char e2[256] = {0};
  int rc = run_graph(m, f, m->in_rank, d, &t, e2, sizeof e2);
  free(f);
  if (rc) { set_err(err, errlen, "ONNX error: %s", e2); return -1; }
  fill_result(&t, out);
  return 0;
}

void orc_free_result(OrcResult *r) {
  if (r && r->data) free(r->data);
  if (r) memset(r, 0, sizeof *r);
}

/* ------------------------------------------------------------------------------------------ */
/* 4. synthetic table + CPU scan baseline                                                       */
/* ------------------------------------------------------------------------------------------ */

uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

float orc_synth_value(uint64_t seed, uint64_t row, uint64_t col, uint64_t ncols) {
  uint64_t u = orc_splitmix64(seed ^ (row * ncols + col));
  return (float)(u >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
}

void orc_synth_fill_rowmajor(float *dst, uint64_t seed, uint64_t row0, uint64_t rows, uint64_t ncols) {
  for (uint64_t r = 0; r < rows; r++)
    for (uint64_t c = 0; c < ncols; c++) dst[r * ncols + c] = orc_synth_value(seed, row0 + r, c, ncols);
}

/* A tagged value standing in for duckdb::Value in the "boxed" gather (cost class only). */
typedef struct {
  int type_id;
  int is_null;
  union { float f; double d; int32_t i32; int64_t i64; } v;
} BoxedValue;

/* Returned by value through a non-inlined call: the cost class of Vector::GetValue for a FLOAT
 * cell (a tagged Value is materialised per cell; numeric Values do not heap-allocate). */
static __attribute__((noinline)) BoxedValue box_get_value(const float *column, size_t row) {
  BoxedValue b;
  b.type_id = 1;
  b.is_null = 0;
  b.v.f = column[row];
  return b;
}

typedef struct {
  const OrcModel *m;
  uint64_t rows, ncols, seed;
  int chunk_rows, boxed;
  uint64_t *next_chunk; /* shared atomic counter */
  double checksum;
  int failed;
} ScanArg;

static void *scan_worker(void *p) {
  ScanArg *a = (ScanArg *)p;
  size_t F = (size_t)a->ncols, CH = (size_t)a->chunk_rows;
  float *cols = (float *)xmalloc(F * CH * 4);  /* columnar chunk: F vectors of CH floats */
  float *feat = (float *)xmalloc(F * CH * 4);  /* row-major gather target */
  float *resv = (float *)xmalloc(CH * 64 * 4); /* result vector */
  uint64_t nchunks = (a->rows + CH - 1) / CH;
  t_arena.cap = (size_t)256 << 20;
  t_arena.base = (char *)xmalloc(t_arena.cap);
  t_arena.active = 1;
  for (;;) {
    uint64_t c = __atomic_fetch_add(a->next_chunk, 1, __ATOMIC_RELAXED);
    if (c >= nchunks) break;
    uint64_t r0 = c * CH;
    size_t nr = (size_t)((a->rows - r0) < CH ? (a->rows - r0) : CH);
    /* table generation (outside what the reference would time, but identical for every backend) */
    for (size_t j = 0; j < F; j++)
      for (size_t r = 0; r < nr; r++) cols[j * CH + r] = orc_synth_value(a->seed, r0 + r, j, F);
    /* ExtractFeatures, infera_extension.cpp:199-227: row-outer, col-inner  (boxed == 2: plain gather + the blocked GEMM, "best CPU") */
    t_gemm_blocked = a->boxed == 2;
    if (a->boxed == 1) {
      size_t k = 0;
      for (size_t r = 0; r < nr; r++)
        for (size_t j = 0; j < F; j++) {
          BoxedValue b = box_get_value(cols + j * CH, r);
          if (b.is_null) { a->failed = 1; }
          float v;
          switch (b.type_id) {
            case 1: v = b.v.f; break;
            case 2: v = (float)b.v.d; break;
            case 3: v = (float)b.v.i32; break;
            default: v = (float)b.v.i64; break;
          }
          feat[k++] = v;
        }
    } else {
      for (size_t r = 0; r < nr; r++)
        for (size_t j = 0; j < F; j++) feat[r * F + j] = cols[j * CH + r];
    }
    OrcResult res;
    char err[256];
    t_arena.off = 0;
    /* models with an [N,C,H,W] input take the BLOB route, as in the reference (infera_predict_from_blob -> engine.rs:199-263) */
    if (a->m->in_rank > 2 ? orc_predict_blob(a->m, (const uint8_t *)feat, nr * F * 4, &res, err, sizeof err)
                          : orc_predict(a->m, feat, nr, F, &res, err, sizeof err)) { a->failed = 1; break; }
    size_t take = res.len < CH * 64 ? res.len : CH * 64;
    for (size_t i = 0; i < take; i++) { resv[i] = res.data[i]; a->checksum += (double)res.data[i]; }
    orc_free_result(&res);
  }
  t_arena.active = 0;
  free(t_arena.base);
  free(cols);
  free(feat);
  free(resv);
  return NULL;
}

double orc_bench_scan(const OrcModel *m, uint64_t rows, uint64_t ncols, uint64_t seed, int threads, int chunk_rows,
                      int boxed, double *checksum) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  uint64_t next = 0;
  pthread_t th[256];
  ScanArg args[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < threads; i++) {
    args[i] = (ScanArg){m, rows, ncols, seed, chunk_rows, boxed, &next, 0.0, 0};
    pthread_create(&th[i], NULL, scan_worker, &args[i]);
  }
  double cs = 0.0;
  int failed = 0;
  for (int i = 0; i < threads; i++) {
    pthread_join(th[i], NULL);
    cs += args[i].checksum;
    failed |= args[i].failed;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (checksum) *checksum = cs;
  if (failed) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- the same scan over a MATERIALISED columnar table (bench.py's cpu_baseline leg) -----------
 * The table is generated once by the caller, OUTSIDE every timed region (8d: "excludes ... data generation"),
 * in DuckDB's storage shape: row groups of ORC_ROW_GROUP rows, one contiguous run per column inside a group.
 * Each worker takes 2048-row chunks from a shared counter, gathers them the way ExtractFeatures does
 * (infera_extension.cpp:199-227: row-outer, column-inner, one boxed value per cell) or with a plain strided
 * copy (boxed = 0, the best a CPU gather can do), and runs the graph single-threaded on the chunk
 * (engine.rs:139-154: one Tract run per call; the reference's parallelism is DuckDB's worker threads). */
#define ORC_ROW_GROUP 122880

typedef struct {
  const OrcModel *m;
  const float *table;
  uint64_t rows, ncols;
  int chunk_rows, boxed;
  uint64_t *next_chunk;
  double checksum;
  int failed;
} TableScanArg;

static void *table_scan_worker(void *p) {
  TableScanArg *a = (TableScanArg *)p;
  size_t F = (size_t)a->ncols, CH = (size_t)a->chunk_rows;
  float *feat = (float *)xmalloc(F * CH * 4);  /* row-major gather target */
  float *resv = (float *)xmalloc(CH * 64 * 4); /* result vector */
  uint64_t nchunks = (a->rows + CH - 1) / CH;
  t_arena.cap = (size_t)256 << 20;
  t_arena.base = (char *)xmalloc(t_arena.cap);
  t_arena.active = 1;
  for (;;) {
    uint64_t c = __atomic_fetch_add(a->next_chunk, 1, __ATOMIC_RELAXED);
    if (c >= nchunks) break;
    uint64_t r0 = c * CH, g0 = r0 / ORC_ROW_GROUP * ORC_ROW_GROUP;
    uint64_t gr = a->rows - g0 < ORC_ROW_GROUP ? a->rows - g0 : ORC_ROW_GROUP;
    size_t nr = (size_t)((a->rows - r0) < CH ? (a->rows - r0) : CH);
    const float *base = a->table + g0 * F + (r0 - g0); /* column j of this chunk: base + j*gr */
    if (a->boxed == 1) {
      size_t k = 0;
      for (size_t r = 0; r < nr; r++)
        for (size_t j = 0; j < F; j++) {
          BoxedValue b = box_get_value(base + j * gr, r);
          if (b.is_null) { a->failed = 1; }
          float v;
          switch (b.type_id) {
            case 1: v = b.v.f; break;
            case 2: v = (float)b.v.d; break;
            case 3: v = (float)b.v.i32; break;
            default: v = (float)b.v.i64; break;
          }
          feat[k++] = v;
        }
    } else if (a->boxed == 2) {
      /* best CPU: 8-row x 8-column tiles, so the column runs are read as 32-byte pieces and the row-major target is written as
       * 32-byte pieces (the plain strided copy below touches one float per 512-byte stride) */
      size_t r = 0;
      for (; r + 8 <= nr; r += 8) {
        size_t j = 0;
        for (; j + 8 <= F; j += 8)
          for (size_t jj = 0; jj < 8; jj++) {
            const float *src = base + (j + jj) * gr + r;
            for (size_t rr = 0; rr < 8; rr++) feat[(r + rr) * F + j + jj] = src[rr];
          }
        for (; j < F; j++)
          for (size_t rr = 0; rr < 8; rr++) feat[(r + rr) * F + j] = base[j * gr + r + rr];
      }
      for (; r < nr; r++)
        for (size_t j = 0; j < F; j++) feat[r * F + j] = base[j * gr + r];
    } else {
      for (size_t r = 0; r < nr; r++)
        for (size_t j = 0; j < F; j++) feat[r * F + j] = base[j * gr + r];
    }
    OrcResult res;
    char err[256];
    t_arena.off = 0;
    t_gemm_blocked = a->boxed == 2;
    /* models with an [N,C,H,W] input take the BLOB route, as in the reference (infera_predict_from_blob -> engine.rs:199-263) */
    if (a->m->in_rank > 2 ? orc_predict_blob(a->m, (const uint8_t *)feat, nr * F * 4, &res, err, sizeof err)
                          : orc_predict(a->m, feat, nr, F, &res, err, sizeof err)) { a->failed = 1; break; }
    size_t take = res.len < CH * 64 ? res.len : CH * 64;
    for (size_t i = 0; i < take; i++) { resv[i] = res.data[i]; a->checksum += (double)res.data[i]; }
    orc_free_result(&res);
  }
  t_arena.active = 0;
  free(t_arena.base);
  free(feat);
  free(resv);
  return NULL;
}

double orc_bench_scan_table(const OrcModel *m, const float *table, uint64_t rows, uint64_t ncols, int threads, int chunk_rows,
                            int boxed, double *checksum) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (chunk_rows < 1 || ORC_ROW_GROUP % chunk_rows) return -1.0;
  uint64_t next = 0;
  pthread_t th[256];
  TableScanArg args[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < threads; i++) {
    args[i] = (TableScanArg){m, table, rows, ncols, chunk_rows, boxed, &next, 0.0, 0};
    pthread_create(&th[i], NULL, table_scan_worker, &args[i]);
  }
  double cs = 0.0;
  int failed = 0;
  for (int i = 0; i < threads; i++) {
    pthread_join(th[i], NULL);
    cs += args[i].checksum;
    failed |= args[i].failed;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (checksum) *checksum = cs;
  if (failed) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
