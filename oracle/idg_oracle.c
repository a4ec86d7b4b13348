/*
 * oracle/idg_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the arithmetic the ID-GRec reference performs on
 * the LightGCN hot path.  It exists only to check libidgrec.so's HIP kernels (tests/,
 * __graft_entry__.smoke(), and bench.py's cpu_baseline leg).  Nothing under id-grec_amd/,
 * models/ or utility/ may link, load or call it.
 *
 * Pinned by tests/golden/ (vectors dumped from the imported reference by
 * oracle/gen_golden.py): orc_spmm_f32 reproduces torch 2.10 CPU torch.sparse.mm bit for
 * bit, orc_propagate_mean_f32 reproduces LightGCN.aggregate / SimGCL.aggregate bit for
 * bit, the loss/gradient/Adam functions agree with torch autograd to fp32 rounding.
 *
 * Each function cites the reference lines it follows (paths relative to the reference repo).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* torch.sparse.mm(Graph, X) on CPU (models/LightGCN.py:44): for each row, for each stored
 * entry in CSR order, acc = fmaf(val, X[col][f], acc), starting from +0. */
void orc_spmm_f32(int64_t n_rows, const int64_t* indptr, const int32_t* indices, const float* values,
                  const float* X, int64_t d, float* Y) {
  for (int64_t r = 0; r < n_rows; ++r) {
    float* y = Y + r * d;
    for (int64_t f = 0; f < d; ++f) y[f] = 0.0f;
    for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
      const float v = values[k];
      const float* x = X + (int64_t)indices[k] * d;
      for (int64_t f = 0; f < d; ++f) y[f] = fmaf(v, x[f], y[f]);
    }
  }
}

/* Same product under libidgrec's published split schedule (include/idgrec.h,
 * idg_graph_long_rows): the rows listed in long_rows (ascending) are cut into consecutive
 * segments of seg_len[i] entries, each a fmaf chain from +0; the segment partials are combined
 * 4-way strided: s_q = p_q + p_{q+4} + ... for q = 0..3, row = ((s_0 + s_1) + s_2) + s_3.
 * With chunk_len[i] = C > 0 the row is first cut into chunks of C entries, the rule above gives each
 * chunk's sum, and the chunk sums are combined by the same 4-way rule.
 * All other rows are the plain sequential chain.  This restates the summation ORDER only;
 * the operands are the reference's. */
enum { ORC_WAYS = 4 };

/* SEG(entries [s,e), S) of include/idgrec.h: segments of S entries, each a sequential fmaf chain
 * from +0, combined 4-way strided.  part: d floats, way: 4*d floats of scratch; result in y. */
static void orc_seg_sum(const int32_t* indices, const float* values, const float* X, int64_t d, int64_t s, int64_t e,
                        int64_t S, float* part, float* way, float* y) {
  int64_t nseg = 0;
  for (int64_t b = s; b < e; b += S, ++nseg) {
    const int64_t be = b + S < e ? b + S : e;
    for (int64_t f = 0; f < d; ++f) part[f] = 0.0f;
    for (int64_t k = b; k < be; ++k) {
      const float v = values[k];
      const float* x = X + (int64_t)indices[k] * d;
      for (int64_t f = 0; f < d; ++f) part[f] = fmaf(v, x[f], part[f]);
    }
    float* w = way + (nseg % ORC_WAYS) * d;
    if (nseg < ORC_WAYS)
      for (int64_t f = 0; f < d; ++f) w[f] = part[f];
    else
      for (int64_t f = 0; f < d; ++f) w[f] = w[f] + part[f];
  }
  for (int64_t f = 0; f < d; ++f) y[f] = nseg > 0 ? way[f] : 0.0f;
  for (int64_t q = 1; q < ORC_WAYS && q < nseg; ++q)
    for (int64_t f = 0; f < d; ++f) y[f] = y[f] + way[q * d + f];
}

void orc_spmm_sched_f32(int64_t n_rows, const int64_t* indptr, const int32_t* indices, const float* values,
                        const float* X, int64_t d, const int64_t* long_rows, const int64_t* seg_len,
                        const int64_t* chunk_len, int64_t n_long, float* Y) {
  int64_t li = 0;
  float* part = (float*)malloc((size_t)d * sizeof(float));
  float* way = (float*)malloc((size_t)d * ORC_WAYS * sizeof(float));
  float* csum = (float*)malloc((size_t)d * sizeof(float));
  float* cway = (float*)malloc((size_t)d * ORC_WAYS * sizeof(float));
  for (int64_t r = 0; r < n_rows; ++r) {
    float* y = Y + r * d;
    const int64_t s = indptr[r], e = indptr[r + 1];
    if (li < n_long && long_rows[li] == r) {
      const int64_t S = seg_len[li], Cn = chunk_len ? chunk_len[li] : 0;
      ++li;
      if (Cn <= 0) {
        orc_seg_sum(indices, values, X, d, s, e, S, part, way, y);
        continue;
      }
      /* chunk sums c_k = SEG(chunk k, S), then the same 4-way strided rule over the c_k */
      int64_t nch = 0;
      for (int64_t b = s; b < e; b += Cn, ++nch) {
        orc_seg_sum(indices, values, X, d, b, b + Cn < e ? b + Cn : e, S, part, way, csum);
        float* w = cway + (nch % ORC_WAYS) * d;
        if (nch < ORC_WAYS)
          for (int64_t f = 0; f < d; ++f) w[f] = csum[f];
        else
          for (int64_t f = 0; f < d; ++f) w[f] = w[f] + csum[f];
      }
      for (int64_t f = 0; f < d; ++f) y[f] = nch > 0 ? cway[f] : 0.0f;
      for (int64_t q = 1; q < ORC_WAYS && q < nch; ++q)
        for (int64_t f = 0; f < d; ++f) y[f] = y[f] + cway[q * d + f];
    } else {
      for (int64_t f = 0; f < d; ++f) y[f] = 0.0f;
      for (int64_t k = s; k < e; ++k) {
        const float v = values[k];
        const float* x = X + (int64_t)indices[k] * d;
        for (int64_t f = 0; f < d; ++f) y[f] = fmaf(v, x[f], y[f]);
      }
    }
  }
  free(part);
  free(way);
  free(csum);
  free(cway);
}

/* LightGCN.aggregate (models/LightGCN.py:36-52) with include_layer0 = 1;
 * SimGCL.aggregate(perturbed=False) (models/SimGCL.py:39-60) with include_layer0 = 0.
 * torch.mean(torch.stack(layers, dim=1), dim=1) on CPU == left-to-right running sum then a
 * true division by the layer count.  n_long > 0 switches every product to the split
 * schedule.  tmp: 2*n*d floats. */
void orc_propagate_mean_f32(int64_t n, const int64_t* indptr, const int32_t* indices, const float* values,
                            const float* E0, int64_t d, int K, int include_layer0, const int64_t* long_rows,
                            const int64_t* seg_len, const int64_t* chunk_len, int64_t n_long, float* out, float* tmp) {
  const int64_t nd = n * d;
  float* P[2] = {tmp, tmp + nd};
  const float* X = E0;
  int have_sum = 0;
  if (include_layer0) {
    memcpy(out, E0, (size_t)nd * sizeof(float));
    have_sum = 1;
  }
  for (int k = 1; k <= K; ++k) {
    float* Y = P[(k - 1) & 1];
    if (n_long > 0)
      orc_spmm_sched_f32(n, indptr, indices, values, X, d, long_rows, seg_len, chunk_len, n_long, Y);
    else
      orc_spmm_f32(n, indptr, indices, values, X, d, Y);
    if (!have_sum) {
      memcpy(out, Y, (size_t)nd * sizeof(float));
      have_sum = 1;
    } else {
      for (int64_t i = 0; i < nd; ++i) out[i] = out[i] + Y[i];
    }
    X = Y;
  }
  const float cnt = (float)(K + (include_layer0 ? 1 : 0));
  if (cnt != 1.0f)
    for (int64_t i = 0; i < nd; ++i) out[i] = out[i] / cnt;
}

/* Autograd through the above for a symmetric graph (A^T = A):
 * t = g / cnt; gX_K = t; gX_k = t + A.gX_{k+1}; gE0 = [t] + A.gX_1. */
void orc_propagate_mean_bwd_f32(int64_t n, const int64_t* indptr, const int32_t* indices, const float* values,
                                const float* g, int64_t d, int K, int include_layer0, float* gE0, float* tmp) {
  const int64_t nd = n * d;
  float* t = tmp;
  float* h = tmp + nd;
  float* y = tmp + 2 * nd; /* tmp: 3*n*d floats */
  const float cnt = (float)(K + (include_layer0 ? 1 : 0));
  for (int64_t i = 0; i < nd; ++i) t[i] = g[i] / cnt;
  memcpy(h, t, (size_t)nd * sizeof(float));
  for (int k = K; k >= 1; --k) {
    orc_spmm_f32(n, indptr, indices, values, h, d, y);
    if (k > 1 || include_layer0)
      for (int64_t i = 0; i < nd; ++i) h[i] = t[i] + y[i];
    else
      memcpy(h, y, (size_t)nd * sizeof(float));
  }
  memcpy(gE0, h, (size_t)nd * sizeof(float));
}

/* LightGCN.forward (models/LightGCN.py:54-72) given the propagated panel `fin` and the ego
 * panel `ego` ([n,d], users first): get_bpr_loss (utility/utility_function/losses.py:4-13),
 * reg_lambda * get_reg_loss (:16-21), and their gradients, contributions added in batch
 * order.  g_final / g_ego are accumulated into (caller zeroes).  Either may be NULL. */
void orc_bpr_f32(const float* fin, const float* ego, int64_t num_users, const int64_t* users, const int64_t* pos,
                 const int64_t* neg, int64_t B, int64_t d, float reg_lambda, float* loss, float* g_final,
                 float* g_ego) {
  double lsum = 0.0;
  double sq[3] = {0.0, 0.0, 0.0};
  for (int64_t i = 0; i < B; ++i) {
    const int64_t ru = users[i], rp = num_users + pos[i], rn = num_users + neg[i];
    const float *fu = fin + ru * d, *fp = fin + rp * d, *fn = fin + rn * d;
    const float *eu = ego + ru * d, *ep = ego + rp * d, *en = ego + rn * d;
    float sp = 0.f, sn = 0.f;
    for (int64_t f = 0; f < d; ++f) {
      sp += fu[f] * fp[f];
      sn += fu[f] * fn[f];
      sq[0] += (double)eu[f] * eu[f];
      sq[1] += (double)ep[f] * ep[f];
      sq[2] += (double)en[f] * en[f];
    }
    const float x = sp - sn;
    const float sig = 1.0f / (1.0f + expf(-x));
    lsum += (double)(-logf(sig + 1e-7f)); /* losses.py:11: 10e-8 */
    const float c = -(sig * (1.0f - sig)) / (sig + 1e-7f) / (float)B;
    if (g_final) {
      float *gu = g_final + ru * d, *gp = g_final + rp * d, *gn = g_final + rn * d;
      for (int64_t f = 0; f < d; ++f) {
        const float u = fu[f], p = fp[f], nn = fn[f];
        gu[f] += c * (p - nn);
        gp[f] += c * u;
        gn[f] += -c * u;
      }
    }
    if (g_ego) {
      const float rs = reg_lambda / (float)B;
      float *gu = g_ego + ru * d, *gp = g_ego + rp * d, *gn = g_ego + rn * d;
      for (int64_t f = 0; f < d; ++f) {
        gu[f] += rs * eu[f];
        gp[f] += rs * ep[f];
        gn[f] += rs * en[f];
      }
    }
  }
  loss[0] = (float)(lsum / (double)B);
  loss[1] = (float)((double)reg_lambda * 0.5 * (sq[0] + sq[1] + sq[2]) / (double)B);
}

/* torch.optim.Adam defaults, single-tensor form (utility/utility_train/trainer.py:11,56). */
void orc_adam_f32(float* p, const float* g, float* m, float* v, int64_t count, double lr, double beta1,
                  double beta2, double eps, int64_t step) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2), fe = (float)eps;
  for (int64_t i = 0; i < count; ++i) {
    m[i] = m[i] + w1 * (g[i] - m[i]);
    v[i] = v[i] * b2 + w2 * g[i] * g[i];
    const float denom = sqrtf(v[i]) / bc2_sqrt + fe;
    p[i] = p[i] - step_size * (m[i] / denom);
  }
}

/* get_rating_for_test (models/LightGCN.py:74-80): sigmoid(U[users] . V^T), plain k-order
 * dot products in double rounded once (the reference's BLAS order is unspecified; tests
 * compare with a tolerance). */
void orc_score_f32(const float* U, const float* V, const int64_t* users, int64_t Bt, int64_t I, int64_t d,
                   int apply_sigmoid, float* rating) {
  for (int64_t b = 0; b < Bt; ++b) {
    const float* u = U + users[b] * d;
    for (int64_t i = 0; i < I; ++i) {
      const float* v = V + i * d;
      double acc = 0.0;
      for (int64_t f = 0; f < d; ++f) acc += (double)u[f] * (double)v[f];
      float s = (float)acc;
      if (apply_sigmoid) s = 1.0f / (1.0f + expf(-s));
      rating[b * I + i] = s;
    }
  }
}
