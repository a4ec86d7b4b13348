// One library call per training step (include/idgrec.h: idg_step_*).
//
// The reference's step is one Python statement sequence (utility/utility_train/trainer.py:42-56: model(batch), backward(),
// Adam.step()).  The kernels behind it were already here; what this file removes is the HOST cost of driving them: the
// engine used to issue ~8 ctypes calls per step plus tensor slicing, event bookkeeping and slot search in Python, 0.24 ms
// of host time per 0.26 ms step.  A plan names every buffer once; a step is then ONE call that walks the same chain —
// the next batch's index-only preparation on the side stream, this batch's products / loss / backward / Adam on the
// caller's stream — with the slot ring, the events and the pacing kept here.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <new>

#include "idg_common.h"

struct idg_step {
  idg_step_desc d;
  struct Slot {
    hipEvent_t rows_done = nullptr, plan_done = nullptr, free_ev = nullptr;
    bool used = false;        // free_ev has been recorded at least once
    bool prepared = false;    // plan_done has been recorded at least once (the side stream has written this slot's buffers)
    bool has_key = false;     // holds a prepared batch nobody has run yet
    const int64_t *users = nullptr, *pos = nullptr, *neg = nullptr;
    int64_t B = 0;
    uint64_t stamp = 0;
  } slots[IDG_STEP_SLOTS];
  hipEvent_t fork = nullptr;
  hipEvent_t ends[2] = {nullptr, nullptr};  // end-of-step events of the last two steps, oldest first
  int n_ends = 0;
  uint64_t stamp = 0, token = 0;
  bool forked_once = false;
  int last_slot = -1;
  int64_t n_steps = 0, ns_in_calls = 0, ns_blocked = 0, n_waits_skipped = 0;  // idg_step_stats
};

namespace {

#define IDG_TRY(call)                  \
  do {                                 \
    int rc_ = (call);                  \
    if (rc_ != IDG_OK) return rc_;     \
  } while (0)

int take_slot(idg_step* p) {
  int best = -1;
  bool any_free = false;
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) any_free = any_free || !p->slots[i].has_key;
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) {
    if (any_free && p->slots[i].has_key) continue;  // an unconsumed prefetch is recycled only when nothing else is left
    if (best < 0 || p->slots[i].stamp < p->slots[best].stamp) best = i;
  }
  p->slots[best].stamp = ++p->stamp;
  return best;
}

// index-only work of one batch on the side stream: the bitmap of its <= 3B panel rows, the live work units of that bitmap
// (the row-restricted last forward layer runs one wave per unit), the sorted scatter plan of its (row, slot) pairs
int prepare(idg_step* p, int si, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, uint64_t token,
            void* main_stream) {
  const idg_step_desc& d = p->d;
  idg_step::Slot& s = p->slots[si];
  void* side = d.side_stream;
  if (s.used) IDG_TRY(idg_stream_wait_event(side, s.free_ev));  // the step that last used this slot has consumed it
  // the side stream must not read the ids before the caller's stream has produced them: once per storage (batches are
  // slices of one epoch-long tensor; per batch it would also queue this batch's index work behind the previous step).
  // `token` names the storage of the batch BEING PREPARED — a lookahead batch brings its own (the step's
  // token used for the next batch let a lookahead in a new storage through without the fork)
  if (!s.used || token == 0 || token != p->token || !p->forked_once) {
    p->token = token;
    p->forked_once = true;
    IDG_TRY(idg_event_record(p->fork, main_stream));
    IDG_TRY(idg_stream_wait_event(side, p->fork));
  }
  // the row bitmap is set by the plan's first kernel (idg_bpr_plan_rows_f32); with a graph its live work units follow
  // (with a graph the side stream has slack and the host does not: up to 4096 pairs are sorted by one launch, and the
  //  plan's first kernel also clears the header of the unit list built next — two launches less per step)
  IDG_TRY(idg_bitmap_clear(d.slot_bitmap[si], d.n, side));
  IDG_TRY(idg::bpr_plan_rows(users, pos, neg, B, d.num_users, d.n, d.slot_bpr_ws[si], d.slot_bitmap[si],
                             d.graph ? reinterpret_cast<uint32_t*>(d.slot_units[si]) : nullptr, d.graph != nullptr, side));
  if (d.graph) {
    IDG_TRY(idg::live_units_prezeroed(d.graph, d.slot_bitmap[si], d.slot_units[si], 3 * B, side));
    IDG_TRY(idg_event_record(s.rows_done, side));
  }
  IDG_TRY(idg_event_record(s.plan_done, side));
  s.prepared = true;
  s.has_key = true;
  s.users = users, s.pos = pos, s.neg = neg, s.B = B;
  return IDG_OK;
}

int check_batch(const idg_step* p, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, const char* who) {
  if (!users || !pos || !neg) return idg::fail(IDG_E_INVALID, "%s: NULL id array", who);
  if (B <= 0 || B > p->d.batch_capacity)
    return idg::fail(IDG_E_INVALID, "%s: batch of %lld triples, plan built for up to %lld", who, (long long)B,
                     (long long)p->d.batch_capacity);
  return IDG_OK;
}

}  // namespace

extern "C" {

int idg_step_create(const idg_step_desc* desc, idg_step** out) {
  IDG_REQUIRE(desc && out, "idg_step_create: NULL argument");
  const idg_step_desc& d = *desc;
  IDG_REQUIRE(d.num_users >= 0 && d.n >= d.num_users && d.n > 0 && d.d > 0 && d.batch_capacity > 0, "idg_step_create: bad sizes");
  IDG_REQUIRE(d.params && d.grad && d.exp_avg && d.exp_avg_sq && d.side_stream, "idg_step_create: NULL panel / side stream");
  IDG_REQUIRE(!d.graph || (d.final_panel && d.g_final && d.prop_ws && d.n_layers >= 1),
              "idg_step_create: a propagating plan needs final_panel, g_final, prop_ws and n_layers >= 1");
  IDG_REQUIRE(d.graph || (d.d % 4 == 0 && d.d <= 1024), "idg_step_create: without propagation the width must be a multiple of 4, at most 1024");
  for (int i = 0; i < IDG_STEP_SLOTS; ++i)
    IDG_REQUIRE(d.slot_bitmap[i] && d.slot_bpr_ws[i] && (!d.graph || d.slot_units[i]), "idg_step_create: slot %d lacks a buffer", i);
  idg_step* p = new (std::nothrow) idg_step();
  if (!p) return idg::fail(IDG_E_NOMEM, "idg_step_create: out of host memory");
  p->d = d;
  int rc = idg_event_create((void**)&p->fork);
  for (int i = 0; i < IDG_STEP_SLOTS && rc == IDG_OK; ++i) {
    rc = idg_event_create((void**)&p->slots[i].rows_done);
    if (rc == IDG_OK) rc = idg_event_create((void**)&p->slots[i].plan_done);
    if (rc == IDG_OK) rc = idg_event_create((void**)&p->slots[i].free_ev);
  }
  if (rc != IDG_OK) {
    idg_step_destroy(p);
    return rc;
  }
  *out = p;
  return IDG_OK;
}

int idg_step_destroy(idg_step* p) {
  if (!p) return IDG_OK;
  if (p->d.graph)
    for (int i = 0; i < IDG_STEP_SLOTS; ++i) {
      // the unit lists registered for the slots' bitmaps name caller-owned buffers: they go with the plan
      idg_graph_forget_live_units(p->d.graph, p->d.slot_bitmap[i]);
      idg_graph_forget_units_ws(p->d.slot_units[i]);
    }
  idg_event_destroy(p->fork);
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) {
    idg_event_destroy(p->slots[i].rows_done);
    idg_event_destroy(p->slots[i].plan_done);
    idg_event_destroy(p->slots[i].free_ev);
  }
  delete p;
  return IDG_OK;
}

int idg_step_prefetch(idg_step* p, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, uint64_t ids_token,
                      void* stream) {
  IDG_REQUIRE(p, "idg_step_prefetch: NULL plan");
  IDG_TRY(check_batch(p, users, pos, neg, B, "idg_step_prefetch"));
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) {
    const idg_step::Slot& s = p->slots[i];
    if (s.has_key && s.users == users && s.pos == pos && s.neg == neg && s.B == B) return IDG_OK;  // prepared already
  }
  return prepare(p, take_slot(p), users, pos, neg, B, ids_token, stream);
}

int idg_step_run_f32(idg_step* p, const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                     const int64_t* next_users, const int64_t* next_pos, const int64_t* next_neg, int64_t next_B,
                     uint64_t ids_token, uint64_t next_ids_token, float* loss, int64_t adam_step, double lr, double beta1,
                     double beta2, double eps, int flags, void* stream) {
  IDG_REQUIRE(p && loss && adam_step >= 1, "idg_step_run_f32: NULL plan / loss, or adam_step < 1");
  IDG_TRY(check_batch(p, users, pos, neg, B, "idg_step_run_f32"));
  const auto t_in = std::chrono::steady_clock::now();
  const idg_step_desc& d = p->d;
  // this batch's slot: prepared a call ahead, or now
  int si = -1;
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) {
    const idg_step::Slot& s = p->slots[i];
    if (s.has_key && s.users == users && s.pos == pos && s.neg == neg && s.B == B) si = i;
  }
  const bool ahead = si >= 0;
  if (!ahead) {
    si = take_slot(p);
    IDG_TRY(prepare(p, si, users, pos, neg, B, ids_token, stream));
  }
  idg_step::Slot& s = p->slots[si];
  // A prepared batch is honoured by the NEXT step only: whatever else was prepared and not run (a lookahead nobody came
  // for) is dropped now — its id arrays may be gone, and the same addresses may come back holding other ids
  for (int i = 0; i < IDG_STEP_SLOTS; ++i) p->slots[i].has_key = false;
  // the next batch's index-only work goes out first: it runs on the side stream under this step's products
  if (next_B > 0) IDG_TRY(idg_step_prefetch(p, next_users, next_pos, next_neg, next_B, next_ids_token, stream));
  // pacing: at most two steps queued.  A batch prepared ahead started its preparation when ITS slot's last step, three
  // back, had finished — complete by now, which the host can see; the step's stream then needs no wait at all
  const bool paced = (d.flags & IDG_STEP_PACED) != 0;
  if (paced && p->n_ends >= 2) {
    const auto t_b = std::chrono::steady_clock::now();
    IDG_TRY(idg_event_synchronize(p->ends[0]));
    p->ns_blocked += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_b).count();
  }
  int ready = 0;
  if (ahead && paced) IDG_TRY(idg_event_query(s.plan_done, &ready));
  p->n_waits_skipped += ready;
  const int store = ((flags | d.flags) & IDG_STEP_STORE_GRAD) ? 1 : 0;
  if (d.graph) {
    if (!ready) IDG_TRY(idg_stream_wait_event(stream, ahead ? s.plan_done : s.rows_done));
    IDG_TRY(idg_propagate_mean_f32(d.graph, d.params, d.final_panel, d.slot_bitmap[si], d.n_layers, d.include_layer0, d.d,
                                   d.prop_ws, stream));
    if (!ahead) IDG_TRY(idg_stream_wait_event(stream, s.plan_done));
    // reached rows of g_final and of the regulariser's gradient (grad) are STORED; the backward reads flagged rows only
    IDG_TRY(idg_bpr_fused_f32(d.final_panel, d.params, d.num_users, d.n, users, pos, neg, B, d.d, d.reg_lambda, loss, d.g_final,
                              d.grad, IDG_BPR_PLANNED | IDG_BPR_TOUCHED_PRESET, d.slot_bitmap[si], d.slot_bpr_ws[si], stream));
    IDG_TRY(idg_propagate_mean_bwd_adam_f32(d.graph, d.g_final, d.slot_bitmap[si], d.grad, d.n_layers, d.include_layer0, d.d,
                                            1 | (store ? 0 : IDG_ADAM_DISCARD_GRAD), d.params, d.exp_avg, d.exp_avg_sq, lr,
                                            beta1, beta2, eps, adam_step, d.prop_ws, stream));
  } else {
    if (!ready) IDG_TRY(idg_stream_wait_event(stream, s.plan_done));
    // (a caller that reads the gradient panel wants zeros outside the batch's rows; the update itself does not need them)
    if (store) IDG_HIP(hipMemsetAsync(d.grad, 0, (size_t)d.n * (size_t)d.d * sizeof(float), (hipStream_t)stream));
    IDG_TRY(idg_bpr_fused_f32(d.params, d.params, d.num_users, d.n, users, pos, neg, B, d.d, d.reg_lambda, loss, d.grad, d.grad,
                              IDG_BPR_PLANNED | IDG_BPR_TOUCHED_PRESET, d.slot_bitmap[si], d.slot_bpr_ws[si], stream));
    IDG_TRY(idg_adam_rows_f32(d.params, d.grad, d.slot_bitmap[si], d.exp_avg, d.exp_avg_sq, d.n, d.d, lr, beta1, beta2, eps,
                              adam_step, stream));
  }
  IDG_TRY(idg_event_record(s.free_ev, stream));
  s.used = true;
  p->last_slot = si;
  if (p->n_ends == 2) p->ends[0] = p->ends[1], p->n_ends = 1;
  p->ends[p->n_ends++] = s.free_ev;
  p->n_steps += 1;
  p->ns_in_calls += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_in).count();
  return IDG_OK;
}

int idg_step_stats(const idg_step* p, int64_t out[4]) {
  IDG_REQUIRE(p && out, "idg_step_stats: NULL argument");
  out[0] = p->n_steps, out[1] = p->ns_in_calls, out[2] = p->ns_blocked, out[3] = p->n_waits_skipped;
  return IDG_OK;
}

int idg_step_last_bitmap(const idg_step* p, const uint32_t** out_bitmap) {
  IDG_REQUIRE(p && out_bitmap, "idg_step_last_bitmap: NULL argument");
  IDG_REQUIRE(p->last_slot >= 0, "idg_step_last_bitmap: no step has run yet");
  *out_bitmap = p->d.slot_bitmap[p->last_slot];
  return IDG_OK;
}

int idg_step_synchronize(idg_step* p) {
  IDG_REQUIRE(p, "idg_step_synchronize: NULL plan");
  for (int i = 0; i < p->n_ends; ++i) IDG_TRY(idg_event_synchronize(p->ends[i]));
  // ... and every preparation enqueued on the SIDE stream: a lookahead nobody ran may still be writing its slot's
  // bitmap / unit list / plan workspace, which the caller is about to release
  for (int i = 0; i < IDG_STEP_SLOTS; ++i)
    if (p->slots[i].prepared) IDG_TRY(idg_event_synchronize(p->slots[i].plan_done));
  return IDG_OK;
}

}  // extern "C"
