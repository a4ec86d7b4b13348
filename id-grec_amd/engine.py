"""Fused training / evaluation engine for the LightGCN family: the reference's per-batch
sequence (models/LightGCN.py:54-72 + utility/utility_train/trainer.py:42-56)

    aggregate() -> gather -> bpr + reg loss -> backward -> Adam.step()

as a fixed chain of C-ABI calls on preallocated device buffers — no autograd graph, no
temporaries, no host synchronisation.  Numerically it is the same chain the autograd
operators in ops.py execute (tests compare the two), so models may use either.

Embedding layout: ONE [num_users + num_items, d] panel, users first; the model's two
nn.Embedding weights are views into it (no torch.cat per step).
"""
import os

import torch

from . import native, ops


class StepPlan:
    """idg_step (include/idgrec.h): the plain LightGCN / MFBPR training step as ONE library call.  Owns the three slots'
    buffers (row bitmap, live-unit list, scatter workspace) and the plan handle; everything else belongs to the engine."""

    def __init__(self, eng, batch_capacity):
        import ctypes as C

        g, dev = eng.graph, eng.device
        self.B_cap = int(batch_capacity)
        words = (eng.n + 31) // 32
        self.bitmaps = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(native.IDG_STEP_SLOTS)]
        self.bpr_ws = [ops.bpr_workspace(self.B_cap, eng.d, dev) for _ in range(native.IDG_STEP_SLOTS)]
        self.units = [None] * native.IDG_STEP_SLOTS
        desc = native.StepDesc()
        if g is not None:
            nbytes = int(native.lib.idg_graph_live_units_bytes(g._h, 3 * self.B_cap))
            self.units = [torch.empty(nbytes // 4 + 1, dtype=torch.int32, device=dev) for _ in range(native.IDG_STEP_SLOTS)]
            self.prop_ws = g._workspace("prop", eng.d)
            desc.graph, desc.prop_ws = g._h, self.prop_ws.data_ptr()
            desc.final_panel, desc.g_final = eng.final.data_ptr(), eng.g_final.data_ptr()
        desc.num_users, desc.n, desc.d = eng.U, eng.n, eng.d
        desc.n_layers, desc.include_layer0, desc.reg_lambda = eng.K, int(eng.inc), eng.reg_lambda
        desc.params, desc.grad = eng.params.data_ptr(), eng.grad.data_ptr()
        desc.exp_avg, desc.exp_avg_sq = eng.exp_avg.data_ptr(), eng.exp_avg_sq.data_ptr()
        desc.batch_capacity = self.B_cap
        for i in range(native.IDG_STEP_SLOTS):
            desc.slot_bitmap[i] = self.bitmaps[i].data_ptr()
            desc.slot_units[i] = None if self.units[i] is None else self.units[i].data_ptr()
            desc.slot_bpr_ws[i] = self.bpr_ws[i].data_ptr()
        desc.side_stream = eng._side_raw
        desc.flags = native.IDG_STEP_PACED if eng._paced else 0
        self.key = self.key_of(eng)
        self._graph = g  # (the plan names the handle: keep it alive)
        self._h = C.c_void_p()
        native.check(native.lib.idg_step_create(C.byref(desc), C.byref(self._h)), "idg_step_create")
        self._by_ptr = {b.data_ptr(): b for b in self.bitmaps}
        self._bm = C.c_void_p()

    @staticmethod
    def key_of(eng):
        return (eng.params.data_ptr(), eng.grad.data_ptr(), eng.exp_avg.data_ptr(), eng.exp_avg_sq.data_ptr(),
                None if eng.final is None else eng.final.data_ptr(), eng.K, eng.inc, eng.reg_lambda, eng._paced)

    def last_bitmap(self):
        import ctypes as C

        native.check(native.lib.idg_step_last_bitmap(self._h, C.byref(self._bm)), "idg_step_last_bitmap")
        return self._by_ptr[self._bm.value]

    def stats(self):
        """{steps, ms_in_calls, ms_blocked, waits_skipped} since the plan was made (idg_step_stats)."""
        import ctypes as C

        out = (C.c_int64 * 4)()
        native.check(native.lib.idg_step_stats(self._h, out), "idg_step_stats")
        return {"steps": int(out[0]), "ms_in_calls": out[1] / 1e6, "ms_blocked": out[2] / 1e6, "waits_skipped": int(out[3])}

    def close(self):
        if getattr(self, "_h", None):
            native.lib.idg_step_synchronize(self._h)
            native.lib.idg_step_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class PropagationEngine:
    def __init__(self, graph, num_users, num_items, dim, n_layers, include_layer0=True, reg_lambda=1e-4, lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, deterministic=True, params=None):
        self.graph = graph  # None => no propagation (MFBPR)
        self.U, self.I, self.d = int(num_users), int(num_items), int(dim)
        self.n = self.U + self.I
        self.K, self.inc = int(n_layers), bool(include_layer0)
        self.reg_lambda, self.lr, self.betas, self.eps = float(reg_lambda), float(lr), betas, float(eps)
        self.deterministic = bool(deterministic)
        dev = graph.device if graph is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.empty((self.n, self.d), **f32) if params is None else params
        assert self.params.is_cuda and self.params.is_contiguous() and self.params.shape == (self.n, self.d)
        self.grad = torch.zeros((self.n, self.d), **f32)      # d loss / d E0
        # train_step() with the Adam update in the last backward product's epilogue: does self.grad receive the finished
        # gradient too?  True (the default): yes — loss_and_grad() semantics, what the parity tests read.  False: the update
        # consumes it in registers and the 4 B per element are not stored (what a trainer that only wants the step sets)
        self.store_grad = True
        self.exp_avg = self.exp_avg_sq = None                 # Adam moments: allocated by the first train_step()
        self.final = torch.empty((self.n, self.d), **f32) if graph is not None else None
        self.g_final = torch.zeros((self.n, self.d), **f32) if graph is not None else None
        words = (self.n + 31) // 32
        # two slots: the batch being processed and the one prepared ahead (measured: 3 or 4 change nothing)
        # three prepared-batch slots: the batch in its step, the one prepared ahead, and one whose last user (two steps back)
        # the host has SEEN complete — see _pace()
        self._slots = [self._Slot(words if graph is not None else 1, dev) for _ in range(3)] if self.deterministic else None
        self._ends = []  # end-of-step events of the last steps, oldest first
        self._paced = os.environ.get("IDG_PACE", "1") != "0"
        self._touched = None
        self._stamp = 0
        self._loss3 = torch.zeros(3, **f32)  # [bpr, reg_lambda * reg, ssl_lambda * InfoNCE]
        self.loss = self._loss3[:2]
        self.step_count = 0
        self._final_version = -1  # step_count the cached propagation belongs to
        # side stream for index-only work, claimed at construction (ops.side_stream: hardware-queue placement)
        self._side = ops.side_stream(dev) if self.deterministic else None
        self._side_raw = self._side.cuda_stream if self._side is not None else None  # launches name their stream explicitly
        self._fork = ops.LocalEvent() if self._side is not None else None  # device-local events (ops.LocalEvent) throughout
        # Receptive-field propagation (plain LightGCN step, K = 2 or 3): layer K is read at the batch's rows, so layer
        # k is produced on their (K - k)-hop neighbourhood only, and the gradient flows back through the same sets —
        # exact, and most of every product on a graph much larger than that neighbourhood (configs[4]: 15 M rows, the
        # two-hop set of a 1024-triple batch ~17 % of them).  On graphs two hops cover (yelp2018, amazon-book) it only
        # costs the hop bitmaps, hence the row threshold.  IDG_FIELDS=0/1 forces it off / on.
        want = os.environ.get("IDG_FIELDS", "")
        self._fields = (self.graph is not None and self.deterministic and self.K in (2, 3)
                        and (want == "1" or (want != "0" and self.n >= 4_000_000)))
        # IDG_COMPACT_INPUTS=1: the first backward product's flag / scan / compaction done ahead of time on the side stream
        # (Graph.compact_inputs).  Measured and left OFF (HISTORY.md, round 4): the product drops 24.3 -> 20.6 us — what
        # remains is the per-vrow cost of producing every output row — while the side-stream kernel takes 29 us of the same
        # GPU: 0.2732 vs 0.2692 ms/step on one box.
        self._compact = (self.graph is not None and self.deterministic and os.environ.get("IDG_COMPACT_INPUTS", "0") == "1"
                         and self.graph.nnz * 8 <= (2 << 30))
        self._id_storage = None  # storages of the id tensors last ordered against the main stream
        self._pp = None     # ping-pong panels of the instrumented (layer-by-layer) forward
        self.fuse_adam = True  # train_step(): Adam in the last backward epilogue (False: separate idg_adam_step_f32)
        self.ssl = None        # (eps, temperature, ssl_lambda): SimGCL's perturbed views + InfoNCE inside the fused step
        self.xssl = None       # the same triple for XSimGCL (one perturbed pass, cl_layer = 1)
        self.sgl = None        # (temperature, ssl_lambda, sub_graph_1, sub_graph_2): SGL's two edge-dropped views
        self._views = self._ssl_loss = None
        # one library call per step (StepPlan / idg_step_run_f32) for the plain LightGCN / MFBPR step; IDG_STEP_PLAN=0: the
        # call-by-call chain below (same kernels, same bits)
        self._plan, self._plan_on = None, os.environ.get("IDG_STEP_PLAN", "1") != "0"
        self._next = None      # the batch prefetch() named: handed to the step call as its next_* arguments
        self._alive = []       # id tensors of the last few batches (the side stream may still be reading them)
        self.exchange = None   # replicas (replicated.py): hook(slot, loss) -> bitmap, swaps this batch's gradient rows for
        #                        the average over all ranks' batches before the (linear) backward propagation

    def __del__(self):
        # the graph may outlive this engine: its registered bitmaps must not (a later buffer at the same address would
        # inherit a stale unit list)
        try:
            if self.graph is not None and self._slots:
                for sl in self._slots:
                    self.graph.forget_live_units(sl.bitmap)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ---- views handed to nn.Embedding
    def user_weight(self):
        return self.params[: self.U]

    def item_weight(self):
        return self.params[self.U:]

    # ---- forward only (evaluation)
    @torch.no_grad()
    def propagate(self, force=False):
        """Final user/item panels for scoring.  LightGCN re-runs aggregate() for every test
        batch (models/LightGCN.py:75) although the weights do not change between batches; the
        result is cached per optimizer step."""
        if self.graph is None:
            return self.params
        if force or self._final_version != self.step_count:
            self.graph.propagate_mean_raw(self.params, self.K, self.inc, out=self.final)
            self._final_version = self.step_count  # invalidated by loss_and_grad() / model.train()
        return self.final

    # ---- index-only preparation of a batch on the side stream (row bitmap + sorted scatter plan)
    class _Slot:
        def __init__(self, words, device):
            self.bitmap = torch.zeros(words, dtype=torch.int32, device=device)
            self.units, self.units_B = None, -1  # live work units of the bitmap (Graph.live_units), rebuilt with it
            self.compact = None  # the tiles' entry lists compacted to the bitmap's rows (Graph.compact_inputs), rebuilt with it
            self.hops = None  # [rows of the batch, + 1 hop, + 2 hops ...] (receptive-field propagation)
            self.use_fields = False
            self.ids = None
            self.ws = None
            self.ssl_ws, self.ssl_key, self.ssl_B = None, None, -1  # InfoNCE workspace holding this batch's id lists (ops.infonce_plan_raw)
            self.key = None
            self.rows_done = self.plan_done = None
            self.free = None  # = free_ev once recorded on the main stream: the step that used this slot is done
            self.free_ev = None
            self.stamp = 0    # order of last hand-out (the ring recycles the oldest)

    def _prepare(self, slot, users, pos, neg):
        main = torch.cuda.current_stream()
        B = users.shape[0]
        slot.ssl_key = None  # a plan is honoured for the batch it was built for only: set again below
        # Workspaces come from torch's caching allocator on the MAIN stream but are first written by the SIDE stream: a block
        # recycled from a tensor whose last reader is still queued on main must not be overwritten early — any
        # (re)allocation (a new batch size: the short last batch of an epoch) orders the side stream behind main once
        ssl_form = self.graph is not None and (self.ssl is not None or self.xssl is not None or self.sgl is not None)
        fresh = (slot.ws is None or slot.ws_B != B or (self.graph is not None and slot.units_B != B)
                 or (ssl_form and (slot.ssl_ws is None or slot.ssl_B != B)))
        if slot.ws is None or slot.ws_B != B:
            slot.ws, slot.ws_B = ops.bpr_workspace(B, self.d, self.device), B
            slot.rows_done, slot.plan_done, slot.free_ev = ops.LocalEvent(), ops.LocalEvent(), ops.LocalEvent()
        if slot.free is not None:
            slot.free.wait(self._side_raw)     # the step that last used this slot has consumed it
        # The side stream must not read the ids before the main stream has produced them.  Batches are slices of
        # one epoch-long tensor: ordering after the main stream once per storage is enough (doing it per batch would
        # also queue this batch's index work behind the previous step's kernels: measured +6 us/step).
        src = (users.untyped_storage().data_ptr(), pos.untyped_storage().data_ptr(), neg.untyped_storage().data_ptr())
        if slot.free is None or src != self._id_storage or fresh:
            self._id_storage = src
            self._fork.record(main.cuda_stream)
            self._fork.wait(self._side_raw)
        if self.graph is not None:  # (without propagation nothing is restricted to the batch's rows: MFBPR)
            ops.bpr_touch_rows_raw(users, pos, neg, self.U, slot.bitmap, stream=self._side_raw, clear_bits=self.n)
            # the bitmap's rows as a list of work units: the row-restricted last layer then runs one wave per unit
            # instead of visiting every tile (same bits)
            slot.units = self.graph.live_units(slot.bitmap, 3 * B, ws=slot.units if slot.units_B == B else None,
                                               stream=self._side_raw)
            slot.units_B = B
            # (... and much larger than THIS batch's neighbourhood: with ~256 distinct two-hop rows per batch row the
            #  field of a 2^20-triple batch is the graph, and the restricted kernels only cost: 355 vs 342 ms measured)
            slot.use_fields = self._fields and 3 * B * 256 <= self.n
            # the first backward product gathers from the batch's <= 3B live rows: its flag / scan / compaction of every
            # tile's entries is index-only too — done here, off the critical path (graphs whose entry list fits twice)
            if self._compact and not slot.use_fields:
                slot.compact = self.graph.compact_inputs(slot.bitmap, ws=slot.compact, stream=self._side_raw)
            if slot.use_fields:
                if slot.hops is None:
                    slot.hops = [slot.bitmap] + [torch.zeros_like(slot.bitmap) for _ in range(self.K - 1)]
                for j in range(1, self.K):
                    self.graph.expand_rows(slot.hops[j - 1], slot.hops[j], stream=self._side_raw)
            slot.rows_done.record(self._side_raw)  # needed by the last forward layer
        ops.bpr_plan_raw(users, pos, neg, self.U, self.n, self.d, ws=slot.ws, stream=self._side_raw)
        if self.graph is not None and (self.ssl is not None or self.xssl is not None or self.sgl is not None):
            # the id-list stage of the step's InfoNCE call (unique rows of the batch; SGL: raw lists, their repeat flags
            # and positions) is index-only too: ~19 us of small launches per call off the main stream
            mode = ops.SSL_RAW if self.sgl is not None else ops.SSL_UNIQUE
            if slot.ssl_ws is None or slot.ssl_B != B:
                slot.ssl_ws, slot.ssl_B = ops.infonce_workspace(self.n, B, self.d, self.device), B
            ops.infonce_plan_raw(users, pos, self.U, self.n, self.d, mode, slot.ssl_ws, stream=self._side_raw)
            slot.ssl_key = (B, mode)
        slot.plan_done.record(self._side_raw)  # needed by the gradient scatter (and by the InfoNCE call behind it)
        slot.key = (users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B)
        slot.ids = (users, pos, neg)  # keeps the id tensors alive until the side stream has read them (a caller's temporaries
        #                               would otherwise return to the allocator, and be rewritten, while still being read)

    @staticmethod
    def _ssl_plan(slot, users, mode):
        """Keyword arguments of the step's InfoNCE call: the slot's workspace with the id lists already in it — when this
        slot was prepared for this batch size and list form (an engine switched to an SSL form after a prefetch: no)."""
        if slot.ssl_ws is not None and slot.ssl_key == (int(users.shape[0]), mode):
            return {"ws": slot.ssl_ws, "planned": True}
        return {}

    def _take_slot(self):
        """Next slot of the ring: one whose batch has been consumed the longest ago (or never filled); when every slot
        holds an unconsumed prefetch, the oldest of those is recycled."""
        free = [sl for sl in self._slots if sl.key is None]
        pool = free if free else self._slots
        slot = min(pool, key=lambda sl: sl.stamp)
        self._stamp += 1
        slot.stamp = self._stamp
        return slot

    def prefetch(self, users, pos, neg):
        """Optional one-batch lookahead: prepare the NEXT batch's row bitmap and scatter plan now, so
        they are ready long before its step starts (otherwise the step prepares them itself and the
        main stream waits a few microseconds for the side stream)."""
        if self.deterministic:
            if self._plan_eligible(int(users.shape[0])):
                self._next = (users, pos, neg)  # prepared by the NEXT train_step() call, inside the library
            else:
                self._prepare(self._take_slot(), users, pos, neg)

    def _plan_eligible(self, B):
        """The one-call step covers the plain form: deterministic scatter, Adam in the epilogue, no SSL views, no gradient
        exchange, no receptive-field restriction (graphs >= 4 M rows), a tiled width."""
        if not (self._plan_on and self.deterministic and self.fuse_adam and self.ssl is None and self.xssl is None
                and self.sgl is None and self.exchange is None and not self._compact):
            return False
        if self.graph is None:
            return self.d % 4 == 0 and self.d <= 1024  # idg_step_create / idg_adam_rows_f32's limits
        return not (self._fields and 3 * B * 256 <= self.n) and self.K >= 2 and self.d in (32, 64, 128, 256, 512)

    @staticmethod
    def _ids_token(users, pos, neg):
        """idg_step_*'s ids_token of a batch: changes whenever the STORAGE of its id tensors does (never 0 = "every call")."""
        return (users.untyped_storage().data_ptr() ^ (pos.untyped_storage().data_ptr() << 1)
                ^ (neg.untyped_storage().data_ptr() << 2)) & 0xFFFFFFFFFFFFFFFF | 1

    def _run_plan(self, users, pos, neg, loss_out):
        B = int(users.shape[0])
        plan = self._plan
        if plan is None or plan.key != StepPlan.key_of(self) or B > plan.B_cap:
            if plan is not None:
                plan.close()
            plan = self._plan = StepPlan(self, max(B, plan.B_cap if plan is not None else 0))
        loss = self.loss if loss_out is None else loss_out
        nxt = self._next
        self._next = None
        if nxt is not None and (nxt[0].data_ptr() == users.data_ptr() or int(nxt[0].shape[0]) > plan.B_cap):
            nxt = None
        tok = self._ids_token(users, pos, neg)
        # the lookahead batch may live in ANOTHER storage (a new epoch's tensor, a clone, a gather result): its own token
        # orders the side stream behind this stream before it reads those ids
        ntok = 0 if nxt is None else self._ids_token(*nxt)
        self._alive.append((users, pos, neg, nxt))
        del self._alive[:-4]
        self.step_count += 1
        rc = native.lib.idg_step_run_f32(plan._h, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B,
                                         None if nxt is None else nxt[0].data_ptr(), None if nxt is None else nxt[1].data_ptr(),
                                         None if nxt is None else nxt[2].data_ptr(), 0 if nxt is None else int(nxt[0].shape[0]),
                                         tok, ntok, loss.data_ptr(), self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                         native.IDG_STEP_STORE_GRAD if self.store_grad else 0, ops._stream())
        if rc:
            native.check(rc, "idg_step_run_f32")
        self._final_version = -1
        return loss

    @property
    def touched(self):
        """Bitmap of the panel rows the last step touched."""
        if self._plan is not None and self._touched is None:
            return self._plan.last_bitmap()
        return self._touched

    @touched.setter
    def touched(self, value):
        self._touched = value

    # ---- forward + backward: losses [bpr, reg_lambda*reg] and d(sum)/dE0 into self.grad
    @torch.no_grad()
    def loss_and_grad(self, users, pos, neg, loss_out=None, _adam_step=0):
        """_adam_step > 0 (train_step): the Adam update of self.params rides in the epilogue of the last
        backward product (idg_propagate_mean_bwd_adam_f32) instead of a separate pass over the gradient.
        With self.ssl set (SimGCL): two noise-perturbed encoder passes next to the clean one, InfoNCE between
        them over the batch's unique users / positive items, its gradients added (times ssl_lambda) to the BPR
        gradient before the ONE backward propagation all three passes share; loss gets a third entry."""
        three = self.ssl is not None or self.xssl is not None or self.sgl is not None
        loss = (self._loss3 if three else self.loss) if loss_out is None else loss_out
        main = torch.cuda.current_stream()
        if self.graph is None:
            self.grad.zero_()
            if not self.deterministic:
                ops.bpr_fused_raw(self.params, self.params, users, pos, neg, self.U, self.reg_lambda, self.grad,
                                  self.grad, loss=loss, deterministic=0)
                return loss
            # the sorted scatter plan depends on the indices only: prepared on the side stream (by prefetch() during the
            # previous step, or right now) — in-call it is half of an MFBPR step
            key = (users.data_ptr(), pos.data_ptr(), neg.data_ptr(), users.shape[0])
            slot = next((sl for sl in self._slots if sl.key == key), None)
            if slot is None:
                slot = self._take_slot()
                self._prepare(slot, users, pos, neg)
            slot.key = None
            slot.plan_done.wait(main.cuda_stream)
            ops.bpr_fused_raw(self.params, self.params, users, pos, neg, self.U, self.reg_lambda, self.grad,
                              self.grad, loss=loss, deterministic=2, ws=slot.ws)
            slot.free = slot.free_ev
            slot.free.record(main.cuda_stream)
            self._final_version = -1
            return loss
        if not self.deterministic:
            self.grad.zero_()
            self.g_final.zero_()
            self.graph.propagate_mean_raw(self.params, self.K, self.inc, out=self.final)
            ops.bpr_fused_raw(self.final, self.params, users, pos, neg, self.U, self.reg_lambda, self.g_final,
                              self.grad, loss=loss, deterministic=0)
            self.graph.propagate_mean_bwd_raw(self.g_final, self.K, self.inc, out=self.grad, accumulate=True)
            self._final_version = -1
            return loss
        # deterministic path.  The batch's rows are flagged in a bitmap and its (row, slot) pairs sorted —
        # index-only work done on a side stream (by prefetch() during the previous step, or right now).
        key = (users.data_ptr(), pos.data_ptr(), neg.data_ptr(), users.shape[0])
        slot = next((sl for sl in self._slots if sl.key == key), None)
        ahead = slot is not None
        if slot is None:  # not prefetched: prepare it now (the main stream then waits for the side stream)
            slot = self._take_slot()
            self._prepare(slot, users, pos, neg)
        slot.key = None
        self.touched = slot.bitmap
        # Pacing.  The host enqueues a step in less time than the device runs it and would get hundreds of steps ahead; it
        # now blocks until the step before the previous one has finished (two steps stay queued: the device never runs dry).
        # A batch prepared ahead — its preparation started when ITS slot's last step, three back, had finished — is then
        # complete by the time its step is enqueued, the host can see that (event query), and the step's stream does not
        # need the wait: a barrier packet costs 4.5 us between two kernels whether or not its event has fired.
        if self._paced and len(self._ends) >= 2:
            self._ends[-2].synchronize()
        del self._ends[:-2]  # (also when not paced: the list must not grow by an entry per step)
        ready = ahead and self._paced and slot.plan_done.query()
        if not ready:
            (slot.plan_done if ahead else slot.rows_done).wait(main.cuda_stream)
        if self.ssl is not None:
            # SimGCL (models/SimGCL.py:62-66): the clean pass and two perturbed ones, read at rows of the batch only
            # (unique users / positives are a subset of the bitmap); the first product is shared between the passes
            eps, temperature, ssl_lambda = self.ssl
            if self._views is None:
                self._views = tuple(torch.empty_like(self.params) for _ in range(4))  # two views + two scratch panels
                self._ssl_loss = torch.zeros(2, dtype=torch.float32, device=self.device)
            streams = [ops._next_noise_stream(), ops._next_noise_stream()]  # the sequence ops.propagate_views draws
            ops.propagate_views_raw(self.graph, self.params, self.K, self.inc, eps, streams,
                                    [self.final, self._views[0], self._views[1]], out_rows=slot.bitmap,
                                    scratch=self._views[2:])
        elif self.xssl is not None:
            # XSimGCL (models/XSimGCL.py:40-60): ONE perturbed pass; BPR reads its layer mean, InfoNCE contrasts the
            # first layer's output with that mean.  Both are wanted at rows of the batch only.
            eps, temperature, ssl_lambda = self.xssl
            if self._views is None:
                self._views = (torch.empty_like(self.params), torch.empty_like(self.params))  # (layer-1 view, its gradient)
                self._ssl_loss = torch.zeros(2, dtype=torch.float32, device=self.device)
            stream = ops._next_noise_stream()
            self.graph.propagate_mean_noise_raw(self.params, self.K, self.inc, eps, stream[0], stream[1], out=self.final,
                                                out_rows=slot.bitmap)
            seed1, sid1 = ops.layer_noise_stream(stream, 1)  # layer 1 again, on its own, for the rows of the batch
            ops.spmm_noise_raw(self.graph, self.params, eps, seed1, sid1, out=self._views[0], out_rows=slot.bitmap)
        elif slot.use_fields and self.sgl is None and self.exchange is None:
            # layer 1 on the widest hop set ... layer K on the batch's rows.  (The widest set holds the popular items
            # and the active users — most stored entries point into it although it is a minority of the rows — and the
            # row-restricted kernel is no faster there than the dense one: leaving layer 1 dense measured 216 vs 212 ms.)
            self.graph.propagate_mean_fields_raw(self.params, self.K, self.inc, self.final, slot.hops[::-1])
        else:
            self.graph.propagate_mean_raw(self.params, self.K, self.inc, out=self.final, out_rows=slot.bitmap)
        if self.sgl is not None:
            # SGL (models/SGL.py:75-101): the same encoder on two edge-dropped sub-graphs of this epoch; their layer
            # means are contrasted at the batch's users / positives (raw ids, duplicates count)
            temperature, ssl_lambda, sub_1, sub_2 = self.sgl
            if self._views is None:
                self._views = tuple(torch.empty_like(self.params) for _ in range(4))  # two views, their gradients
                self._ssl_loss = torch.zeros(2, dtype=torch.float32, device=self.device)
            for sub in (sub_1, sub_2):  # copies of the full graph's handle share its schedule: the same unit list serves
                if getattr(sub, "_base", None) is self.graph and slot.units is not None:
                    sub.bind_live_units(slot.bitmap, slot.units, 3 * users.shape[0])
            sub_1.propagate_mean_raw(self.params, self.K, self.inc, out=self._views[0], out_rows=slot.bitmap)
            sub_2.propagate_mean_raw(self.params, self.K, self.inc, out=self._views[1], out_rows=slot.bitmap)
        assert self.exchange is None or not three, "gradient-row exchange: LightGCN-family steps only"
        if not ahead:
            slot.plan_done.wait(main.cuda_stream)
        # reached rows of g_final and of the regulariser gradient (self.grad) are STORED and the backward
        # propagation reads flagged rows only: neither panel is ever zero-filled
        # (the bitmap already holds the batch's rows: with TOUCHED_PRESET the scatter stores its rows without writing it, so
        #  the lists registered for it — the compacted inputs of the first backward product — stay valid)
        ops.bpr_fused_raw(self.final, self.params, users, pos, neg, self.U, self.reg_lambda, self.g_final,
                          self.grad, loss=loss[:2], deterministic=2 | native.IDG_BPR_TOUCHED_PRESET, touched=slot.bitmap,
                          ws=slot.ws)
        if self.ssl is not None:
            # d(ssl_lambda * InfoNCE)/d view_1 + d(...)/d view_2 join the BPR gradient in g_final's (stored) rows
            ops.infonce_pair_raw(self._views[0], self._views[1], users, pos, self.U, temperature, g1=self.g_final,
                                 g2=self.g_final, loss=self._ssl_loss, grad_scale=ssl_lambda, accumulate=True,
                                 **self._ssl_plan(slot, users, ops.SSL_UNIQUE))
            torch.sum(self._ssl_loss, dim=0, keepdim=True, out=loss[2:3])
            loss[2:3].mul_(ssl_lambda)
        if self.sgl is not None:
            g_1, g_2 = self._views[2], self._views[3]
            g_1.zero_()
            g_2.zero_()
            ops.infonce_pair_raw(self._views[0], self._views[1], users, pos, self.U, temperature, g1=g_1, g2=g_2,
                                 loss=self._ssl_loss, dedup=False, grad_scale=ssl_lambda, accumulate=True,
                                 **self._ssl_plan(slot, users, ops.SSL_RAW))
            torch.sum(self._ssl_loss, dim=0, keepdim=True, out=loss[2:3])
            loss[2:3].mul_(ssl_lambda)
            # three encoders, three backward propagations (each sub-graph is its own symmetric operator), one gradient
            self.graph.propagate_mean_bwd_raw(self.g_final, self.K, self.inc, out=self.grad, accumulate=True, mask=slot.bitmap)
            sub_1.propagate_mean_bwd_raw(g_1, self.K, self.inc, out=self.grad, accumulate=True)
            sub_2.propagate_mean_bwd_raw(g_2, self.K, self.inc, out=self.grad, accumulate=True)
            if _adam_step > 0:
                ops.adam_step_raw(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, _adam_step, self.betas[0],
                                  self.betas[1], self.eps)
            slot.free = slot.free_ev
            slot.free.record(main.cuda_stream)
            self._ends.append(slot.free)
            self._final_version = -1
            return loss
        if self.xssl is not None:
            # InfoNCE(layer-1 view, layer mean): the mean's share joins the BPR gradient in g_final's stored rows; the
            # view's share (added into its own, cleared, panel) reaches E0 through one more product, A . g_view, below
            g_view = self._views[1]
            g_view.zero_()
            ops.infonce_pair_raw(self._views[0], self.final, users, pos, self.U, temperature, g1=g_view, g2=self.g_final,
                                 loss=self._ssl_loss, grad_scale=ssl_lambda, accumulate=True,
                                 **self._ssl_plan(slot, users, ops.SSL_UNIQUE))
            torch.sum(self._ssl_loss, dim=0, keepdim=True, out=loss[2:3])
            loss[2:3].mul_(ssl_lambda)
            if _adam_step > 0 and self.K >= 2 and not self.inc:
                # d loss / d E0 = (A g + .. + A^K g) / K + A g_view = A ((g + A (g + ..)) / K + g_view): the Horner chain with
                # its division moved in front of the LAST product, whose input then also carries the view's gradient rows
                # (added by the step before it: sum_out accumulates into the panel the InfoNCE call filled) and whose
                # epilogue applies Adam — no extra product for A . g_view, no separate Adam pass, no gradient panel
                # (round 4: -28 us of a 430 us step; rounding differs from the form below in the last place)
                bm, K = slot.bitmap, self.K
                if getattr(self, "_xb", None) is None:
                    self._xb = [torch.empty_like(self.params), torch.empty_like(self.params)]
                X, xr = self.g_final, bm
                for k in range(1, K - 1):  # h <- A . h + g
                    ops.spmm_epi_raw(self.graph, X, Y=self._xb[(k - 1) & 1], addend=self.g_final, mask=bm, x_rows=xr)
                    X, xr = self._xb[(k - 1) & 1], None
                ops.spmm_epi_raw(self.graph, X, sum_in=self.g_final, sum_out=g_view, div=float(K), accumulate=True, mask=bm,
                                 x_rows=xr)
                ops.spmm_epi_raw(self.graph, g_view, sum_out=self.grad, accumulate=True, mask=bm,
                                 adam=(self.params, self.exp_avg, self.exp_avg_sq, self.lr, _adam_step, self.betas[0],
                                       self.betas[1], self.eps), adam_discard_grad=not self.store_grad)
            else:
                self.graph.propagate_mean_bwd_raw(self.g_final, self.K, self.inc, out=self.grad, accumulate=True, mask=slot.bitmap)
                # grad += A . g_view (d view / d E0 = A for the first layer): live rows of g_view are the batch's unique
                # users and positives, a subset of the bitmap (its other rows read zeros)
                ops.spmm_ex_raw(self.graph, g_view, sum_in=self.grad, sum_out=self.grad, x_rows=slot.bitmap)
                if _adam_step > 0:
                    ops.adam_step_raw(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, _adam_step, self.betas[0],
                                      self.betas[1], self.eps)
            slot.free = slot.free_ev
            slot.free.record(main.cuda_stream)
            self._ends.append(slot.free)
            self._final_version = -1
            return loss
        mask = slot.bitmap
        if self.exchange is not None:
            mask = self.touched = self.exchange(slot, loss)
        if _adam_step > 0 and slot.use_fields and not three and self.exchange is None:
            # g_final is non-zero on the batch's rows; step k's input lives on their (k - 1)-hop set
            self.graph.propagate_mean_bwd_adam_fields_raw(self.g_final, self.K, self.inc, self.grad, True,
                                                          slot.hops[: self.K - 1] + [None], self.params, self.exp_avg,
                                                          self.exp_avg_sq, self.lr, _adam_step, self.betas[0],
                                                          self.betas[1], self.eps, discard_grad=not self.store_grad)
        elif _adam_step > 0:
            self.graph.propagate_mean_bwd_adam_raw(self.g_final, self.K, self.inc, self.grad, True, mask, self.params,
                                                   self.exp_avg, self.exp_avg_sq, self.lr, _adam_step, self.betas[0],
                                                   self.betas[1], self.eps, discard_grad=not self.store_grad)
        else:
            self.graph.propagate_mean_bwd_raw(self.g_final, self.K, self.inc, out=self.grad, accumulate=True, mask=mask)
        slot.free = slot.free_ev
        slot.free.record(main.cuda_stream)
        self._ends.append(slot.free)
        self._final_version = -1
        return loss

    # ---- one whole training step (loss_and_grad + dense Adam), for callers without a torch optimizer
    @torch.no_grad()
    def train_step(self, users, pos, neg, loss_out=None):
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.params), torch.zeros_like(self.params)
        if self._plan_eligible(int(users.shape[0])):
            for t, dt in ((users, torch.int64), (pos, torch.int64), (neg, torch.int64), (loss_out, torch.float32)):
                if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == dt):
                    raise TypeError("train_step needs contiguous device tensors: int64 ids, a float32 loss vector")
            self._touched = None
            return self._run_plan(users, pos, neg, loss_out)
        if self._next is not None:  # named for the one-call step, which does not apply after all: prepare it the other way
            nxt, self._next = self._next, None
            self._prepare(self._take_slot(), *nxt)
        if self.graph is not None and self.deterministic and self.fuse_adam:
            self.step_count += 1
            return self.loss_and_grad(users, pos, neg, loss_out, _adam_step=self.step_count)
        loss = self.loss_and_grad(users, pos, neg, loss_out)
        self.step_count += 1
        ops.adam_step_raw(self.params, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.step_count,
                          self.betas[0], self.betas[1], self.eps)
        return loss

    def forward_layer(self, k, out_rows=None):
        """Layer k (1-based, k < K) of the forward propagation as ONE idg_spmm_ex_f32 call, in the form
        idg_propagate_mean_f32 launches it inside a training step: a plain product into a layer buffer (for K <= 3 the
        layer sum is formed by the last product's epilogue only).  bench.py times the dominant dense launch through
        this, outside its timed region."""
        assert 1 <= k < self.K
        if self._pp is None:
            self._pp = [torch.empty_like(self.params), torch.empty_like(self.params)]
        X = self.params if k == 1 else self._pp[(k - 2) & 1]
        ops.spmm_ex_raw(self.graph, X, Y=self._pp[(k - 1) & 1])

    # ---- evaluation
    @torch.no_grad()
    def topk(self, users, k, excl_indptr=None, excl_items=None):
        fin = self.propagate()
        return ops.score_topk(fin[: self.U], fin[self.U:], users, k, excl_indptr, excl_items)

    @torch.no_grad()
    def rating(self, users):
        fin = self.propagate()
        return ops.score_dense(fin[: self.U], fin[self.U:], users)


class BatchPrep:
    """The index-only work of a batch — the bitmap of its <= 3B panel rows, (optionally) the live work units of that bitmap
    on a graph, the sorted scatter plan of its (row, slot) pairs — on a SIDE stream, one batch ahead of the step that uses
    it: what PropagationEngine does for the LightGCN family, for engines with their own step (EgcfEngine, NgcfEngine).
    Three slots: the batch being processed, the one prepared ahead, and one whose last step the host has seen complete.

        prep.prefetch(users, pos, neg)            # the trainer's lookahead (optional)
        slot = prep.take(users, pos, neg)         # the step: finds the prepared slot (or prepares now), orders the main
        ...  slot.bitmap / slot.units / slot.ws   #   stream behind it
        prep.release(slot)                        # after the last launch that reads them
    """

    class _Slot:
        def __init__(self, words, device):
            self.bitmap = torch.zeros(words, dtype=torch.int32, device=device)
            self.units, self.ws, self.B = None, None, -1
            self.done, self.free_ev, self.free = ops.LocalEvent(), ops.LocalEvent(), None
            self.key, self.ids, self.stamp = None, None, 0

    def __init__(self, num_users, n_rows, dim, device, units_graph=None, extra=None):
        """extra(slot, users, pos, neg, raw_stream): more index-only work of the batch, enqueued on the side stream before the
        slot's `done` event (EgcfEngine: the id lists of its two InfoNCE calls)."""
        self.U, self.n, self.d, self.device, self.graph = int(num_users), int(n_rows), int(dim), device, units_graph
        self.extra = extra
        self._slots = [self._Slot((self.n + 31) // 32, device) for _ in range(3)]  # (three: see PropagationEngine's pacing)
        self._ends = []
        self._paced = os.environ.get("IDG_PACE", "1") != "0"
        self._side = ops.side_stream(device)
        self._side_raw = self._side.cuda_stream
        self._fork = ops.LocalEvent()
        self._id_storage, self._stamp = None, 0

    def _prepare(self, slot, users, pos, neg):
        main = torch.cuda.current_stream()
        B = int(users.shape[0])
        fresh = slot.B != B  # workspaces (re)allocated on main, first written on the side stream: order it behind main once
        if fresh:
            slot.ws, slot.units, slot.B = ops.bpr_workspace(B, self.d, self.device), None, B
        if slot.free is not None:
            slot.free.wait(self._side_raw)  # the step that last used this slot has consumed it
        # the side stream must not read the ids before the main stream has produced them: once per storage (batches are
        # slices of one epoch-long tensor), as PropagationEngine._prepare does
        src = (users.untyped_storage().data_ptr(), pos.untyped_storage().data_ptr(), neg.untyped_storage().data_ptr())
        if slot.free is None or src != self._id_storage or fresh:
            self._id_storage = src
            self._fork.record(main.cuda_stream)
            self._fork.wait(self._side_raw)
        ops.bpr_touch_rows_raw(users, pos, neg, self.U, slot.bitmap, stream=self._side_raw, clear_bits=self.n)
        if self.graph is not None:
            slot.units = self.graph.live_units(slot.bitmap, 3 * B, ws=slot.units, stream=self._side_raw)
        ops.bpr_plan_raw(users, pos, neg, self.U, self.n, self.d, ws=slot.ws, stream=self._side_raw)
        if self.extra is not None:
            self.extra(slot, users, pos, neg, self._side_raw)
        slot.done.record(self._side_raw)
        slot.key = (users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B)
        slot.ids = (users, pos, neg)  # alive until the side stream has read them

    def _next(self):
        free = [sl for sl in self._slots if sl.key is None]
        slot = min(free if free else self._slots, key=lambda sl: sl.stamp)
        self._stamp += 1
        slot.stamp = self._stamp
        return slot

    def prefetch(self, users, pos, neg):
        self._prepare(self._next(), users, pos, neg)

    def take(self, users, pos, neg):
        key = (users.data_ptr(), pos.data_ptr(), neg.data_ptr(), int(users.shape[0]))
        slot = next((sl for sl in self._slots if sl.key == key), None)
        ahead = slot is not None
        if slot is None:
            slot = self._next()
            self._prepare(slot, users, pos, neg)
        slot.key = None
        # the host stays at most two steps ahead of the device; a batch prepared ahead is then known to be ready when its
        # step is enqueued, and the step's stream needs no wait (PropagationEngine.loss_and_grad: 4.5 us per barrier packet)
        if self._paced and len(self._ends) >= 2:
            self._ends[-2].synchronize()
        del self._ends[:-2]  # (also when not paced)
        if not (ahead and self._paced and slot.done.query()):
            slot.done.wait(torch.cuda.current_stream().cuda_stream)
        return slot

    def release(self, slot):
        slot.free = slot.free_ev
        slot.free.record(torch.cuda.current_stream().cuda_stream)
        self._ends.append(slot.free)
