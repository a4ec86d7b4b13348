"""Device operators: torch tensors in, torch tensors out, computed by libidgrec.so's HIP
kernels through the C ABI.  PyTorch only supplies device memory, streams and autograd
bookkeeping.  Every operator requires CUDA(=HIP) tensors; there is no CPU implementation —
calling one with CPU tensors raises.

Operator boundary: `spmm(graph, X)` stands where the reference calls
`torch.sparse.mm(self.Graph, X)` (models/LightGCN.py:44); `propagate_mean` is the whole
`aggregate()` loop (models/LightGCN.py:36-52); `bpr_loss` is the gather + get_bpr_loss +
reg_lambda*get_reg_loss of `forward()` (models/LightGCN.py:57-68); `score_topk` is
get_rating_for_test + mask + torch.topk (utility/utility_train/batch_test.py:59-68).
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import native
from .native import check, lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~8 us): with ~40 launches per sharded step that alone made the host
    the bottleneck, so the raw accessor is used when this torch has it."""
    if _raw_stream is not None:
        return _raw_stream(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "idgrec_amd operators run on MI355X only: got a %s tensor. There is no CPU fallback; "
                "move the model and inputs to the GPU (device='cuda')." % t.device)


def _require_ids(*tensors):
    """The raw entry points read ids as contiguous int64 (the reference's batches, trainer.py:27-29): anything else
    would be read past its end."""
    for t in tensors:
        if t.dtype != torch.int64 or not t.is_contiguous():
            raise TypeError("batch ids must be contiguous int64 tensors (got %s, contiguous=%s)" % (t.dtype, t.is_contiguous()))


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _i64c(t, name):
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return t.data_ptr() if t is not None else None  # ctypes converts int / None for void* parameters


def _forget_units_ws(ws_ptr):
    """Finalizer of a list buffer: whatever is registered IN that buffer goes (a newer list of the same bitmap, built into
    another buffer, stays: ADVICE r03)."""
    if lib is not None:
        lib.idg_graph_forget_units_ws(ws_ptr)


def _forget_units_of(graph_ref, bitmap_ptr):
    g = graph_ref()
    if g is not None and getattr(g, "_h", None) and lib is not None:
        lib.idg_graph_forget_live_units(g._h, bitmap_ptr)


class Graph:
    """Device-resident CSR adjacency + its row-block tile schedule (idg_graph).

    Built from HOST CSR arrays (numpy / scipy).  `symmetric=True` declares A == A^T so the
    backward pass reuses the same handle (true for the normalised bipartite adjacency,
    SURVEY §0.6); otherwise the transposed CSR is built once on the host.
    """

    def __init__(self, indptr, indices, values, n_rows, n_cols, device=None, symmetric=True, exact_order=False,
                 split_threshold=0, build_transpose=True, _transpose_of=None):
        if not torch.cuda.is_available():
            raise RuntimeError("idgrec_amd.Graph needs a HIP device (torch.cuda.is_available() is False); "
                               "this library has no CPU path.")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("idgrec_amd.Graph: device must be a cuda(HIP) device, got %s" % dev)
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        self.nnz = int(indices.shape[0])
        self.symmetric = bool(symmetric)
        flags = (native.IDG_GRAPH_SYMMETRIC if symmetric else 0) | (native.IDG_GRAPH_EXACT_ORDER if exact_order else 0)
        h = C.c_void_p()
        check(lib.idg_graph_create(self.device.index, self.n_rows, self.n_cols, self.nnz,
                                   native.np_ptr(indptr, C.c_int64), native.np_ptr(indices, C.c_int32),
                                   native.np_ptr(values, C.c_float), flags, int(split_threshold), C.byref(h)),
              "idg_graph_create")
        self._h = h
        self._ws = {}
        self._T = _transpose_of
        if not symmetric and build_transpose and _transpose_of is None:
            import scipy.sparse as sp

            At = sp.csr_matrix((values, indices, indptr), shape=(self.n_rows, self.n_cols)).T.tocsr()
            At.sort_indices()
            self._T = Graph(At.indptr, At.indices, At.data, self.n_cols, self.n_rows, device=self.device,
                            symmetric=False, exact_order=exact_order, split_threshold=split_threshold,
                            _transpose_of=self)

    @classmethod
    def from_torch_sparse(cls, t, symmetric=True, exact_order=False, split_threshold=0):
        """From the reference's own operand: a coalesced sparse COO (or CSR) fp32 tensor that lives on the device
        (models/LightGCN.py:31-32: convert_sp_mat_to_sp_tensor(...).coalesce().to(device)) — idg_graph_create_from_device.
        Non-symmetric graphs get no transposed handle here (build it from t.t().coalesce())."""
        if not t.is_cuda:
            raise RuntimeError("Graph.from_torch_sparse needs a device tensor; idgrec_amd has no CPU path")
        csr = t if t.layout == torch.sparse_csr else t.coalesce().to_sparse_csr()
        crow = csr.crow_indices().to(torch.int64).contiguous()
        col = csr.col_indices().to(torch.int32).contiguous()
        val = csr.values().to(torch.float32).contiguous()
        g = cls.__new__(cls)
        g.device, (g.n_rows, g.n_cols), g.nnz = t.device, (int(t.shape[0]), int(t.shape[1])), int(col.shape[0])
        g.symmetric, g._ws, g._T = bool(symmetric), {}, None
        flags = (native.IDG_GRAPH_SYMMETRIC if symmetric else 0) | (native.IDG_GRAPH_EXACT_ORDER if exact_order else 0)
        h = C.c_void_p()
        with torch.cuda.device(t.device):
            check(lib.idg_graph_create_from_device(t.device.index, g.n_rows, g.n_cols, g.nnz, _ptr(crow), _ptr(col), _ptr(val),
                                                   flags, int(split_threshold), _stream(), C.byref(h)),
                  "idg_graph_create_from_device")
        g._h = h
        return g

    @classmethod
    def from_scipy(cls, mat, **kw):
        m = mat.tocsr()
        m.sort_indices()
        return cls(m.indptr, m.indices, m.data, m.shape[0], m.shape[1], **kw)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # module globals are already gone at interpreter shutdown
            lib.idg_graph_destroy(h)

    @property
    def T(self):
        return self if self.symmetric else self._T

    # ---- live work units of a row bitmap (idg_graph_live_units): restricted launches naming the bitmap then run one
    #      wave per unit instead of visiting every tile
    def live_units(self, bitmap, max_rows, ws=None, stream=None):
        """Build (on `stream`, default: current) and register the unit list of `bitmap` (int32 tensor); returns the
        list's buffer, which the caller keeps alive while the bitmap is in use — and re-builds whenever the bitmap's
        contents change."""
        _require_device(bitmap, ws)
        if ws is None:
            ws = torch.empty(int(lib.idg_graph_live_units_bytes(self._h, int(max_rows))) // 4, dtype=torch.int32,
                             device=self.device)
        check(lib.idg_graph_live_units(self._h, _ptr(bitmap), _ptr(ws), int(max_rows), _stream() if stream is None else stream),
              "idg_graph_live_units")
        # the registration names two caller-owned buffers: it must not outlive either (the allocator may hand the same
        # address to something else).  One finalizer per tensor object, however often the list is rebuilt.
        # (the bitmap's finalizer forgets by bitmap, the list buffer's by buffer: a dying OLD buffer must not take a NEW
        #  registration of the same bitmap with it)
        if not getattr(bitmap, "_idg_units_finalizer", False):
            weakref.finalize(bitmap, _forget_units_of, weakref.ref(self), bitmap.data_ptr())
            bitmap._idg_units_finalizer = True
        if not getattr(ws, "_idg_units_finalizer", False):
            weakref.finalize(ws, _forget_units_ws, ws.data_ptr())
            ws._idg_units_finalizer = True
        return ws

    @staticmethod
    def live_units_check(ws, stream=None):
        """Raises if the list in `ws` could not hold every row of its bitmap (idg_graph_live_units_check; synchronises)."""
        check(lib.idg_graph_live_units_check(_ptr(ws), _stream() if stream is None else stream), "idg_graph_live_units_check")

    def compact_inputs(self, bitmap, ws=None, stream=None):
        """idg_graph_compact_inputs: the tiles' entry lists compacted to the live INPUT rows of `bitmap`, built on `stream`
        and registered for it; a product naming the bitmap as x_rows / gout mask then walks them.  Returns the buffer
        (keep it alive while the bitmap is in use; pass it back to rebuild in place)."""
        _require_device(bitmap, ws)
        if ws is None:
            ws = torch.empty(int(lib.idg_graph_compact_inputs_bytes(self._h)), dtype=torch.uint8, device=self.device)
        check(lib.idg_graph_compact_inputs(self._h, _ptr(bitmap), _ptr(ws), _stream() if stream is None else stream),
              "idg_graph_compact_inputs")
        # (the bitmap's finalizer forgets by bitmap, the list buffer's by buffer: a dying OLD buffer must not take a NEW
        #  registration of the same bitmap with it)
        if not getattr(bitmap, "_idg_units_finalizer", False):
            weakref.finalize(bitmap, _forget_units_of, weakref.ref(self), bitmap.data_ptr())
            bitmap._idg_units_finalizer = True
        if not getattr(ws, "_idg_units_finalizer", False):
            weakref.finalize(ws, _forget_units_ws, ws.data_ptr())
            ws._idg_units_finalizer = True
        return ws

    def bind_live_units(self, bitmap, ws, max_rows):
        """Register an existing list (built on a handle with the same schedule: the base of a masked / revalued copy)."""
        check(lib.idg_graph_bind_live_units(self._h, _ptr(bitmap), _ptr(ws), int(max_rows)), "idg_graph_bind_live_units")

    def forget_live_units(self, bitmap=None):
        if getattr(self, "_h", None):
            lib.idg_graph_forget_live_units(self._h, _ptr(bitmap))

    def revalued_copy(self, d_indptr, d_indices, d_values):
        """A Graph on this handle's structure and tile schedule with the values of the DEVICE CSR (d_indptr int64,
        d_indices int32, d_values fp32) — idg_graph_revalued_copy.  No host work: SGL's per-epoch edge-dropped views."""
        _require_device(d_indptr, d_indices, d_values)
        if d_indptr.dtype != torch.int64 or d_indices.dtype != torch.int32 or d_values.dtype != torch.float32:
            raise TypeError("revalued_copy: indptr int64, indices int32, values float32")
        g = Graph.__new__(Graph)
        g.device, g.n_rows, g.n_cols, g.nnz, g.symmetric = self.device, self.n_rows, self.n_cols, self.nnz, self.symmetric
        g._ws, g._base, g._T = self._ws, self, None
        h = C.c_void_p()
        check(lib.idg_graph_revalued_copy(self._h, _ptr(d_indptr), _ptr(d_indices), _ptr(d_values), _stream(), C.byref(h)),
              "idg_graph_revalued_copy")
        g._h = h
        return g

    def dropout_copy(self, keep_prob, stream=None, reuse=None):
        """NGCF.node_dropout (models/NGCF.py:56-65) of a SYMMETRIC graph as a new Graph on the same tile schedule:
        every stored entry is kept where int(u + (1 - keep_prob)) != 0, u ~ U[0, 1) — i.e. with probability
        1 - keep_prob, the reference's own (inverted-looking) rule — and divided by (1 - keep_prob); its `.T` (used by
        the backward of spmm) carries the transposed mask.  stream: (seed, stream id) of the draw; default: the next
        one of the device seed's sequence.  reuse: a Graph an earlier call on this handle returned — its two entry
        lists are redrawn in place (no allocation, nothing synchronous: the per-forward form)."""
        if not self.symmetric:
            raise ValueError("dropout_copy needs a symmetric graph (the transposed mask is read off the same structure)")
        seed, sid = _next_noise_stream() if stream is None else stream
        add = 1.0 - float(keep_prob)
        if reuse is not None and getattr(reuse, "_base", None) is self and reuse._T is not None:
            for g, transpose in ((reuse, 0), (reuse._T, 1)):
                check(lib.idg_graph_remask(self._h, g._h, add, add, C.c_uint64(seed), C.c_uint64(sid), transpose, _stream()),
                      "idg_graph_remask")
            return reuse
        out = []
        for transpose in (0, 1):
            g = Graph.__new__(Graph)
            g.device, g.n_rows, g.n_cols, g.nnz, g.symmetric = self.device, self.n_rows, self.n_cols, self.nnz, False
            g._ws, g._base = self._ws, self  # same workspaces (same schedule); keeps the base handle alive
            h = C.c_void_p()
            check(lib.idg_graph_masked_copy(self._h, add, add, C.c_uint64(seed), C.c_uint64(sid), transpose, _stream(),
                                            C.byref(h)), "idg_graph_masked_copy")
            g._h = h
            out.append(g)
        out[0]._T, out[1]._T = out[1], out[0]
        return out[0]

    def info(self):
        a = (C.c_int64 * 8)()
        check(lib.idg_graph_info(self._h, a), "idg_graph_info")
        keys = ("n_rows", "n_cols", "nnz", "n_tiles", "n_long_rows", "n_segments", "split_threshold", "flags")
        return dict(zip(keys, [int(x) for x in a]))

    def long_rows(self):
        n = self.info()["n_long_rows"]
        rows = np.empty(n, dtype=np.int64)
        seg = np.empty(n, dtype=np.int64)
        chunk = np.empty(n, dtype=np.int64)
        check(lib.idg_graph_long_rows(self._h, native.np_ptr(rows, C.c_int64), native.np_ptr(seg, C.c_int64),
                                      native.np_ptr(chunk, C.c_int64)), "idg_graph_long_rows")
        return rows, seg, chunk

    def _workspace(self, kind, d):
        key = (kind, int(d))
        ws = self._ws.get(key)
        if ws is None:
            fn = lib.idg_spmm_workspace_bytes if kind == "spmm" else lib.idg_propagate_workspace_bytes
            nbytes = int(fn(self._h, int(d)))
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    # ---- raw (non-autograd) calls
    def spmm_raw(self, X, addend=None, out=None):
        _require_device(X, addend, out)
        X = _f32c(X, "X")
        if X.dim() != 2 or X.shape[0] != self.n_cols:
            raise ValueError("X must be [%d, d], got %s" % (self.n_cols, tuple(X.shape)))
        d = X.shape[1]
        Y = torch.empty((self.n_rows, d), dtype=torch.float32, device=X.device) if out is None else out
        if addend is not None:
            addend = _f32c(addend, "addend")
            if addend.shape != Y.shape:
                raise ValueError("addend shape %s != output shape %s" % (tuple(addend.shape), tuple(Y.shape)))
        ws = self._workspace("spmm", d)
        check(lib.idg_spmm_f32(self._h, _ptr(X), d, _ptr(Y), d, _ptr(addend), d, _ptr(ws), _stream()), "idg_spmm_f32")
        return Y

    def propagate_mean_raw(self, E0, K, include_layer0=True, out=None, out_rows=None):
        """out_rows: int32 bitmap tensor of the rows of the result that will be read (others are left
        untouched), or None for the whole panel."""
        _require_device(E0, out, out_rows)
        E0 = _f32c(E0, "E0")
        if E0.shape[0] != self.n_rows:
            raise ValueError("E0 must have %d rows, got %d" % (self.n_rows, E0.shape[0]))
        d = E0.shape[1]
        out = torch.empty_like(E0) if out is None else out
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_f32(self._h, _ptr(E0), _ptr(out), _ptr(out_rows), int(K), int(bool(include_layer0)),
                                         d, _ptr(ws), _stream()), "idg_propagate_mean_f32")
        return out

    def propagate_mean_noise_raw(self, E0, K, include_layer0, eps, seed, stream_id, out=None, out_rows=None):
        """idg_propagate_mean_noise_f32 (SimGCL's perturbed encoder pass); out_rows as in propagate_mean_raw."""
        _require_device(E0, out, out_rows)
        E0 = _f32c(E0, "E0")
        d = E0.shape[1]
        out = torch.empty_like(E0) if out is None else out
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_noise_f32(self._h, _ptr(E0), _ptr(out), _ptr(out_rows), int(K), int(bool(include_layer0)),
                                               d, float(eps), C.c_uint64(seed), C.c_uint64(stream_id), _ptr(ws), _stream()),
              "idg_propagate_mean_noise_f32")
        return out

    def propagate_mean_bwd_raw(self, gout, K, include_layer0=True, out=None, accumulate=False, mask=None):
        """mask: int32/uint32 bitmap tensor of gout's live rows (see idg_propagate_mean_bwd_f32), or None."""
        _require_device(gout, out, mask)
        gout = _f32c(gout, "gout")
        d = gout.shape[1]
        if out is None:
            if accumulate:
                raise ValueError("accumulate=True needs an existing `out`")
            out = torch.empty_like(gout)
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_bwd_f32(self._h, _ptr(gout), _ptr(mask), _ptr(out), int(K), int(bool(include_layer0)), d,
                                             int(bool(accumulate)), _ptr(ws), _stream()),
              "idg_propagate_mean_bwd_f32")
        return out

    # ---- the batch's receptive field (idg_graph_expand_rows, idg_propagate_mean*_fields_f32)
    def expand_rows(self, in_rows, out_rows, stream=None):
        """out_rows = in_rows | neighbours of the rows flagged in in_rows (int32 bitmap tensors of n_rows bits)."""
        _require_device(in_rows, out_rows)
        check(lib.idg_graph_expand_rows(self._h, _ptr(in_rows), _ptr(out_rows), _stream() if stream is None else stream),
              "idg_graph_expand_rows")

    def flag_cols(self, in_rows, col_flags, stream=None):
        """col_flags[c] = 1.0 for the columns of the stored entries of the rows flagged in in_rows (float32 [n_cols],
        zeroed by the caller): idg_graph_flag_cols."""
        _require_device(in_rows, col_flags)
        check(lib.idg_graph_flag_cols(self._h, _ptr(in_rows), _ptr(col_flags), _stream() if stream is None else stream),
              "idg_graph_flag_cols")

    def mark_cols(self, in_rows, col_bits, stream=None):
        """col_bits |= the columns of the stored entries of the rows flagged in in_rows (int32 bitmap of n_cols bits, not
        cleared here): idg_graph_mark_cols."""
        _require_device(in_rows, col_bits)
        check(lib.idg_graph_mark_cols(self._h, _ptr(in_rows), _ptr(col_bits), _stream() if stream is None else stream),
              "idg_graph_mark_cols")

    @staticmethod
    def _bitmap_array(bitmaps):
        arr = (C.c_void_p * len(bitmaps))(*[None if b is None else b.data_ptr() for b in bitmaps])
        return arr

    def propagate_mean_fields_raw(self, E0, K, include_layer0, out, layer_rows):
        """idg_propagate_mean_fields_f32: layer k produces the rows of layer_rows[k-1] only (K bitmaps, the last = the
        batch's rows)."""
        _require_device(E0, out, *layer_rows)
        E0 = _f32c(E0, "E0")
        d = E0.shape[1]
        if len(layer_rows) != K:
            raise ValueError("layer_rows needs K bitmaps")
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_fields_f32(self._h, _ptr(E0), _ptr(out), self._bitmap_array(layer_rows), int(K),
                                                int(bool(include_layer0)), d, _ptr(ws), _stream()),
              "idg_propagate_mean_fields_f32")
        return out

    def propagate_mean_bwd_adam_fields_raw(self, gout, K, include_layer0, out, accumulate, step_rows, param, exp_avg,
                                           exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, discard_grad=False):
        """idg_propagate_mean_bwd_adam_fields_f32: step k's input is zero outside step_rows[k-1] (None = dense)."""
        _require_device(gout, out, param, exp_avg, exp_avg_sq, *[b for b in step_rows if b is not None])
        gout = _f32c(gout, "gout")
        d = gout.shape[1]
        if len(step_rows) != K or step_rows[0] is None:
            raise ValueError("step_rows needs K entries, the first a bitmap")
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_bwd_adam_fields_f32(self._h, _ptr(gout), self._bitmap_array(step_rows), _ptr(out), int(K),
                                                         int(bool(include_layer0)), d, int(bool(accumulate)) | (native.IDG_ADAM_DISCARD_GRAD if discard_grad else 0), _ptr(param),
                                                         _ptr(exp_avg), _ptr(exp_avg_sq), float(lr), float(beta1),
                                                         float(beta2), float(eps), int(step), _ptr(ws), _stream()),
              "idg_propagate_mean_bwd_adam_fields_f32")
        return out

    def propagate_mean_bwd_adam_raw(self, gout, K, include_layer0, out, accumulate, mask, param, exp_avg, exp_avg_sq, lr,
                                    step, beta1=0.9, beta2=0.999, eps=1e-8, discard_grad=False):
        """idg_propagate_mean_bwd_adam_f32: the backward above with the dense Adam step on `param` applied in the
        epilogue of its last product (bit-identical to propagate_mean_bwd_raw + adam_step_raw).  discard_grad: the
        finished gradient is not written back to `out` (IDG_ADAM_DISCARD_GRAD)."""
        _require_device(gout, out, mask, param, exp_avg, exp_avg_sq)
        gout = _f32c(gout, "gout")
        d = gout.shape[1]
        for t in (out, param, exp_avg, exp_avg_sq):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != gout.shape:
                raise TypeError("propagate_mean_bwd_adam_raw needs contiguous float32 panels shaped like gout")
        ws = self._workspace("prop", d)
        check(lib.idg_propagate_mean_bwd_adam_f32(self._h, _ptr(gout), _ptr(mask), _ptr(out), int(K), int(bool(include_layer0)),
                                                  d, int(bool(accumulate)) | (native.IDG_ADAM_DISCARD_GRAD if discard_grad else 0),
                                                  _ptr(param), _ptr(exp_avg), _ptr(exp_avg_sq),
                                                  float(lr), float(beta1), float(beta2), float(eps), int(step), _ptr(ws),
                                                  _stream()), "idg_propagate_mean_bwd_adam_f32")
        return out


def spmm_ex_raw(graph, X, Y=None, addend=None, sum_in=None, sum_out=None, div=1.0, accumulate=False, out_rows=None,
                x_rows=None):
    """idg_spmm_ex_f32: t = A.X (+ addend); Y = t; sum_out (+)= (sum_in + t) / div.  All [*, d] contiguous.
    out_rows / x_rows: optional int32 bitmaps of the output rows wanted / of the non-zero rows of X."""
    _require_device(X, Y, addend, sum_in, sum_out, out_rows, x_rows)
    d = X.shape[1]
    for t in (X, Y, addend, sum_in, sum_out):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1] != d):
            raise TypeError("spmm_ex_raw needs contiguous float32 [*, %d] panels" % d)
    ws = graph._workspace("spmm", d)
    check(lib.idg_spmm_ex_f32(graph._h, _ptr(X), d, _ptr(Y), _ptr(addend), _ptr(sum_in), _ptr(sum_out), d, float(div),
                              int(bool(accumulate)), _ptr(out_rows), _ptr(x_rows), d, _ptr(ws), _stream()), "idg_spmm_ex_f32")


def spmm_epi_raw(graph, X, Y=None, addend=None, sum_in=None, sum_in2=None, sum_in3=None, sum_out=None, div=1.0,
                 accumulate=False, mask=None, adam=None, out_rows=None, x_rows=None, adam_discard_grad=False, act=0,
                 act_src=None, act_rows=0):
    """idg_spmm_epi_f32: the product with every epilogue option (include/idgrec.h, idg_epilogue).  mask: bitmap of the
    live rows of addend / sum_in* / the accumulate target; adam = (param, exp_avg, exp_avg_sq, lr, step[, beta1, beta2,
    eps]): the Adam update of the rows of `param` with gradient sum_out in the same launch (adam_discard_grad: without
    writing that gradient).  out_rows and x_rows may be combined.  act: native.ACT_TANH (t = tanh(t)) or
    native.ACT_TANH_BWD (t *= 1 - act_src^2) applied to the rows below act_rows (0: all) before anything is stored."""
    _require_device(X, Y, addend, sum_in, sum_in2, sum_in3, sum_out, mask, out_rows, x_rows, act_src)
    d = X.shape[1]
    for t in (X, Y, addend, sum_in, sum_in2, sum_in3, sum_out):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1] != d):
            raise TypeError("spmm_epi_raw needs contiguous float32 [*, %d] panels" % d)
    e = native.Epilogue(_ptr(Y), _ptr(addend), _ptr(sum_in), _ptr(sum_in2), _ptr(sum_in3), _ptr(sum_out), d, float(div),
                        int(bool(accumulate)), _ptr(mask))
    if adam is not None:
        p, m, v, lr, step = adam[:5]
        b1, b2, eps = (tuple(adam[5:8]) + (0.9, 0.999, 1e-8)[len(adam) - 5:])[:3] if len(adam) > 5 else (0.9, 0.999, 1e-8)
        _require_device(p, m, v)
        e.adam_param, e.adam_exp_avg, e.adam_exp_avg_sq = _ptr(p), _ptr(m), _ptr(v)
        e.adam_lr, e.adam_beta1, e.adam_beta2, e.adam_eps, e.adam_step = float(lr), float(b1), float(b2), float(eps), int(step)
        e.adam_discard_grad = int(bool(adam_discard_grad))
    if act:
        e.act, e.act_src, e.act_rows = int(act), _ptr(act_src), int(act_rows)
    ws = graph._workspace("spmm", d)
    check(lib.idg_spmm_epi_f32(graph._h, _ptr(X), d, d, C.byref(e), _ptr(out_rows), _ptr(x_rows), _ptr(ws), _stream()),
          "idg_spmm_epi_f32")


_ssl_ws = {}  # InfoNCE workspaces by (n, B, d, device): the pair and cross forms of one step share one


def rows_tanh_bwd_raw(grad, y, rows, out):
    """out[r] = grad[r] * (1 - y[r]^2) at the rows of the bitmap `rows` (None: all) — idg_rows_tanh_bwd_f32."""
    _require_device(grad, y, rows, out)
    check(lib.idg_rows_tanh_bwd_f32(_ptr(grad), _ptr(y), _ptr(rows), int(grad.shape[0]), int(grad.shape[1]), _ptr(out), _stream()),
          "idg_rows_tanh_bwd_f32")


SSL_UNIQUE, SSL_RAW, SSL_CROSS = 0, 1, 2  # idg_infonce_plan modes


def infonce_workspace(n, B, d, device):
    """A workspace of idg_infonce_*_f32 for [n, d] panels and batches of B ids (one per batch in flight)."""
    return torch.empty(int(lib.idg_infonce_workspace_bytes(int(n), int(B), int(d))), dtype=torch.uint8, device=device)


def infonce_plan_raw(users, items, num_users, n, d, mode, ws, stream=None):
    """idg_infonce_plan: the id-list stage of an InfoNCE call (index-only work) into `ws`, on `stream` (raw handle; default:
    the current stream) — then infonce_pair_raw / infonce_cross_raw(..., ws=ws, planned=True) on the stream that has
    waited for it."""
    _require_device(users, items, ws)
    _require_ids(users, items)
    check(lib.idg_infonce_plan(_ptr(users), _ptr(items), users.shape[0], int(num_users), int(n), int(d), int(mode), _ptr(ws),
                               _stream() if stream is None else stream), "idg_infonce_plan")


def infonce_cross_raw(view, users, items, num_users, temperature, g=None, loss=None, grad_scale=1.0, ws=None, planned=False):
    """get_InfoNCE_loss(view[users], view[num_users + items], t) on the raw batch rows (models/EGCF.py:103), forward and
    backward: loss [1] and the gradient rows ADDED into g (idg_infonce_cross_f32).  Returns (loss, ws)."""
    _require_device(view, users, items, g, loss, ws)
    n, d = view.shape
    B = int(users.shape[0])
    if loss is None:
        loss = torch.zeros(2, dtype=torch.float32, device=view.device)
    if ws is None:
        key = (n, B, d, view.device)
        ws = _ssl_ws.get(key)
        if ws is None:
            ws = _ssl_ws[key] = torch.empty(int(lib.idg_infonce_workspace_bytes(n, B, d)), dtype=torch.uint8, device=view.device)
    check(lib.idg_infonce_cross_ex_f32(_ptr(view), n, d, _ptr(users), _ptr(items), B, int(num_users), float(temperature), _ptr(loss),
                                       _ptr(g), float(grad_scale), int(bool(planned)), _ptr(ws), _stream()), "idg_infonce_cross_f32")
    return loss, ws


def rows_gather2_raw(dst0, src0, dst1, src1, idx):
    """dstP[t] = srcP[idx[t]] (zeros where idx[t] < 0) for two panel pairs in one launch (idg_rows_gather2_f32)."""
    _require_device(dst0, src0, dst1, src1, idx)
    check(lib.idg_rows_gather2_f32(_ptr(dst0), _ptr(src0), _ptr(dst1), _ptr(src1), _ptr(idx), int(idx.shape[0]),
                                   int(dst0.shape[1]), _stream()), "idg_rows_gather2_f32")


def rows_scatter_raw(dst, idx, src):
    """dst[idx[j]] = src[j] (idg_rows_scatter_f32)."""
    _require_device(dst, idx, src)
    check(lib.idg_rows_scatter_f32(_ptr(dst), _ptr(idx), _ptr(src), int(src.shape[0]), int(dst.shape[1]), _stream()),
          "idg_rows_scatter_f32")


def rows_chain_store2_raw(dst0, src0, dst1, src1, idx, nxt):
    """dstP[idx[t]] = srcP[t] + srcP[nxt[t]] + ... for every head t, stored (idg_rows_chain_store2_f32)."""
    _require_device(dst0, src0, dst1, src1, idx, nxt)
    check(lib.idg_rows_chain_store2_f32(_ptr(dst0), _ptr(src0), _ptr(dst1), _ptr(src1), _ptr(idx), _ptr(nxt),
                                        int(idx.shape[0]), int(dst0.shape[1]), _stream()), "idg_rows_chain_store2_f32")


def rows_layer_mean_raw(out, ids, terms, last, div):
    """out[ids[j]] = (((a + b) + c)[ids[j]] + last[j]) / div, terms = up to three panels (idg_rows_layer_mean_f32); more
    terms (K > 3 layers), added left to right the same way: idg_rows_layer_mean_n_f32."""
    terms = [t for t in terms if t is not None]
    _require_device(out, ids, last, *terms)
    if len(terms) > 3:
        arr = (C.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
        check(lib.idg_rows_layer_mean_n_f32(_ptr(out), _ptr(ids), int(ids.shape[0]), arr, len(terms), _ptr(last), float(div),
                                            int(out.shape[1]), _stream()), "idg_rows_layer_mean_n_f32")
        return
    a, b, c = (terms + [None, None, None])[:3]
    check(lib.idg_rows_layer_mean_f32(_ptr(out), _ptr(ids), int(ids.shape[0]), _ptr(a), _ptr(b), _ptr(c), _ptr(last),
                                      float(div), int(out.shape[1]), _stream()), "idg_rows_layer_mean_f32")


def grad_tail_adam_raw(t, g, G, live_bits, row0, include_layer0, cnt, store_grad, param, exp_avg, exp_avg_sq, lr, step,
                       beta1=0.9, beta2=0.999, eps=1e-8):
    """idg_grad_tail_adam_f32 on a block of rows: every tensor is the block's [rows, d] slice of its panel."""
    _require_device(t, g, G, live_bits, param, exp_avg, exp_avg_sq)
    check(lib.idg_grad_tail_adam_f32(_ptr(t), _ptr(g), _ptr(G), _ptr(live_bits), int(row0), int(t.shape[0]), int(t.shape[1]),
                                     int(bool(include_layer0)), float(cnt), int(bool(store_grad)), _ptr(param),
                                     _ptr(exp_avg), _ptr(exp_avg_sq), float(lr), float(beta1), float(beta2), float(eps),
                                     int(step), _stream()), "idg_grad_tail_adam_f32")


def lincomb_raw(out, x, a, y=None, b=0.0):
    """out = a*x + b*y (y may be None)."""
    _require_device(out, x, y)
    check(lib.idg_lincomb_f32(_ptr(out), _ptr(x), float(a), _ptr(y), float(b), out.numel(), _stream()), "idg_lincomb_f32")


def subgraph_values_raw(row_of_entry, col_of_entry, edge_of_entry, kept_bits, dinv, out=None):
    """idg_subgraph_values_f32: the normalised values of an edge-dropped bipartite adjacency, in CSR entry order."""
    _require_device(row_of_entry, col_of_entry, edge_of_entry, kept_bits, dinv, out)
    nnz = row_of_entry.shape[0]
    out = torch.empty(nnz, dtype=torch.float32, device=dinv.device) if out is None else out
    check(lib.idg_subgraph_values_f32(nnz, _ptr(row_of_entry), _ptr(col_of_entry), _ptr(edge_of_entry), _ptr(kept_bits),
                                      _ptr(dinv), _ptr(out), _stream()), "idg_subgraph_values_f32")
    return out


def flags_compact_raw(flags, cap, ids=None, count=None, ws=None):
    """ids[0..cap) = ascending indices of the non-zero entries of the float vector `flags`, slots past the last one
    repeating it; count (int64 [1], optional) = their number.  No host synchronisation (idg_flags_compact_f32).
    Returns (ids, ws) — pass both back to reuse the buffers."""
    _require_device(flags, ids, count, ws)
    if flags.dtype != torch.float32 or not flags.is_contiguous() or flags.dim() != 1:
        raise TypeError("flags_compact_raw: contiguous float32 vector")
    n = int(flags.shape[0])
    if ids is None:
        ids = torch.empty(int(cap), dtype=torch.int64, device=flags.device)
    if ws is None:
        ws = torch.empty(int(lib.idg_flags_compact_workspace_bytes(n)), dtype=torch.uint8, device=flags.device)
    check(lib.idg_flags_compact_f32(_ptr(flags), n, _ptr(ids), int(cap), _ptr(count), _ptr(ws), _stream()),
          "idg_flags_compact_f32")
    return ids, ws


def rows_gather_raw(dst, src, idx):
    """idg_rows_gather_f32: dst[t] = src[idx[t]] (zeros where idx[t] < 0)."""
    _require_device(dst, src, idx)
    check(lib.idg_rows_gather_f32(_ptr(dst), _ptr(src), _ptr(idx), idx.shape[0], src.shape[1], _stream()), "idg_rows_gather_f32")


def rows_chain_add_raw(dst, src, idx, nxt):
    """idg_rows_chain_add_f32: dst[idx[t]] += src[t] + src[nxt[t]] + ... for every chain head t (idx[t] >= 0)."""
    _require_device(dst, src, idx, nxt)
    check(lib.idg_rows_chain_add_f32(_ptr(dst), _ptr(src), _ptr(idx), _ptr(nxt), idx.shape[0], src.shape[1], _stream()),
          "idg_rows_chain_add_f32")


_noise_stream = [0]


def _next_noise_stream():
    """(seed, stream id) of the next perturbation: the seed follows torch's device generator
    (tools.set_seed -> torch.cuda.manual_seed), the stream id counts perturbed layers drawn so far."""
    _noise_stream[0] += 1
    return int(torch.cuda.initial_seed()) & 0xFFFFFFFFFFFFFFFF, _noise_stream[0]


def reset_noise_stream(value=0):
    _noise_stream[0] = int(value)


class _SpMMNoise(torch.autograd.Function):
    """Y = A.X, Y += sign(Y) * normalize(U[0,1)) * eps.  d Y / d X = A (sign has zero gradient)."""

    @staticmethod
    def forward(ctx, X, graph, eps, stream):
        _require_device(X)
        X = _f32c(X, "X")
        ctx.graph = graph
        seed, stream_id = _next_noise_stream() if stream is None else stream
        return spmm_noise_raw(graph, X, eps, seed, stream_id)

    @staticmethod
    def backward(ctx, gY):
        return ctx.graph.T.spmm_raw(gY), None, None, None


def spmm_noise_raw(graph, X, eps, seed, stream_id, out=None, out_rows=None):
    """idg_spmm_noise_f32: one perturbed layer; out_rows: bitmap of the rows to produce."""
    _require_device(X, out, out_rows)
    d = X.shape[1]
    Y = torch.empty((graph.n_rows, d), dtype=torch.float32, device=X.device) if out is None else out
    ws = graph._workspace("spmm", d)
    check(lib.idg_spmm_noise_f32(graph._h, _ptr(X), d, _ptr(Y), d, _ptr(out_rows), d, float(eps), C.c_uint64(seed),
                                 C.c_uint64(stream_id), _ptr(ws), _stream()), "idg_spmm_noise_f32")
    return Y


def spmm_perturbed(graph, X, eps, stream=None):
    """One SimGCL/XSimGCL layer: torch.sparse.mm followed by the in-place noise (models/XSimGCL.py:51-54).
    stream: (seed, stream id) of the noise; default: the next one of the device seed's sequence."""
    return _SpMMNoise.apply(X, graph, float(eps), stream)


def layer_noise_stream(pass_stream, layer):
    """Noise stream of layer `layer` (1-based) of a perturbed K-layer pass drawn as ONE (seed, stream id) —
    what idg_propagate_mean_noise_f32 uses internally, so a layer can be re-produced on its own."""
    seed, sid = pass_stream
    return seed, sid * 64 + int(layer)


def perturb_raw(X, eps, seed, stream_id, out=None, rows=None):
    """idg_perturb_f32: out = X + sign(X) * normalize(u) * eps row by row (the epilogue's perturbation on its own)."""
    _require_device(X, out, rows)
    X = _f32c(X, "X")
    out = torch.empty_like(X) if out is None else out
    check(lib.idg_perturb_f32(_ptr(X), _ptr(out), X.shape[0], X.shape[1], _ptr(rows), float(eps), C.c_uint64(seed),
                              C.c_uint64(stream_id), _stream()), "idg_perturb_f32")
    return out


_compose_views = [False]  # tests: True = assemble the passes from the single-purpose entry points instead


def propagate_views_raw(graph, E0, K, include_layer0, eps, streams, outs, out_rows=None, scratch=None):
    """outs[0] = the clean layer mean of E0, outs[1 + i] = the perturbed one drawn from streams[i] = (seed, stream id)
    (SimGCL's encoder passes, models/SimGCL.py:63-65).  Without layer 0 in the mean and K >= 2 the first product,
    A.E0, is the same in every pass: it is computed ONCE, each view perturbs its own copy (sub-stream 0 of its
    stream) and all passes continue from layer 2 — mean(X1..XK) = propagate_mean(X1, K - 1, include_layer0=True).
    scratch: two [n, d] panels (allocated when None).  out_rows as in propagate_mean_raw."""
    d = E0.shape[1]
    shared = (not include_layer0) and K >= 2 and d in (32, 64, 128, 256, 512)
    if not shared:
        graph.propagate_mean_raw(E0, K, include_layer0, out=outs[0], out_rows=out_rows)
        for out, (seed, sid) in zip(outs[1:], streams):
            graph.propagate_mean_noise_raw(E0, K, include_layer0, eps, seed, sid, out=out, out_rows=out_rows)
        return outs
    if len(streams) <= 2 and not _compose_views[0]:
        # one C call: shared first product, per-view perturbation, and (with out_rows) ONE multi-panel launch for the
        # last layer of all passes — same values as the composition below
        n_views = len(streams)
        key = ("views", d, n_views)
        ws = graph._ws.get(key)
        if ws is None:
            ws = graph._ws[key] = torch.empty(int(lib.idg_propagate_views_workspace_bytes(graph._h, d, n_views)),
                                              dtype=torch.uint8, device=E0.device)
        seeds = (C.c_uint64 * n_views)(*[int(sd) for sd, _ in streams])
        sids = (C.c_uint64 * n_views)(*[int(si) for _, si in streams])
        views = (C.c_void_p * n_views)(*[o.data_ptr() for o in outs[1:]])
        _require_device(E0, out_rows, *outs)
        check(lib.idg_propagate_views_f32(graph._h, _ptr(_f32c(E0, "E0")), int(K), d, float(eps), n_views, seeds, sids,
                                          _ptr(outs[0]), views, _ptr(out_rows), _ptr(ws), _stream()),
              "idg_propagate_views_f32")
        return outs
    T, X1 = scratch if scratch is not None else (torch.empty_like(E0), torch.empty_like(E0))
    graph.spmm_raw(E0, out=T)
    graph.propagate_mean_raw(T, K - 1, True, out=outs[0], out_rows=out_rows)
    for out, (seed, sid) in zip(outs[1:], streams):
        perturb_raw(T, eps, seed, sid * 64, out=X1)  # layer 1 of this view; layers 2..K draw sub-streams 1..K-1
        graph.propagate_mean_noise_raw(X1, K - 1, True, eps, seed, sid, out=out, out_rows=out_rows)
    return outs


class _PropagateViews(torch.autograd.Function):
    """(clean, view_1, ..., view_n): the unperturbed layer mean and n independently perturbed ones.
    Every output has the same Jacobian w.r.t. E0 — (1/cnt) sum_k A^k — so backward adds the incoming
    gradients and propagates ONCE (autograd over separate passes would propagate n + 1 times)."""

    @staticmethod
    def forward(ctx, E0, graph, K, include_layer0, eps, n_views):
        _require_device(E0)
        E0 = _f32c(E0, "E0")
        ctx.graph, ctx.K, ctx.inc = graph, K, include_layer0
        outs = [torch.empty_like(E0) for _ in range(n_views + 1)]
        streams = [_next_noise_stream() for _ in range(n_views)]
        propagate_views_raw(graph, E0, K, include_layer0, eps, streams, outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        total = None
        for g in grads:
            if g is None:
                continue
            total = g if total is None else total + g
        if total is None:
            return None, None, None, None, None, None
        return ctx.graph.propagate_mean_bwd_raw(total.contiguous(), ctx.K, ctx.inc), None, None, None, None, None


def propagate_views(graph, E0, K, include_layer0, eps, n_views=2):
    """SimGCL's three encoder passes of one step (models/SimGCL.py:63-65) as one differentiable op."""
    return _PropagateViews.apply(E0, graph, int(K), bool(include_layer0), float(eps), int(n_views))


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, graph):
        ctx.graph = graph
        return graph.spmm_raw(X)

    @staticmethod
    def backward(ctx, gY):
        return ctx.graph.T.spmm_raw(gY), None


def spmm(graph, X):
    """Y = A.X — the drop-in for torch.sparse.mm(Graph, X) (models/LightGCN.py:44)."""
    return _SpMM.apply(X, graph)


class _PropagateMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E0, graph, K, include_layer0):
        ctx.graph, ctx.K, ctx.inc = graph, K, include_layer0
        return graph.propagate_mean_raw(E0, K, include_layer0)

    @staticmethod
    def backward(ctx, g):
        if not ctx.graph.symmetric:
            raise RuntimeError("propagate_mean backward needs a symmetric graph; chain spmm() instead")
        return ctx.graph.propagate_mean_bwd_raw(g, ctx.K, ctx.inc), None, None, None


def propagate_mean(graph, E0, K, include_layer0=True):
    """mean_k(A^k E0), k = (0|1)..K — LightGCN.aggregate / SimGCL.aggregate(perturbed=False)."""
    return _PropagateMean.apply(E0, graph, int(K), bool(include_layer0))


# ------------------------------------------------------------------------------------ BPR
_bpr_ws_cache = {}


class LocalEvent:
    """A device-local event (idg_event_*): orders two streams of one device without the system-scope fence of a default
    HIP / torch event (see include/idgrec.h).  record / wait take raw stream handles (torch.cuda.Stream.cuda_stream)."""

    __slots__ = ("_h", "_done")

    def __init__(self):
        import ctypes as C

        h = C.c_void_p()
        check(lib.idg_event_create(C.byref(h)), "idg_event_create")
        self._h, self._done = h, C.c_int(0)

    def record(self, stream):
        check(lib.idg_event_record(self._h, stream), "idg_event_record")

    def wait(self, stream):
        """Make `stream` wait for the recorded work."""
        check(lib.idg_stream_wait_event(stream, self._h), "idg_stream_wait_event")

    def synchronize(self):
        """Block the host until the recorded work has completed."""
        check(lib.idg_event_synchronize(self._h), "idg_event_synchronize")

    def query(self):
        import ctypes as C

        check(lib.idg_event_query(self._h, C.byref(self._done)), "idg_event_query")
        return bool(self._done.value)

    def __del__(self):
        try:
            lib.idg_event_destroy(self._h)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def side_stream(device):
    """A second stream for index-only work, bound to a hardware queue NOW.  HIP multiplexes a process's streams over a
    few hardware queues, binding each at its first use; a side stream first used after other streams have taken theirs
    (a communicator's) can land on the MAIN stream's queue, and its kernels then serialise with the step's (measured:
    +30 us per step).  Engines therefore create theirs at construction — before any communicator exists — and touch it
    once.  (A high-priority stream would have a queue class of its own, but waits between the two classes cost far
    more than they save: measured 0.33 -> 0.91 ms per step.)"""
    s = torch.cuda.Stream(device=device)
    with torch.cuda.stream(s):
        torch.zeros(1, dtype=torch.int32, device=device)
    s.synchronize()
    return s


def bpr_workspace(B, d, device):
    """A private BPR scratch buffer (callers that pipeline batches keep one per in-flight batch)."""
    return torch.empty(int(lib.idg_bpr_workspace_bytes(int(B), int(d))), dtype=torch.uint8, device=device)


def _bpr_ws(B, d, device):
    """Scratch of the fused BPR calls.  One buffer per (batch size, width, device): the forward and
    backward calls of a step (and the side-stream plan) must see the SAME buffer.  Callers that run
    two BPR steps concurrently on different streams need their own workspace (not done here)."""
    key = (int(B), int(d), device)
    ws = _bpr_ws_cache.get(key)
    if ws is None:
        ws = torch.empty(int(lib.idg_bpr_workspace_bytes(int(B), int(d))), dtype=torch.uint8, device=device)
        _bpr_ws_cache[key] = ws
    return ws


class _BprLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, final_panel, ego_panel, users, pos, neg, num_users, reg_lambda, deterministic):
        _require_device(final_panel, ego_panel, users, pos, neg)
        fin = _f32c(final_panel, "final_panel")
        ego = fin if ego_panel is final_panel else _f32c(ego_panel, "ego_panel")
        users, pos, neg = _i64c(users, "users"), _i64c(pos, "pos"), _i64c(neg, "neg")
        n, d = fin.shape
        B = users.shape[0]
        loss = torch.empty(2, dtype=torch.float32, device=fin.device)
        ws = _bpr_ws(B, d, fin.device)
        check(lib.idg_bpr_forward_f32(_ptr(fin), _ptr(ego), int(num_users), n, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                      float(reg_lambda), _ptr(loss), _ptr(ws), _stream()), "idg_bpr_forward_f32")
        ctx.save_for_backward(fin, ego, users, pos, neg)
        ctx.meta = (int(num_users), float(reg_lambda), bool(deterministic), ego_panel is final_panel, ws)
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g_bpr, g_reg):
        fin, ego, users, pos, neg = ctx.saved_tensors
        num_users, reg_lambda, deterministic, same, ws = ctx.meta
        n, d = fin.shape
        B = users.shape[0]
        up = torch.stack([g_bpr.to(torch.float32).reshape(()), g_reg.to(torch.float32).reshape(())]).contiguous()
        g_final = torch.zeros_like(fin)
        g_ego = g_final if same else torch.zeros_like(ego)
        # ws still holds this batch's coefficients: forward and backward of one step are adjacent on the stream
        check(lib.idg_bpr_backward_f32(_ptr(fin), _ptr(ego), num_users, n, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                       reg_lambda, _ptr(up), _ptr(g_final), _ptr(g_ego), int(deterministic), None,
                                       _ptr(ws), _stream()), "idg_bpr_backward_f32")
        return g_final, (None if same else g_ego), None, None, None, None, None, None


def bpr_loss(final_panel, ego_panel, users, pos, neg, num_users, reg_lambda, deterministic=True):
    """(bpr_loss, reg_lambda * reg_loss) as 0-d tensors, differentiable w.r.t. both panels.

    final_panel / ego_panel: [num_users + num_items, d], users first.  Passing the same
    tensor twice is the MFBPR case (models/MFBPR.py:29-42)."""
    if ego_panel is final_panel:
        out = _BprLossSame.apply(final_panel, users, pos, neg, num_users, reg_lambda, deterministic)
    else:
        out = _BprLoss.apply(final_panel, ego_panel, users, pos, neg, num_users, reg_lambda, deterministic)
    return out


class _BprLossSame(torch.autograd.Function):
    """final == ego: one differentiable input (autograd would otherwise see two aliases)."""

    @staticmethod
    def forward(ctx, panel, users, pos, neg, num_users, reg_lambda, deterministic):
        return _BprLoss.forward(ctx, panel, panel, users, pos, neg, num_users, reg_lambda, deterministic)

    @staticmethod
    def backward(ctx, g_bpr, g_reg):
        g = _BprLoss.backward(ctx, g_bpr, g_reg)
        return g[0], None, None, None, None, None, None


def bpr_touch_rows_raw(users, pos, neg, num_users, bitmap, stream=None, clear_bits=0):
    """Set the bits of the panel rows this batch touches (idg_bpr_touch_rows); bitmap: int32 [ceil(n/32)], zeroed
    by the caller or here when clear_bits = n.  stream: raw handle of the stream to launch on (default: current)."""
    _require_device(users, pos, neg, bitmap)
    _require_ids(users, pos, neg)
    st = _stream() if stream is None else stream
    if clear_bits:
        check(lib.idg_bitmap_clear(_ptr(bitmap), int(clear_bits), st), "idg_bitmap_clear")
    check(lib.idg_bpr_touch_rows(_ptr(users), _ptr(pos), _ptr(neg), users.shape[0], int(num_users), _ptr(bitmap), st),
          "idg_bpr_touch_rows")


def bitmap_clear_raw(bitmap, n_bits, stream=None):
    """idg_bitmap_clear: zero the first n_bits bits of an int32 bitmap tensor."""
    _require_device(bitmap)
    check(lib.idg_bitmap_clear(_ptr(bitmap), int(n_bits), _stream() if stream is None else stream), "idg_bitmap_clear")


def bpr_plan_raw(users, pos, neg, num_users, n, d, ws=None, stream=None):
    """Sort this batch's (row, slot) pairs into the BPR workspace (idg_bpr_plan_f32) on the CURRENT
    stream.  Index-only work: run it on a side stream while the propagation is in flight, then call
    bpr_fused_raw(..., deterministic=2)."""
    _require_device(users, pos, neg)
    _require_ids(users, pos, neg)
    B = users.shape[0]
    ws = _bpr_ws(B, d, users.device) if ws is None else ws
    check(lib.idg_bpr_plan_f32(_ptr(users), _ptr(pos), _ptr(neg), B, int(num_users), int(n), _ptr(ws),
                               _stream() if stream is None else stream), "idg_bpr_plan_f32")


def bpr_fused_raw(final_panel, ego_panel, users, pos, neg, num_users, reg_lambda, g_final, g_ego, loss=None,
                  deterministic=True, touched=None, ws=None):
    """No-autograd form: loss[2] plus gradients accumulated into g_final / g_ego (caller zeroes).
    deterministic: False/0 float atomics, True/1 sort in-call, 2 plan already built (bpr_plan_raw).
    touched: zeroed int32 bitmap [ceil(n/32)]; if given, reached g_final rows are stored + flagged and
    g_final needs no zero-fill (feed the bitmap to Graph.propagate_mean_bwd_raw(mask=...))."""
    _require_device(final_panel, ego_panel, users, pos, neg, g_final, g_ego)
    _require_ids(users, pos, neg)
    n, d = final_panel.shape
    B = users.shape[0]
    loss = torch.empty(2, dtype=torch.float32, device=final_panel.device) if loss is None else loss
    ws = _bpr_ws(B, d, final_panel.device) if ws is None else ws
    check(lib.idg_bpr_fused_f32(_ptr(final_panel), _ptr(ego_panel), int(num_users), n, _ptr(users), _ptr(pos),
                                _ptr(neg), B, d, float(reg_lambda), _ptr(loss), _ptr(g_final), _ptr(g_ego),
                                int(deterministic), _ptr(touched), _ptr(ws), _stream()), "idg_bpr_fused_f32")
    return loss


def bpr_rows_message_floats(B, d):
    """Length (fp32 words) of one rank's gradient-row message (idg_bpr_rows_message_floats)."""
    return int(lib.idg_bpr_rows_message_floats(int(B), int(d)))


def bpr_pack_rows_raw(ws, B, g_final, loss, message, clear=None, clear_bits=0):
    """After bpr_fused_raw(..., deterministic=2, touched=...): this batch's stored g_final rows, the plan's sorted row
    keys (ids and multiplicities) and loss[2] -> `message` (idg_bpr_pack_rows_f32).  clear / clear_bits: a bitmap to zero
    in the same launch (the union bitmap of the coming merge)."""
    _require_device(ws, g_final, loss, message)
    check(lib.idg_bpr_pack_rows_f32(_ptr(ws), int(B), g_final.shape[1], _ptr(g_final), _ptr(loss), _ptr(message),
                                    _ptr(clear), int(clear_bits), _stream()), "idg_bpr_pack_rows_f32")


def bpr_unpack_rows_raw(messages, world, B, ego_panel, reg_lambda, g_final, g_ego, touched, loss, touched_is_clear=False):
    """Merge `world` all-gathered messages in rank order into g_final / g_ego (averaged over ranks), the union
    bitmap `touched` (cleared first) and loss[2] (idg_bpr_unpack_rows_f32)."""
    _require_device(messages, ego_panel, g_final, g_ego, touched, loss)
    n, d = ego_panel.shape
    check(lib.idg_bpr_unpack_rows_f32(_ptr(messages), int(world), int(B), d, n, _ptr(ego_panel), float(reg_lambda),
                                      _ptr(g_final), _ptr(g_ego), _ptr(touched), int(bool(touched_is_clear)), _ptr(loss),
                                      _stream()),
          "idg_bpr_unpack_rows_f32")


def bpr_fwd_bwd_raw(final_panel, ego_panel, users, pos, neg, num_users, reg_lambda, upstream, g_final, g_ego, loss,
                    deterministic=True, ws=None):
    """Forward then backward with the two upstream gradient scalars read from the DEVICE tensor
    `upstream` (the sharded path scales a rank's share of the global batch this way).
    deterministic=2 with `ws` from bpr_plan_raw: the sorted scatter plan is already in the workspace."""
    _require_device(final_panel, ego_panel, users, pos, neg, upstream, g_final, g_ego, loss)
    n, d = final_panel.shape
    B = users.shape[0]
    if ws is None:
        if int(deterministic) == 2:
            raise ValueError("deterministic=2 (planned scatter) needs the workspace bpr_plan_raw filled")
        ws = _bpr_ws(B, d, final_panel.device)
    args = (_ptr(final_panel), _ptr(ego_panel), int(num_users), n, _ptr(users), _ptr(pos), _ptr(neg), B, d,
            float(reg_lambda))
    check(lib.idg_bpr_forward_f32(*args, _ptr(loss), _ptr(ws), _stream()), "idg_bpr_forward_f32")
    check(lib.idg_bpr_backward_f32(*args, _ptr(upstream), _ptr(g_final), _ptr(g_ego), int(deterministic), None,
                                   _ptr(ws), _stream()), "idg_bpr_backward_f32")
    return loss


# ----------------------------------------------------------------------------------- thin dense layer
_wgrad_ws = {}


def linear_wgrad_raw(X, G, out=None, accumulate=False):
    """idg_linear_wgrad_f32: out[d1, d2] (+)= X^T . G for tall X [n, d1], G [n, d2]."""
    _require_device(X, G, out)
    X, G = _f32c(X, "X"), _f32c(G, "G")
    n, d1 = X.shape
    d2 = G.shape[1]
    if G.shape[0] != n:
        raise ValueError("linear_wgrad_raw: X and G must have the same number of rows")
    if out is None:
        out = torch.empty((d1, d2), dtype=torch.float32, device=X.device)
    key = (n, d1, d2, X.device)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(int(lib.idg_linear_wgrad_workspace_bytes(n, d1, d2)), dtype=torch.uint8, device=X.device)
    check(lib.idg_linear_wgrad_f32(_ptr(X), d1, _ptr(G), d2, n, d1, d2, _ptr(out), int(bool(accumulate)), _ptr(ws), _stream()),
          "idg_linear_wgrad_f32")
    return out


_colsum_ws = {}


def colsum_raw(X, out=None, accumulate=False):
    """out[f] (+)= sum_r X[r, f] (idg_colsum_f32: row slices summed in slice order, deterministic)."""
    _require_device(X, out)
    X = _f32c(X, "X")
    n, d = X.shape
    if out is None:
        out = torch.empty(d, dtype=torch.float32, device=X.device)
    ws = _colsum_ws.get((d, X.device))
    if ws is None:
        ws = _colsum_ws[(d, X.device)] = torch.empty(int(lib.idg_colsum_workspace_bytes(d)), dtype=torch.uint8, device=X.device)
    check(lib.idg_colsum_f32(_ptr(X), d, n, d, _ptr(out), int(bool(accumulate)), _ptr(ws), _stream()), "idg_colsum_f32")
    return out


class _TallLinear(torch.autograd.Function):
    """Y = X @ W for X [n, d1] with n >> d1, d2 (NGCF's per-layer transforms).  Forward and the input gradient are
    ordinary small (NN) GEMMs; the weight gradient X^T @ gY is all reduction and goes to idg_linear_wgrad_f32."""

    @staticmethod
    def forward(ctx, X, W):
        ctx.save_for_backward(X, W)
        return torch.matmul(X, W)

    @staticmethod
    def backward(ctx, gY):
        X, W = ctx.saved_tensors
        # gY @ W^T as an NN GEMM on a transposed copy of the small weight: the NT form makes the BLAS library pick
        # a 200 us kernel for [n, 64] x [64, 64] (measured; the NN form of the same product takes 16 us)
        gX = torch.matmul(gY, W.t().contiguous()) if ctx.needs_input_grad[0] else None
        gW = linear_wgrad_raw(X, gY.contiguous()) if ctx.needs_input_grad[1] else None
        return gX, gW


def tall_linear(X, W):
    """torch.matmul(X, W) with a weight gradient sized for n >> d (models/NGCF.py:91-99)."""
    return _TallLinear.apply(X, W)


class _NgcfTransform(torch.autograd.Function):
    """S = side @ W1 + (ego * side) @ W2 on the fp32 matrix cores in one pass over the rows (idg_ngcf_transform_f32);
    backward: input gradients by idg_ngcf_transform_bwd_f32, weight gradients by idg_linear_wgrad_f32."""

    @staticmethod
    def forward(ctx, side, ego, W1, W2):
        _require_device(side, ego, W1, W2)
        side, ego, W1, W2 = _f32c(side, "side"), _f32c(ego, "ego"), _f32c(W1, "W1"), _f32c(W2, "W2")
        n, d1 = side.shape
        d2 = W1.shape[1]
        S = torch.empty((n, d2), dtype=torch.float32, device=side.device)
        need_w = ctx.needs_input_grad[3]
        BI = torch.empty_like(side) if need_w else None
        check(lib.idg_ngcf_transform_f32(_ptr(side), _ptr(ego), _ptr(W1), _ptr(W2), n, d1, d2, _ptr(S), _ptr(BI), _stream()),
              "idg_ngcf_transform_f32")
        ctx.save_for_backward(side, ego, W1, W2, BI if need_w else side)
        ctx.has_bi = need_w
        return S

    @staticmethod
    def backward(ctx, gS):
        side, ego, W1, W2, BI = ctx.saved_tensors
        gS = _f32c(gS, "gS")
        n, d1 = side.shape
        d2 = W1.shape[1]
        g_side = g_ego = gW1 = gW2 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g_side, g_ego = torch.empty_like(side), torch.empty_like(side)
            check(lib.idg_ngcf_transform_bwd_f32(_ptr(gS), _ptr(side), _ptr(ego), _ptr(W1), _ptr(W2), n, d1, d2, _ptr(g_side),
                                                 _ptr(g_ego), _stream()), "idg_ngcf_transform_bwd_f32")
        if ctx.needs_input_grad[2]:
            gW1 = linear_wgrad_raw(side, gS)
        if ctx.needs_input_grad[3]:
            gW2 = linear_wgrad_raw(BI if ctx.has_bi else ego * side, gS)
        return g_side, g_ego, gW1, gW2


def ngcf_transform(side, ego, W1, W2):
    """torch.matmul(side, W1) + torch.matmul(ego * side, W2) (models/NGCF.py:88-99, biases left to ngcf_layer_tail).
    MFMA kernel when the widths allow it (d1 % 64 == 0 and d2 % 64 == 0), the two thin GEMMs otherwise."""
    import os

    d1, d2 = W1.shape
    if d1 % 64 == 0 and d2 % 64 == 0 and side.is_cuda and os.environ.get("IDG_NGCF_MFMA", "1") != "0":
        return _NgcfTransform.apply(side, ego, W1, W2)
    return tall_linear(side, W1) + tall_linear(torch.mul(ego, side), W2)


class _NgcfTail(torch.autograd.Function):
    """(E, N) = NGCF's layer tail of (S1, S2, b1, b2): leaky_relu((S1 + b1) + (S2 + b2)) -> dropout -> (itself,
    its row-normalised copy) in one kernel; backward in one kernel + the bias column sums."""

    @staticmethod
    def forward(ctx, S1, S2, b1, b2, slope, p, stream):
        _require_device(S1, S2, b1, b2)
        S1, S2 = _f32c(S1, "S1"), (None if S2 is None else _f32c(S2, "S2"))
        n, d = S1.shape
        E, N = torch.empty_like(S1), torch.empty_like(S1)
        seed, sid = stream
        check(lib.idg_ngcf_tail_f32(_ptr(S1), _ptr(S2), _ptr(_f32c(b1.reshape(-1), "b1")), _ptr(_f32c(b2.reshape(-1), "b2")), n, d,
                                    float(slope), float(p), C.c_uint64(seed), C.c_uint64(sid), _ptr(E), _ptr(N), _stream()),
              "idg_ngcf_tail_f32")
        ctx.save_for_backward(E)
        ctx.args = (float(slope), float(p), seed, sid, b1.shape, b2.shape)
        return E, N

    @staticmethod
    def backward(ctx, gE, gN):
        (E,) = ctx.saved_tensors
        slope, p, seed, sid, shape1, shape2 = ctx.args
        n, d = E.shape
        gE = None if gE is None else _f32c(gE, "gE")
        gN = None if gN is None else _f32c(gN, "gN")
        gT = torch.empty_like(E)
        check(lib.idg_ngcf_tail_bwd_f32(_ptr(E), _ptr(gE), _ptr(gN), n, d, slope, p, C.c_uint64(seed), C.c_uint64(sid),
                                        _ptr(gT), _stream()), "idg_ngcf_tail_bwd_f32")
        gb = colsum_raw(gT)  # the two bias rows receive the same gradient: the column sums of gT
        return gT, (gT if ctx.needs_input_grad[1] else None), gb.reshape(shape1), gb.reshape(shape2), None, None, None


def ngcf_layer_tail(S1, S2, b1, b2, negative_slope=0.2, p=0.0, stream=None):
    """(next ego, its L2-normalised copy) of one NGCF layer from the two transformed panels (models/NGCF.py:95-108);
    S2 = None: S1 already holds their sum (ngcf_transform).
    stream: (seed, stream id) of the dropout mask; default: the next one of the device seed's sequence."""
    return _NgcfTail.apply(S1, S2, b1, b2, negative_slope, p, _next_noise_stream() if stream is None else stream)


# ----------------------------------------------------------------------------------- InfoNCE


def infonce_pair_raw(view1, view2, users, items, num_users, temperature, g1=None, g2=None, loss=None, dedup=True,
                     grad_scale=1.0, accumulate=False, ws=None, planned=False):
    """idg_infonce_pair_f32: loss[2] = InfoNCE over unique(users) rows and over num_users + unique(items) rows of
    the two [n, d] view panels; g1 / g2 (optional, pre-zeroed) receive d(loss[0] + loss[1]) / d view rows."""
    _require_device(view1, view2, users, items, g1, g2, loss)
    view1, view2 = _f32c(view1, "view1"), _f32c(view2, "view2")
    users, items = _i64c(users, "users"), _i64c(items, "items")
    n, d = view1.shape
    B = users.shape[0]
    if view2.shape != view1.shape or items.shape[0] != B:
        raise ValueError("infonce_pair_raw: view panels / id lists of different shapes")
    if ws is None:
        if planned:
            raise ValueError("infonce_pair_raw: planned=True needs the workspace idg_infonce_plan wrote into")
        key = (n, B, d, view1.device)
        ws = _ssl_ws.get(key)
        if ws is None:
            ws = _ssl_ws[key] = infonce_workspace(n, B, d, view1.device)
    if loss is None:
        loss = torch.empty(2, dtype=torch.float32, device=view1.device)
    check(lib.idg_infonce_pair_f32(_ptr(view1), _ptr(view2), n, d, _ptr(users), _ptr(items), B, int(num_users),
                                   int(bool(dedup)) | (native.IDG_SSL_PLANNED if planned else 0), float(temperature), _ptr(loss),
                                   _ptr(g1), _ptr(g2), float(grad_scale),
                                   int(bool(accumulate)), _ptr(ws), _stream()),
          "idg_infonce_pair_f32")
    return loss


class _InfoNCEPair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, view1, view2, users, items, num_users, temperature, dedup):
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g1 = torch.zeros_like(view1, memory_format=torch.contiguous_format) if need1 else None
        g2 = torch.zeros_like(view2, memory_format=torch.contiguous_format) if need2 else None
        loss = infonce_pair_raw(view1.detach(), view2.detach(), users, items, num_users, temperature, g1, g2, dedup=dedup)
        ctx.saved = (g1, g2)
        return loss.sum()

    @staticmethod
    def backward(ctx, grad_out):
        g1, g2 = ctx.saved
        return (None if g1 is None else g1.mul_(grad_out), None if g2 is None else g2.mul_(grad_out), None, None, None, None,
                None)


def infonce_pair(view1, view2, users, items, num_users, temperature, dedup=True):
    """get_InfoNCE_loss(view1[U-rows of unique(users)], view2[...]) + get_InfoNCE_loss(... unique(items) rows ...)
    — the self-supervised term of SimGCL / XSimGCL (models/SimGCL.py:79-84) — as one differentiable operator on
    the two [n, d] view panels (users first).  dedup=False: the id lists as they are, duplicates included (SGL)."""
    return _InfoNCEPair.apply(view1, view2, users, items, num_users, temperature, bool(dedup))


# ----------------------------------------------------------------------------------- Adam
def adam_step_raw(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    _require_device(param, grad, exp_avg, exp_avg_sq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("adam_step_raw needs contiguous float32 tensors")
    check(lib.idg_adam_step_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                                float(beta1), float(beta2), float(eps), int(step), _stream()), "idg_adam_step_f32")


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's default algorithm (utility/utility_train/trainer.py:11) as one HIP
    kernel per parameter tensor.  Same constructor signature for the arguments the reference
    uses; weight decay / amsgrad are not part of the reference path and are rejected."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("idgrec_amd.Adam implements the reference's plain Adam only")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                adam_step_raw(p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], st["step"], b1, b2, group["eps"])
        return loss


# --------------------------------------------------------------------------- scoring/top-K
def score_dense(user_panel, item_panel, users, apply_sigmoid=True):
    """rating[Bt, I] = act(user_panel[users] @ item_panel.T) — get_rating_for_test."""
    _require_device(user_panel, item_panel, users)
    U, V = _f32c(user_panel, "user_panel"), _f32c(item_panel, "item_panel")
    users = _i64c(users, "users")
    Bt, I, d = users.shape[0], V.shape[0], V.shape[1]
    rating = torch.empty((Bt, I), dtype=torch.float32, device=V.device)
    check(lib.idg_score_dense_f32(_ptr(U), _ptr(V), _ptr(users), Bt, I, d, int(bool(apply_sigmoid)), _ptr(rating),
                                  _stream()), "idg_score_dense_f32")
    return rating


def topk_option(name, value=None):
    """idg_score_topk_option: one of the process-wide knobs of score_topk's form choice — 'form' (-1 by geometry; 0 / 1 / 3),
    'collect' (0: never the threshold + collect form), 'floor', 'wgs', 'chunks', 'fallback_permille' (form 3 hands a call
    to the exact form beyond this share of users it cannot serve; < 0 never).  Returns the previous value; value None only
    reads; name 'reset' restores every default.  The library reads IDG_TOPK_* from the environment once, at first use."""
    from . import native

    if name == "reset":
        check(lib.idg_score_topk_option(native.IDG_TOPK_OPT_RESET, 0, None), "idg_score_topk_option")
        return None
    prev = C.c_int64()
    check(lib.idg_score_topk_option(native.IDG_TOPK_OPTS[name][0], native.IDG_TOPK_OPT_KEEP if value is None else int(value),
                                    C.byref(prev)), "idg_score_topk_option")
    return int(prev.value)


class topk_options:
    """with ops.topk_options(collect=0): ... — knobs of score_topk's form choice for the duration of a block."""

    def __init__(self, **values):
        self.values, self.saved = values, {}

    def __enter__(self):
        for name, value in self.values.items():
            self.saved[name] = topk_option(name, value)
        return self

    def __exit__(self, *exc):
        for name, value in self.saved.items():
            topk_option(name, value)
        return False


def score_topk_form(n_users, n_items, d, k):
    """Which kernel a score_topk call of this geometry takes (idg_score_topk_info without a workspace):
    {form: 0 alternating / 1 producer-consumer / 3 threshold + collect, chunks, floor}."""
    out = (C.c_int64 * 8)()
    check(lib.idg_score_topk_info(int(n_users), int(n_items), int(d), int(k), None, out, None), "idg_score_topk_info")
    return {"form": int(out[0]), "chunks": int(out[1]), "floor": bool(out[2])}


def score_topk(user_panel, item_panel, users, k, excl_indptr=None, excl_items=None, apply_sigmoid=True,
               return_values=False, info=None):
    """Top-k item ids per batch user, train positives masked to -1 (batch_test.py:59-68).
    excl_indptr int64[num_users+1] / excl_items int32: DEVICE CSR of the train matrix.
    info: a dict to fill with idg_score_topk_info's answer for this call (form, chunks, floor; form 3: users redone one by
    one, users the finish could not serve, whether the call fell back to the exact form as a whole; synchronises)."""
    _require_device(user_panel, item_panel, users, excl_indptr, excl_items)
    U, V = _f32c(user_panel, "user_panel"), _f32c(item_panel, "item_panel")
    users = _i64c(users, "users")
    Bt, I, d = users.shape[0], V.shape[0], V.shape[1]
    if excl_indptr is not None:
        if excl_indptr.dtype != torch.int64 or excl_items.dtype != torch.int32:
            raise TypeError("excl_indptr must be int64 and excl_items int32")
    idx = torch.empty((Bt, k), dtype=torch.int64, device=V.device)
    val = torch.empty((Bt, k), dtype=torch.float32, device=V.device) if return_values else None
    # One library call per `per_call` users.  A user's list does not depend on who shares its call, and every test user in
    # one call is what the kernels want — but the workspace grows with the call (the threshold + collect form keeps up to
    # 1024 candidate keys per user: 8-16 KB each), so a call whose workspace would exceed the budget (IDG_TOPK_WS_BYTES,
    # default 8 GiB: ~500,000 users) is cut into calls of a multiple of 16,384 users (256 user tiles: every CU keeps its own).
    budget = int(os.environ.get("IDG_TOPK_WS_BYTES", str(8 << 30)))
    per_call = Bt
    if Bt > 32768 and int(lib.idg_score_topk_workspace_bytes(Bt, I, d, int(k))) > budget:
        per_call = 16384
        while per_call * 2 < Bt and int(lib.idg_score_topk_workspace_bytes(per_call * 2, I, d, int(k))) <= budget:
            per_call *= 2
    redone = unserved = fell_back = 0
    cand_counts = []
    for s0 in range(0, Bt, per_call):
        n = min(per_call, Bt - s0)
        ws = torch.empty(int(lib.idg_score_topk_workspace_bytes(n, I, d, int(k))), dtype=torch.uint8, device=V.device)
        check(lib.idg_score_topk_f32(_ptr(U), _ptr(V), _ptr(users[s0:s0 + n]), n, I, d, _ptr(excl_indptr), _ptr(excl_items),
                                     int(k), int(bool(apply_sigmoid)), _ptr(idx[s0:s0 + n]),
                                     _ptr(val[s0:s0 + n]) if val is not None else None, _ptr(ws), _stream()),
              "idg_score_topk_f32")
        if info is not None:
            out = (C.c_int64 * 8)()
            check(lib.idg_score_topk_info(n, I, d, int(k), _ptr(ws), out, _stream()), "idg_score_topk_info")
            redone += max(int(out[3]), 0)
            unserved += max(int(out[4]), 0)
            fell_back += max(int(out[5]), 0)
            if s0 == 0:  # (form / chunks / floor of the first call: later ones differ only in a shorter last call)
                info.update(form=int(out[0]), chunks=int(out[1]), floor=bool(out[2]), calls=-(-Bt // per_call))
            info["users_redone"] = redone if int(out[3]) >= 0 else int(out[3])
            if int(out[3]) >= 0:  # (form 3)
                info.update(users_unserved=unserved, calls_fallen_back=fell_back, items_irregular=bool(out[6]))
                if info.get("candidates") is not None and int(out[5]) == 0:  # (asked for: info = {"candidates": True})
                    cnt = torch.empty(n, dtype=torch.int32, device=V.device)
                    check(lib.idg_score_topk_candidate_counts(n, I, d, int(k), _ptr(ws), _ptr(cnt), _stream()),
                          "idg_score_topk_candidate_counts")
                    cand_counts.append(cnt)
    if cand_counts:
        c = torch.cat(cand_counts).float()
        info["candidates"] = {"mean": float(c.mean()), "p50": float(c.quantile(0.5)) if c.numel() <= 16_000_000 else None,
                              "p99": float(c.quantile(0.99)) if c.numel() <= 16_000_000 else None, "max": float(c.max())}
    return (idx, val) if return_values else idx
