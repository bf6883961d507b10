"""Tensor-level front end of the C ABI: validates tensors the way the reference extension does with TORCH_CHECK
(device / dtype / contiguity -> RuntimeError), allocates outputs with torch (device memory stays owned by
PyTorch) and launches the HIP kernels on the current torch stream."""
import ctypes
import os
import threading
import weakref

import torch

from ._lib import check, lib


# Library GEMM dispatch: through torch a GEMM with a bias epilogue goes to hipBLASLt (descriptor set-up + heuristic query,
# ~19 us of host time per call), a plain mm to rocBLAS (~7 us) once rocBLAS is the preferred BLAS -- with ~130 GEMMs per pair
# that is the largest host item.  Callers that can fold the bias into the kernel consuming the product (add+LayerNorm,
# GroupNorm) therefore call linear(x, w) without bias.  Process-wide torch setting, set on import of this package.
_HAS_BLAS_SWITCH = hasattr(torch.backends.cuda, 'preferred_blas_library')
if _HAS_BLAS_SWITCH:
    torch.backends.cuda.preferred_blas_library('cublas')          # 'cublas' = rocBLAS on ROCm
_BLAS_SWITCH_LOCK = threading.Lock()
_BIG_GEMM_FLOP = 2.0e9     # above this the hipBLASLt kernels win (KPConv GEMMs: up to 2.7x faster than rocBLAS's choice)


def mm(a, b):
    """a (M, K) @ b (K, N) through the cheaper-to-dispatch rocBLAS path, or hipBLASLt where its kernels win: the large products
    with a row-major right operand (the KPConv (P*6, 36 Cin) @ (36 Cin, Cout) products) and K <= 32.  For the dense layers
    x @ W^T (transposed view of the (N, K) weight) rocBLAS's kernels are the faster ones at every size measured
    (tools/micro/unary_gemm_backends.py: 1.0-1.4x of the HBM / MFMA bound against 1.1-2.6x)."""
    big = 2.0 * a.shape[0] * a.shape[1] * b.shape[1] > _BIG_GEMM_FLOP
    if _HAS_BLAS_SWITCH and big and (b.stride(-1) == 1 or a.shape[1] <= 32):
        with _BLAS_SWITCH_LOCK:          # the preference is process-global: host threads of `--inflight 2` must not interleave the flips
            torch.backends.cuda.preferred_blas_library('cublaslt')
            try:
                return torch.mm(a, b)
            finally:
                torch.backends.cuda.preferred_blas_library('cublas')
    if _HAS_BLAS_SWITCH and threading.active_count() > 1:
        with _BLAS_SWITCH_LOCK:          # (a small product launched while another thread holds the flipped preference would take its path)
            return torch.mm(a, b)
    return torch.mm(a, b)


# ---- dense layers on the f16 matrix cores (csrc/linear_f16.hip) --------------------------------------------------------------
# 'auto': the hand-written f16 hi / lo split kernel for the inference GEMMs it wins on MI355X (tools/micro/linear_shapes.py, gpurun_out ->
# profiles/r03_linear_shapes.txt: in_features >= 256 and >= 20 000 rows: 1.1-1.55x the library's f32-MFMA-bound kernels; the short-K layers
# are HBM bound and stay on the library, which streams them better), the library otherwise; SE3_LINEAR=library forces the library everywhere.
LINEAR_F16 = os.environ.get('SE3_LINEAR', 'auto') != 'library'
LINEAR_F16_MIN_ROWS, LINEAR_F16_MIN_K = 20000, 256
_linear_piece_cache = {}          # (data_ptr, N, K, device) -> (weakref to the weight tensor, its version counter, pieces)


class _Shared:
    """Device tensors built asynchronously on one stream and read from others (weight pieces, index tables: caches shared by the host
    threads of `--inflight N`, one HIP stream each).  A reader on another stream waits -- on the GPU, not the host -- for the event recorded
    behind the kernels that fill them, and tells the caching allocator that its stream uses the memory too.  (Without this the second
    thread's first GEMM could read weight pieces the first thread's split kernel had not written yet.)"""
    __slots__ = ('tensors', 'stream', 'event', 'raw')

    def __init__(self, *tensors):
        self.tensors = tensors
        self.stream = torch.cuda.current_stream()
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        self.raw = self.stream.cuda_stream

    def get(self):
        raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        if raw is not None and raw(self.stream.device.index) == self.raw and torch.cuda.current_device() == self.stream.device.index:
            return self.tensors                              # the builder's own stream: nothing to wait for (and 2.6 us less host time)
        cur = torch.cuda.current_stream()
        if cur != self.stream:
            if not self.event.query():
                cur.wait_event(self.event)
            for t in self.tensors:
                t.record_stream(cur)
        return self.tensors


def _fingerprint(weight):
    """Device-side content fingerprint of a weight tensor (wrapping int64 sum of its bit patterns; one small launch when a cache entry is
    built, no host synchronisation): what validate_weight_caches() compares against the weight's current contents."""
    with torch.no_grad():
        return weight.detach().reshape(-1).view(torch.int32).sum(dtype=torch.int64)


def _linear_weight_pieces(weight, stream):
    """f16 hi / lo MFMA fragments of an (N, K) weight.  An entry is used only for the SAME Parameter object (weak reference compared by
    identity: a freed tensor's address and version can be inherited by another tensor) at the same `_version` (torch bumps it on every
    in-place update: optimizer steps, load_state_dict, copy_).  Writes that bypass the version counter (`p.data.copy_()`, `p.data = ...`,
    raw pointers) are invisible to it: SE3ET.load_state_dict / .to() / ._apply() clear the caches, and validate_weight_caches() compares
    every entry with the weight's current contents on the device.  A column / row block of a Parameter (`weight[:, :k]`: the decoder's
    split dense layer) is keyed by its own address, shape and row stride and owned by its BASE tensor (the view object is a temporary)."""
    N, K = weight.shape
    owner = weight._base if weight._base is not None else weight
    key = (weight.data_ptr(), N, K, weight.stride(0), weight.device.index)
    hit = _linear_piece_cache.get(key)
    if hit is not None and hit[0]() is owner and hit[1] == owner._version:
        return hit[2].get()[0]
    Wp = torch.empty((lib().se3_linear_weight_pieces_bytes(N, K),), dtype=torch.uint8, device=weight.device)
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    check(lib().se3_linear_split_weights_f16(w.data_ptr(), N, K, Wp.data_ptr(), stream), 'se3_linear_split_weights_f16')
    with _TIMING_LOCK:
        if len(_linear_piece_cache) > 512:
            _linear_piece_cache.clear()
        _linear_piece_cache[key] = (weakref.ref(owner), owner._version, _Shared(Wp), _fingerprint(weight), weight.stride(0))
    return Wp


_stacked_weight_cache = {}


def stacked_weight(weight, groups):
    """(N, K) -> (N, groups * K): the weight repeated along its input dimension, so that a dense layer applied to `groups` partial results
    laid side by side along the channels returns the layer of their SUM (cross_attention_eq_stack with key-anchor groups).  Kept per weight
    version like the f16 pieces."""
    key = (weight.data_ptr(), tuple(weight.shape), int(groups), weight.device.index)
    hit = _stacked_weight_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version:
        return hit[2].get()[0]
    with torch.no_grad():
        W = torch.cat([weight.detach()] * int(groups), 1).contiguous()
    with _TIMING_LOCK:
        if len(_stacked_weight_cache) > 128:
            _stacked_weight_cache.clear()
        _stacked_weight_cache[key] = (weakref.ref(weight), weight._version, _Shared(W), _fingerprint(weight))
    return W


def linear_f16_ok(x, weight):
    """True when linear_f16 applies: inference, f32 GPU tensors, unit-stride rows aligned to 16 bytes, enough rows to fill the chip."""
    if not LINEAR_F16 or torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()):
        return False
    K = weight.shape[1]
    rows = x.numel() // max(x.shape[-1], 1)
    return (x.shape[-1] == K and K % 32 == 0 and K >= LINEAR_F16_MIN_K and rows >= LINEAR_F16_MIN_ROWS and x.is_contiguous()
            and x.data_ptr() % 16 == 0)


def linear_f16(x, weight, bias=None, relu=False):
    """y = x W^T [+ bias] [ReLU] for x (..., K) contiguous, weight (N, K): csrc/linear_f16.hip."""
    stream = _stream()
    N, K = weight.shape
    rows = x.numel() // K
    Wp = _linear_weight_pieces(weight, stream)      # (the tensor object itself: the cache entry lives as long as it does)
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    b = None if bias is None else _req(bias.detach().contiguous(), torch.float32, 'bias', 1)
    check(lib().se3_linear_f16(x.data_ptr(), rows, K, K, Wp.data_ptr(), None if b is None else b.data_ptr(), N, 1 if relu else 0,
                               out.data_ptr(), N, stream), 'se3_linear_f16')
    return out


def dense_saturated_rows(reset=True):
    """Rows of the dense launches since the last reset whose values left the headroom of their f16-split scale (csrc/dense_norm.hip: a later
    input channel more than 2^8 times the row's first 32 values) or held NaN / Inf: clamped, counted.  One device synchronisation."""
    return int(lib().se3_debug_dense_saturated_rows(1 if reset else 0))


def attention_saturated(reset=True):
    """NaN / Inf values of K / V^T (and of the cross attention's q) that the operand splits of the f16 attention kernels clamped since the last
    reset (csrc/attention.hip: x6_split_kernel; finite values of any size are scaled, not clamped).  One device synchronisation."""
    return int(lib().se3_debug_attention_saturated(1 if reset else 0))


def linear_stream_ok(x, weight):
    """True when linear_stream applies: inference, f32 GPU tensors, unit-stride rows aligned to 16 bytes, in_features a multiple of 32."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and weight.stride(1) == 1):
        return False
    K = weight.shape[1]
    return x.shape[-1] == K and K % 32 == 0 and x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and (x.is_contiguous() or (x.dim() == 2 and x.stride(0) % 4 == 0))


def linear_stream(x, weight, bias=None, relu=False, out=None):
    """y = x W^T [+ bias] [ReLU] on the streaming f16-split kernel (csrc/dense_norm.hip, plain mode): any row count; x contiguous (..., K) or
    a 2-D view with a row stride; out: optional 2-D view (rows, N) with unit column stride to write into."""
    stream = _stream()
    N, K = weight.shape
    rows = x.numel() // K
    x_rs = x.stride(0) if (x.dim() == 2 and not x.is_contiguous()) else K
    Wp = _linear_weight_pieces(weight, stream)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        out_rs = N
    else:
        if out.dim() != 2 or out.shape[0] != rows or out.shape[1] != N or out.stride(1) != 1 or out.dtype != torch.float32:
            raise RuntimeError('linear_stream: out must be a (rows, N) float32 view with unit column stride')
        out_rs = out.stride(0)
    b = None if bias is None else _req(bias.detach().contiguous(), torch.float32, 'bias', 1)
    check(lib().se3_linear_stream(x.data_ptr(), rows, K, x_rs, Wp.data_ptr(), None if b is None else b.data_ptr(), N, 1 if relu else 0,
                                  out.data_ptr(), out_rs, stream), 'se3_linear_stream')
    return out


def linear_stream_transposed(x, weight, bias, ld=None):
    """x (A, R, K) contiguous -> (A, N, ld) with [a, :, :R] = (x[a] W^T + bias)^T: the value projection in the attention kernels' operand
    layout V^T, one launch for all anchors (columns R .. ld are left as allocated: callers pass ld = R padded rows)."""
    A, R, K = x.shape
    N = weight.shape[0]
    ld = R if ld is None else int(ld)
    stream = _stream()
    Wp = _linear_weight_pieces(weight, stream)
    out = torch.empty((A, N, ld), dtype=torch.float32, device=x.device)
    b = None if bias is None else _req(bias.detach().contiguous(), torch.float32, 'bias', 1)
    check(lib().se3_linear_stream_transposed(x.data_ptr(), A * R, K, K, Wp.data_ptr(), None if b is None else b.data_ptr(), N, R, out.data_ptr(), ld,
                                             stream), 'se3_linear_stream_transposed')
    return out


# ---- optional per-kernel timing with HIP events on the launch stream (enabled by bench.py) -----------------------------
TIMING_TAG = None             # optional tag (e.g. 'self') set by callers that want a separate bucket
KERNEL_TIMINGS = None          # dict name -> list of (start_event, end_event, algorithmic_bytes) while enabled
_TIMING_LOCK = threading.Lock()


class _timed:
    def __init__(self, name, nbytes, kind=None):
        self.name, self.nbytes, self.kind = name, nbytes, kind

    def __enter__(self):
        if KERNEL_TIMINGS is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if KERNEL_TIMINGS is not None:
            self.e1.record()
            KERNEL_TIMINGS.setdefault(self.name, []).append((self.e0, self.e1, self.nbytes))
            if self.kind is not None:
                KERNEL_TIMINGS.setdefault(self.name + '/' + self.kind, []).append((self.e0, self.e1, self.nbytes))
            if TIMING_TAG is not None:
                KERNEL_TIMINGS.setdefault(self.name + '@' + TIMING_TAG, []).append((self.e0, self.e1, self.nbytes))
        return False


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream (torch.cuda.current_stream() costs ~10 us of host time per call; the raw getter
    well under 1 us -- with ~150 launches per pair that is 1.5 ms of the host's 10 ms)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, name, ndim=None):
    if not torch.is_tensor(t):
        raise RuntimeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor (the SE3ET hot path has no CPU implementation)' % name)
    if t.dtype != dtype:
        raise RuntimeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError('%s must have %d dims, got %d' % (name, ndim, t.dim()))
    return t


def _host_lengths(lengths, name):
    """Batch lengths live on the host (as in the reference, where they are CPU LongTensors)."""
    if torch.is_tensor(lengths):
        if lengths.dtype != torch.int64:
            raise RuntimeError('%s must be int64' % name)
        lengths = lengths.tolist()
    arr = (ctypes.c_int64 * len(lengths))(*[int(v) for v in lengths])
    return arr, len(lengths)


GRID_SEARCH_MIN_SUPPORT = 1500        # below this the exhaustive kernel is faster than building a grid


class RadiusGrid:
    """Cell-binned support cloud for repeated radius searches with one radius (csrc/radius_neighbors.hip)."""

    def __init__(self, s_points, s_lengths, radius):
        self.s_points = _req(s_points, torch.float32, 's_points', 2)
        self.lengths, self.batch = _host_lengths(s_lengths, 's_lengths')
        self.radius = float(radius)
        self.ns = s_points.shape[0]
        nbytes = lib().se3_radius_grid_workspace_bytes(self.ns, self.batch)
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=s_points.device)
        check(lib().se3_radius_grid_build(self.s_points.data_ptr(), self.ns, self.lengths, self.batch, self.radius,
                                          self.ws.data_ptr(), nbytes, _stream()), 'se3_radius_grid_build')

    def search(self, q_points, q_lengths, limit, zeroed_max_count=None, ties=None):
        """zeroed_max_count: a (batch,) int32 device tensor the caller has already cleared (one fill for all searches of a pyramid).
        ties: None, or (tie_rows (>= Nq,) int32, tie_count (1,) int32 cleared by the caller) -- see radius_neighbors."""
        _req(q_points, torch.float32, 'q_points', 2)
        ql, nb = _host_lengths(q_lengths, 'q_lengths')
        if nb != self.batch:
            raise RuntimeError('q_lengths and s_lengths differ in batch size')
        nq = q_points.shape[0]
        out = torch.empty((nq, limit), dtype=torch.int64, device=q_points.device)
        if zeroed_max_count is not None:
            max_count = _req(zeroed_max_count, torch.int32, 'zeroed_max_count', 1)
            if max_count.shape[0] != nb or not max_count.is_contiguous():
                raise RuntimeError('radius search: zeroed_max_count must be a contiguous (batch,) int32 tensor')
        else:
            max_count = torch.empty((nb,), dtype=torch.int32, device=q_points.device)
        tr, tc = _tie_buffers(ties, nq)
        check(lib().se3_radius_neighbors_grid_ties(q_points.data_ptr(), nq, ql, self.lengths, self.ns, self.batch, self.ws.data_ptr(),
                                                   self.radius, int(limit), out.data_ptr(), max_count.data_ptr(),
                                                   1 if zeroed_max_count is not None else 0, tr, tc, _stream()),
              'se3_radius_neighbors_grid')
        return out, max_count


def _tie_buffers(ties, nq):
    if ties is None:
        return None, None
    rows, count = ties
    _req(rows, torch.int32, 'tie_rows', 1), _req(count, torch.int32, 'tie_count', 1)
    if rows.shape[0] < nq or count.shape[0] != 1:
        raise RuntimeError('radius search: tie_rows needs room for every query row, tie_count is ONE int32 word')
    return rows.data_ptr(), count.data_ptr()


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, limit, grid=None, zeroed_max_count=None, ties=None):
    """Returns (neighbors (Nq, limit) int64 padded with Ns, max_count (batch,) int32 device tensor: per cloud the largest
    in-radius count).  Large supports go through
    a uniform grid (pass a prebuilt RadiusGrid to share it between searches); results are identical either way.
    Exactly tied distances come in index order; `ties` = (tie_rows (>= Nq,) int32, tie_count (1,) int32, CLEARED by the caller) makes the
    search list the rows whose kept columns hold such a tie (radius_tie_order then gives them the reference's order)."""
    _req(q_points, torch.float32, 'q_points', 2)
    _req(s_points, torch.float32, 's_points', 2)
    if grid is None and s_points.shape[0] >= GRID_SEARCH_MIN_SUPPORT:
        grid = RadiusGrid(s_points, s_lengths, radius)
    if grid is not None:
        if grid.s_points.data_ptr() != s_points.data_ptr() or grid.radius != float(radius):
            raise RuntimeError('radius_neighbors: the grid was built for another support cloud / radius')
        return grid.search(q_points, q_lengths, limit, zeroed_max_count, ties)
    ql, nb = _host_lengths(q_lengths, 'q_lengths')
    sl, nb2 = _host_lengths(s_lengths, 's_lengths')
    if nb != nb2:
        raise RuntimeError('q_lengths and s_lengths differ in batch size')
    nq, ns = q_points.shape[0], s_points.shape[0]
    out = torch.empty((nq, limit), dtype=torch.int64, device=q_points.device)
    max_count = torch.empty((nb,), dtype=torch.int32, device=q_points.device)
    tr, tc = _tie_buffers(ties, nq)
    check(lib().se3_radius_neighbors_ties(q_points.data_ptr(), nq, s_points.data_ptr(), ns, ql, sl, nb, float(radius),
                                          int(limit), out.data_ptr(), max_count.data_ptr(), tr, tc, _stream()),
          'se3_radius_neighbors')
    return out, max_count


# The reference's order of exactly tied distances (csrc/radius_ties.hip).  On by default: clouds without exact ties
# (every jittered synthetic configuration) pay one ballot per row in the search kernels and nothing else.
RADIUS_REFERENCE_TIES = os.environ.get('SE3_RADIUS_REFERENCE_TIES', '1') != '0'       # (0: the kernels' index order; A/B runs)


class ReferenceTree:
    """The reference's k-d tree (nanoflann, leaf size 10) of the support clouds, built on the HOST from a host copy of the points
    (se3_kdtree_build_host: structural preprocessing, 12 + ~6 bytes per point) and uploaded; walked on the GPU by radius_tie_order."""

    def __init__(self, s_points, s_lengths):
        self.s_points = _req(s_points, torch.float32, 's_points', 2)
        self.lengths, self.batch = _host_lengths(s_lengths, 's_lengths')
        self.ns = s_points.shape[0]
        host_pts = s_points.cpu()                                              # (a synchronisation: only ever reached when rows were flagged)
        cap = lib().se3_kdtree_max_bytes(self.ns, self.batch)
        host = torch.empty((cap,), dtype=torch.uint8)                          # (pageable: pinning a fresh buffer per tree costs more than the copy)
        used = ctypes.c_size_t(0)
        check(lib().se3_kdtree_build_host(host_pts.data_ptr(), self.ns, self.lengths, self.batch, host.data_ptr(), cap, ctypes.byref(used)),
              'se3_kdtree_build_host')
        self.tree = host[:int(used.value)].to(s_points.device)


def radius_tie_order(neighbors, q_points, s_points, q_lengths, s_lengths, radius, tie_rows, num_tie_rows, max_hits, tree=None):
    """Rewrites the `num_tie_rows` rows listed in tie_rows (device int32; what a search with `ties` flagged) of the (Nq, limit) table
    `neighbors` IN PLACE with the reference's order / choice of exactly tied distances.  max_hits: the search's largest in-radius count
    (host int).  tree: a ReferenceTree of the same support clouds (built when omitted).  Returns the tree."""
    if num_tie_rows <= 0:
        return tree
    _req(neighbors, torch.int64, 'neighbors', 2)
    tree = tree if tree is not None else ReferenceTree(s_points, s_lengths)
    if tree.s_points.data_ptr() != s_points.data_ptr():
        raise RuntimeError('radius_tie_order: the tree was built for another support cloud')
    ql, nb = _host_lengths(q_lengths, 'q_lengths')
    nbytes = lib().se3_radius_tie_scratch_bytes(int(num_tie_rows), int(max_hits))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=q_points.device)
    check(lib().se3_radius_neighbors_tie_order(q_points.data_ptr(), q_points.shape[0], s_points.data_ptr(), s_points.shape[0], ql, tree.lengths,
                                               nb, tree.tree.data_ptr(), float(radius), neighbors.shape[1], tie_rows.data_ptr(),
                                               int(num_tie_rows), int(max_hits), scratch.data_ptr(), nbytes, neighbors.data_ptr(), _stream()),
          'se3_radius_neighbors_tie_order')
    # the last 256 bytes of the scratch hold the number of rows that did not fit (more matches than max_hits, a tree deeper than the walk's
    # stack): they keep their index order -- callers read the word with their next synchronisation and fail loudly (tie_overflow_check)
    if not hasattr(_tie_tls, 'words'):
        _tie_tls.words = []
    _tie_tls.words.append(scratch[nbytes - 256:nbytes - 252].view(torch.int32))
    return tree


_tie_tls = threading.local()          # per host thread (= per stream): a thread only ever checks the passes it issued itself


def tie_overflow_check():
    """Raises if any row of the tie passes issued since the last check was left in index order (one small device -> host copy; a no-op when
    no tie pass ran)."""
    words = getattr(_tie_tls, 'words', None)
    if not words:
        return
    _tie_tls.words = []
    left = int(torch.cat(words).sum())
    if left:
        raise RuntimeError('radius search: %d row(s) with exactly tied distances could not be given the reference order (more matches than the '
                           'search counted, or a k-d tree deeper than %d levels)' % (left, 48))


def radius_search_reference_order(q_points, s_points, q_lengths, s_lengths, radius, limit):
    """One search with the reference's tie order: (neighbors (Nq, limit), largest in-radius count (host int)).  Two synchronisations when
    rows were flagged (the counts; the host copy of the support points for the tree), one otherwise."""
    nq = q_points.shape[0]
    flags = torch.zeros((nq + 1,), dtype=torch.int32, device=q_points.device) if RADIUS_REFERENCE_TIES else None
    ties = (flags[1:], flags[:1]) if flags is not None else None
    full, max_count = radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, limit, ties=ties)
    host = torch.cat((max_count.max().reshape(1), flags[:1])).tolist() if flags is not None else [int(max_count.max()), 0]
    if host[1] > 0:
        radius_tie_order(full, q_points, s_points, q_lengths, s_lengths, radius, flags[1:], host[1], max(host[0], 1))
        tie_overflow_check()
    return full, host[0]


def grid_subsample(points, lengths, normals, voxel_size):
    """Returns (s_points (N,3) [first sum(s_lengths) rows valid], s_normals or None, s_lengths (B,) int64 device).  `lengths` on the
    host: they sum to the rows of `points`.  `lengths` a DEVICE int64 tensor (the s_lengths of a previous call): no host synchronisation
    -- the rows of `points` are then an upper bound of their sum (se3_grid_subsample_dev)."""
    _req(points, torch.float32, 'points', 2)
    if normals is not None:
        _req(normals, torch.float32, 'normals', 2)
    if torch.is_tensor(lengths) and lengths.is_cuda:
        ld = _req(lengths.contiguous(), torch.int64, 'lengths', 1)
        n, nb, dev = points.shape[0], ld.shape[0], points.device
        ws_bytes = lib().se3_grid_subsample_workspace_bytes(n, nb)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        s_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
        s_normals = torch.empty((n, 3), dtype=torch.float32, device=dev) if normals is not None else None
        s_lengths = torch.empty((nb,), dtype=torch.int64, device=dev)
        check(lib().se3_grid_subsample_dev(points.data_ptr(), normals.data_ptr() if normals is not None else None, n, ld.data_ptr(), nb,
                                           float(voxel_size), s_points.data_ptr(), s_normals.data_ptr() if s_normals is not None else None,
                                           s_lengths.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), 'se3_grid_subsample_dev')
        return s_points, s_normals, s_lengths
    ln, nb = _host_lengths(lengths, 'lengths')
    n = points.shape[0]
    dev = points.device
    ws_bytes = lib().se3_grid_subsample_workspace_bytes(n, nb)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    s_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
    s_normals = torch.empty((n, 3), dtype=torch.float32, device=dev) if normals is not None else None
    s_lengths = torch.empty((nb,), dtype=torch.int64, device=dev)
    check(lib().se3_grid_subsample(points.data_ptr(), normals.data_ptr() if normals is not None else None, n, ln, nb,
                                   float(voxel_size), s_points.data_ptr(),
                                   s_normals.data_ptr() if s_normals is not None else None, s_lengths.data_ptr(),
                                   ws.data_ptr(), ws_bytes, _stream()), 'se3_grid_subsample')
    return s_points, s_normals, s_lengths


# =====================================================================================================================
# Ops whose gfx950 kernel is not bound yet run as eager torch-on-GPU chains (never on the CPU); each entry of
# INTERIM_TORCH is replaced by a C-ABI call as its kernel lands (DESIGN.md tracks the status per SURVEY row).
# =====================================================================================================================
import math

import torch.nn.functional as F

INTERIM_TORCH = set()


def _interim(fn):
    INTERIM_TORCH.add(fn.__name__)
    return fn


def _gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor (the SE3ET hot path has no CPU implementation)' % name)
    return t


def _split_heads(x, h):
    return x.reshape(*x.shape[:-1], h, -1).transpose(-2, -3)


def _merge_heads(x):
    x = x.transpose(-2, -3)
    return x.reshape(*x.shape[:-2], -1)


def log_optimal_transport(scores, row_masks, col_masks, alpha, num_iterations, inf):
    """HIP: one workgroup per patch pair, score matrix in registers for all iterations (csrc/sinkhorn.hip)."""
    scores = _req(scores.contiguous(), torch.float32, 'scores', 3)
    B, R, C = scores.shape
    rm = _req(row_masks.to(torch.uint8).contiguous(), torch.uint8, 'row_masks', 2)
    cm = _req(col_masks.to(torch.uint8).contiguous(), torch.uint8, 'col_masks', 2)
    al = _req(alpha.detach().reshape(1).contiguous(), torch.float32, 'alpha')
    out = torch.empty((B, R + 1, C + 1), dtype=torch.float32, device=scores.device)
    check(lib().se3_log_sinkhorn_fwd(scores.data_ptr(), rm.data_ptr(), cm.data_ptr(), al.data_ptr(), B, R, C,
                                     int(num_iterations), float(inf), out.data_ptr(), _stream()), 'se3_log_sinkhorn_fwd')
    return out


def log_optimal_transport_bwd(grad_out, scores, row_masks, col_masks, alpha, num_iterations, inf):
    """HIP (csrc/sinkhorn.hip): gradients of log_optimal_transport with respect to (scores, alpha)."""
    scores = _req(scores.contiguous(), torch.float32, 'scores', 3)
    B, R, C = scores.shape
    g = _req(grad_out.contiguous(), torch.float32, 'grad_out', 3)
    rm = _req(row_masks.to(torch.uint8).contiguous(), torch.uint8, 'row_masks', 2)
    cm = _req(col_masks.to(torch.uint8).contiguous(), torch.uint8, 'col_masks', 2)
    al = _req(alpha.detach().reshape(1).contiguous(), torch.float32, 'alpha')
    ds = torch.empty_like(scores)
    da = torch.empty((B,), dtype=torch.float32, device=scores.device)
    check(lib().se3_log_sinkhorn_bwd(scores.data_ptr(), rm.data_ptr(), cm.data_ptr(), al.data_ptr(), g.data_ptr(), B, R, C,
                                     int(num_iterations), float(inf), ds.data_ptr(), da.data_ptr(), _stream()), 'se3_log_sinkhorn_bwd')
    return ds, da.sum().reshape(alpha.shape)


def pairwise_distance(x, y, normalized=False):
    """HIP (csrc/pairwise_distance.hip): squared distances (*, N, M) of x (*, N, C) and y (*, M, C), clamped at 0."""
    x = _req(x.contiguous(), torch.float32, 'x')
    y = _req(y.contiguous(), torch.float32, 'y')
    if x.dim() < 2 or x.dim() != y.dim() or x.shape[:-2] != y.shape[:-2] or x.shape[-1] != y.shape[-1]:
        raise RuntimeError('pairwise_distance: shapes %s and %s' % (tuple(x.shape), tuple(y.shape)))
    N, M, C = x.shape[-2], y.shape[-2], x.shape[-1]
    batch = x.numel() // max(N * C, 1) if N * C else 0
    out = torch.empty(x.shape[:-2] + (N, M), dtype=torch.float32, device=x.device)
    if C == 0:
        return out.zero_()
    check(lib().se3_pairwise_distance(x.data_ptr(), y.data_ptr(), batch, N, M, C, N * C, M * C, 1 if normalized else 0, out.data_ptr(), _stream()),
          'se3_pairwise_distance')
    return out


def add_layer_norm(hidden, residual, weight, bias, eps, hidden_bias=None):
    """HIP (csrc/rowops.hip): LayerNorm(hidden [+ hidden_bias] + residual); residual may lack leading (anchor) dims of hidden;
    hidden_bias (C,) is the bias of the linear layer that produced hidden (its GEMM then runs bias-free)."""
    hidden = _req(hidden.contiguous(), torch.float32, 'hidden')
    C = hidden.shape[-1]
    if residual.shape != hidden.shape:
        # supported broadcast: one (N, C) residual block shared by all leading (anchor) slices of hidden
        if tuple(residual.shape[-2:]) != tuple(hidden.shape[-2:]) or residual.numel() != hidden.shape[-2] * C:
            raise RuntimeError('add_layer_norm: unsupported residual broadcast %s vs %s'
                               % (tuple(residual.shape), tuple(hidden.shape)))
    residual = _req(residual.contiguous(), torch.float32, 'residual')
    rows, res_rows = hidden.numel() // C, residual.numel() // C
    out = torch.empty_like(hidden)
    check(lib().se3_add_layer_norm_fwd(hidden.data_ptr(), hidden_bias.data_ptr() if hidden_bias is not None else None,
                                       residual.data_ptr(), weight.data_ptr(), bias.data_ptr(), rows, res_rows, C, float(eps),
                                       out.data_ptr(), _stream()), 'se3_add_layer_norm_fwd')
    return out


def add_layer_norm_bwd(grad_out, hidden, residual, weight, eps, hidden_bias=None):
    """HIP (csrc/rowops.hip): gradients of add_layer_norm -> (d hidden, d residual, d weight, d bias, d hidden_bias or None)."""
    hidden = _req(hidden.contiguous(), torch.float32, 'hidden')
    g = _req(grad_out.contiguous(), torch.float32, 'grad_out')
    C = hidden.shape[-1]
    res = _req(residual.contiguous(), torch.float32, 'residual')
    rows, res_rows = hidden.numel() // C, res.numel() // C
    dh = torch.empty_like(hidden)
    if TRAINING_DETERMINISTIC:
        # per-workgroup partial sums added in order by one torch reduction (no float atomics: bit-identical runs)
        part = torch.empty((int(lib().se3_add_layer_norm_bwd_blocks(rows)), 3, C), dtype=torch.float32, device=hidden.device)
        check(lib().se3_add_layer_norm_bwd_partials(hidden.data_ptr(), hidden_bias.data_ptr() if hidden_bias is not None else None, res.data_ptr(),
                                                    weight.data_ptr(), g.data_ptr(), rows, res_rows, C, float(eps), dh.data_ptr(), part.data_ptr(),
                                                    _stream()), 'se3_add_layer_norm_bwd_partials')
        params = part.sum(0)
    else:
        params = torch.zeros((3, C), dtype=torch.float32, device=hidden.device)
        check(lib().se3_add_layer_norm_bwd(hidden.data_ptr(), hidden_bias.data_ptr() if hidden_bias is not None else None, res.data_ptr(),
                                           weight.data_ptr(), g.data_ptr(), rows, res_rows, C, float(eps), dh.data_ptr(), params.data_ptr(),
                                           _stream()), 'se3_add_layer_norm_bwd')
    dres = dh if res_rows == rows else dh.reshape(rows // res_rows, res_rows, C).sum(0)
    return dh, dres.reshape(residual.shape), params[0], params[1], (params[2] if hidden_bias is not None else None)


def gather_rows_padded(x, idx):
    """HIP: x[idx] with idx == x.shape[0] addressing an all-zero row; idx of any shape."""
    x = _req(x.contiguous(), torch.float32, 'x')
    idx = _req(idx.contiguous(), torch.int64, 'idx')
    n = x.shape[0]
    width = x.numel() // max(n, 1)
    out = torch.empty(tuple(idx.shape) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(lib().se3_gather_rows_padded(x.data_ptr(), idx.data_ptr(), n, idx.numel(), width, out.data_ptr(), _stream()),
          'se3_gather_rows_padded')
    return out


def anchor_max(x, dim):
    """HIP (csrc/rowops.hip): maximum over the anchor axis `dim` of a 3-D float32 tensor with 6 anchors, (A, R, C) -> (R, C) or
    (R, A, C) -> (R, C); other layouts go through torch."""
    if x.dim() != 3 or x.shape[dim] != 6 or dim not in (0, 1) or x.dtype != torch.float32 or not x.is_cuda or x.shape[2] % 4 or \
            x.stride(2) != 1 or x.stride(0) % 4 or x.stride(1) % 4 or x.data_ptr() % 16:
        return x.amax(dim)
    rows, C = x.shape[1 - dim], x.shape[2]
    out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    check(lib().se3_anchor_max(x.data_ptr(), 6, rows, C, x.stride(dim), x.stride(1 - dim), out.data_ptr(), _stream()), 'se3_anchor_max')
    return out


def neighbor_max_pool(x, idx):
    x = _req(x.contiguous(), torch.float32, 'x')
    idx = _req(idx.contiguous(), torch.int64, 'idx', 2)
    n = x.shape[0]
    width = x.numel() // max(n, 1)
    out = torch.empty((idx.shape[0],) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    check(lib().se3_neighbor_max_pool(x.data_ptr(), idx.data_ptr(), n, idx.shape[0], idx.shape[1], width, out.data_ptr(),
                                      _stream()), 'se3_neighbor_max_pool')
    return out


def neighbor_max_pool_bwd(x, idx, grad_out):
    """HIP (csrc/rowops.hip): gradient of neighbor_max_pool with respect to x (to the arg-max neighbour of every output element)."""
    x = _req(x.contiguous(), torch.float32, 'x')
    idx = _req(idx.contiguous(), torch.int64, 'idx', 2)
    g = _req(grad_out.contiguous(), torch.float32, 'grad_out')
    n = x.shape[0]
    width = x.numel() // max(n, 1)
    if idx.shape[0] == 0 or g.numel() == 0:                  # nothing pooled: no gradient (empty tensors have no address to hand to the C side)
        return torch.zeros_like(x)
    if TRAINING_DETERMINISTIC:
        bound = g.abs().max().reshape(1)
        fixed = torch.zeros(x.shape, dtype=torch.int64, device=x.device)
        dx = torch.empty_like(x)
        check(lib().se3_neighbor_max_pool_bwd_fixed(x.data_ptr(), idx.data_ptr(), g.data_ptr(), n, idx.shape[0], idx.shape[1], width,
                                                    bound.data_ptr(), fixed.data_ptr(), _stream()), 'se3_neighbor_max_pool_bwd_fixed')
        check(lib().se3_fixed_to_float(fixed.data_ptr(), fixed.numel(), bound.data_ptr(), idx.shape[0], 0, dx.data_ptr(), _stream()), 'se3_fixed_to_float')
        return dx
    dx = torch.zeros_like(x)
    check(lib().se3_neighbor_max_pool_bwd(x.data_ptr(), idx.data_ptr(), g.data_ptr(), n, idx.shape[0], idx.shape[1], width, dx.data_ptr(),
                                          _stream()), 'se3_neighbor_max_pool_bwd')
    return dx


def scatter_add_rows(g, idx, n):
    """Transpose of gather_rows_padded (its backward): out[idx[i]] += g[i] for idx[i] in [0, n), as order-independent 64-bit fixed-point sums
    (csrc/rowops.hip: scatter_add_rows_fixed_kernel; bit-identical runs).  g (m.., width..), idx (m..) -> (n, width..)."""
    g = _req(g.contiguous(), torch.float32, 'grad')
    idx = _req(idx.contiguous(), torch.int64, 'idx')
    m = idx.numel()
    width = g.numel() // max(m, 1)
    tail = tuple(g.shape[idx.dim():])
    if m == 0 or g.numel() == 0:
        return torch.zeros((n,) + tail, dtype=torch.float32, device=g.device)
    bound = g.abs().max().reshape(1)
    fixed = torch.zeros((n,) + tail, dtype=torch.int64, device=g.device)
    out = torch.empty((n,) + tail, dtype=torch.float32, device=g.device)
    check(lib().se3_scatter_add_rows_fixed(g.data_ptr(), idx.data_ptr(), n, m, width, bound.data_ptr(), fixed.data_ptr(), _stream()),
          'se3_scatter_add_rows_fixed')
    check(lib().se3_fixed_to_float(fixed.data_ptr(), fixed.numel(), bound.data_ptr(), m, 0, out.data_ptr(), _stream()), 'se3_fixed_to_float')
    return out


_gn_workspace = {}       # (device, stream) -> partial-statistics workspace


def group_norm_rows(x, weight, bias, groups, eps, leaky_slope, residual, x_bias=None, segments=None):
    """HIP (csrc/rowops.hip): GroupNorm of x [+ x_bias] with statistics over all leading dims, fused residual add + LeakyReLU;
    x_bias (C,) is the bias of the linear layer that produced x (its GEMM then runs bias-free).  segments: optional row
    offsets (S + 1 host ints, rows = all leading dims flattened) of independently normalised row ranges (one per pair)."""
    x = _req(x.contiguous(), torch.float32, 'x')
    C = x.shape[-1]
    rows = x.numel() // C
    if residual is not None:
        residual = _req(residual.contiguous(), torch.float32, 'residual')
        if residual.shape != x.shape:
            raise RuntimeError('group_norm_rows: residual shape mismatch')
    ws_bytes = lib().se3_group_norm_workspace_bytes(rows, C, groups)
    stream = _stream()
    key = (x.device, stream.value)                 # one workspace per launch stream (calls on a stream are ordered)
    ws = _gn_workspace.get(key)
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty((max(ws_bytes, 1 << 22),), dtype=torch.uint8, device=x.device)
        _gn_workspace[key] = ws
    out = torch.empty_like(x)
    nseg = 1 if segments is None else len(segments) - 1
    check(lib().se3_group_norm_segments_fwd(x.data_ptr(), x_bias.data_ptr() if x_bias is not None else None,
                                            residual.data_ptr() if residual is not None else None, weight.data_ptr(),
                                            bias.data_ptr(), rows, C, int(groups), _i64_array(segments) if nseg > 1 else None,
                                            nseg, float(eps), 1 if leaky_slope is not None else 0, float(leaky_slope or 0.0),
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'se3_group_norm_segments_fwd')
    return out


def group_norm_rows_bwd(grad_out, x, weight, bias, groups, eps, leaky_slope, residual, x_bias=None, segments=None):
    """HIP (csrc/rowops.hip): gradients of group_norm_rows -> (dx, dweight, dbias, dresidual or None, dx_bias or None)."""
    x = _req(x.contiguous(), torch.float32, 'x')
    g = _req(grad_out.contiguous(), torch.float32, 'grad_out')
    C = x.shape[-1]
    rows = x.numel() // C
    if residual is not None:
        residual = _req(residual.contiguous(), torch.float32, 'residual')
    ws_bytes = lib().se3_group_norm_bwd_workspace_bytes(C)
    stream = _stream()
    key = (x.device, stream.value, 'bwd')
    ws = _gn_workspace.get(key)
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty((max(ws_bytes, 1 << 22),), dtype=torch.uint8, device=x.device)
        _gn_workspace[key] = ws
    nseg = 1 if segments is None else len(segments) - 1
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if residual is not None else None
    params = torch.empty((nseg, 3, C), dtype=torch.float32, device=x.device)
    check(lib().se3_group_norm_segments_bwd(x.data_ptr(), x_bias.data_ptr() if x_bias is not None else None,
                                            residual.data_ptr() if residual is not None else None, weight.data_ptr(), bias.data_ptr(),
                                            g.data_ptr(), rows, C, int(groups), _i64_array(segments) if nseg > 1 else None, nseg, float(eps),
                                            1 if leaky_slope is not None else 0, float(leaky_slope or 0.0), dx.data_ptr(),
                                            dres.data_ptr() if dres is not None else None, params.data_ptr(), ws.data_ptr(), ws.numel(),
                                            stream), 'se3_group_norm_segments_bwd')
    params = params.sum(0) if nseg > 1 else params[0]
    return dx, params[0], params[1], dres, (params[2] if x_bias is not None else None)


_dense_ws = {}
_kpconv_split_ws = {}
KPCONV_SPLIT = True           # few-tile layers split their input channels over workgroups (False: A/B runs)


def _zeroed_workspace(cache, device, stream, nbytes):
    """A workspace that begins with arrival counters (in-kernel finalize / split reduction): one per stream, zero when first used; every
    call leaves the counters zero."""
    key = (device, stream.value)
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)
        cache[key] = ws
    return ws


def _dense_workspace(device, stream, nbytes):
    return _zeroed_workspace(_dense_ws, device, stream, nbytes)


def _check_counters(status, what, ws):
    """check() for the calls whose workspace begins with arrival counters: a failed or rejected call may leave them non-zero (every later call on
    the stream would then finalize at the wrong arrival), so the workspace is cleared before the error is raised."""
    if status != 0 and ws is not None:
        ws.zero_()
    check(status, what)


class Pending:
    """An activation in its pending form: `raw` (rows.., C) float32 plus up to two stages v -> lrelu(v * scale + shift, slope) that its
    consumer applies while loading it (the GroupNorm [+ LeakyReLU] of the producer, reduced to per-(segment, channel) affine tables by
    group_norm_stats / dense_norm).  `affines`: list of (nseg, 2, C) tensors; `slopes`: list of floats (1.0 = no LeakyReLU)."""
    __slots__ = ('raw', 'affines', 'slopes', 'segments')

    def __init__(self, raw, affines, slopes, segments):
        self.raw, self.affines, self.slopes, self.segments = raw, list(affines), list(slopes), segments

    def then(self, affine, slope):
        return Pending(self.raw, self.affines + [affine], self.slopes + [slope], self.segments)


class BlockedFeatures:
    """(points, 6, C) features in a layout a fused KPConv kernel reads whole cache lines / rows from (group_norm_apply(blocked=True) writes
    it, kpconv_inter_so3 reads it); `plain()` is the ordinary tensor.
    kind 1: [point][C / 16][anchor pair][16 channels][2 anchors] (csrc/kpconv_mfma.hip: per-lane gathers of 8 bytes)
    kind 2: [point][C / 8][6 anchors][8 channels]                (csrc/kpconv_union.hip: a point's 8-channel chunk is 192 contiguous bytes)"""
    __slots__ = ('data', 'shape', 'kind', 'amax', 'amax_tag')

    def __init__(self, data, shape, kind=1, amax=None, amax_tag=None):
        # amax: one-element device tensor holding the largest |value| (written by the apply pass that produced `data`), or None: the fused
        # kernels scale the features by a power of two from it before their f16 split (csrc/kpconv_sums.h: x_split_scale)
        self.data, self.shape, self.kind, self.amax, self.amax_tag = data, tuple(shape), kind, amax, amax_tag

    def plain(self):
        n, a, c = self.shape
        if self.kind == 2:
            return self.data.view(n, c // 8, a, 8).permute(0, 2, 1, 3).reshape(n, a, c)
        return self.data.view(n, c // 16, a // 2, 16, 2).permute(0, 2, 4, 1, 3).reshape(n, a, c)


KPCONV_BLOCKED = True          # False: the KPConv input in the plain layout (A/B runs)
# The union-staged fused KPConv (csrc/kpconv_union.hip), wherever a spatial order of the query points is registered (se3et_amd/data.py):
# SE3_KPCONV_UNION = 0 never; 1 (default) where it measured faster than the per-lane-gather kernel alone on the GPU (output widths <= 64 of the
# stride-1 layers and the first strided one: profiles/r05_kpconv_union_check.txt: 18-26 % on those layers with the 128-row cap); all: every
# layer.  With the order and plan launches the forward as a whole reads the same or slightly higher in alternating bench runs (488.1 / 488.6
# against 487.7 / 483.0 pairs/s; with the first form's 160-row cap it read 1-1.5 % lower and was off).
_KPCONV_UNION_ENV = os.environ.get('SE3_KPCONV_UNION', '1')
KPCONV_UNION = _KPCONV_UNION_ENV != '0'
KPCONV_UNION_ALL = _KPCONV_UNION_ENV == 'all'
KPCONV_UNION_BIG = os.environ.get('SE3_KPCONV_UNION_BIG', '0') == '1'      # A/B: the policy's layers also for clouds beyond 8192 points (order through torch.sort)
KPCONV_UNION_MIN_POINTS = 24000           # stage-0 points of a pyramid from which the pyramid builder registers orders (single pairs stay on the gather kernel)


if os.environ.get('SE3_KPCONV_UNION_WGS'):        # A/B runs: workgroups per launch of the union-staged kernel (default: 256 up to 64 output channels, 1024 beyond)
    lib().se3_debug_set_kpconv_union_variant(int(os.environ['SE3_KPCONV_UNION_WGS']) << 8)


_amax_rings = {}


def _amax_slot(device, stream):
    """A zeroed device word for one apply pass to atomicMax the largest magnitude of its output into: taken from a per-stream ring of 4096
    words that is cleared (one fill, in stream order) every time it wraps."""
    key = (device, stream.value)
    ring = _amax_rings.get(key)
    if ring is None:
        ring = _amax_rings[key] = [torch.zeros((4096,), dtype=torch.float32, device=device), 0, 0]
    if ring[1] == 4096:
        ring[0].zero_()
        ring[1] = 0
        ring[2] += 1                     # generation: words handed out before the wrap are no longer their owners'
    slot = ring[0][ring[1]:ring[1] + 1]
    ring[1] += 1
    return slot, (ring, ring[2], ring[1] - 1)


def _amax_live(blocked):
    """The magnitude word of a BlockedFeatures, or None when the ring it came from has wrapped since (an object kept across a wrap: its word
    was zeroed and may describe another tensor by now -- the kernels then split the features as they are)."""
    if blocked.amax is None or blocked.amax_tag is None:
        return blocked.amax
    ring, gen, pos = blocked.amax_tag
    return blocked.amax if ring[2] == gen else None        # (the wrap zeroes the WHOLE ring: every word of an earlier generation is dead)


def _kpconv_union_pays(Cin, Cout, same_cloud):
    return KPCONV_UNION_ALL or (Cout <= 64 and (same_cloud or Cout <= 32))


def dense_norm_ok(x, weight, groups):
    """True when the fused dense layer + GroupNorm statistics kernel (csrc/dense_norm.hip) takes this layer."""
    raw = x.raw if isinstance(x, Pending) else x
    if torch.is_grad_enabled() and (raw.requires_grad or weight.requires_grad):
        return False
    N, K = weight.shape
    return (raw.is_cuda and raw.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_contiguous() and raw.is_contiguous()
            and raw.shape[-1] == K and 32 <= K <= 1024 and K & (K - 1) == 0 and (N in (32, 64, 128) or N % 256 == 0) and N % groups == 0
            and N // groups in (1, 2, 4, 8, 16, 32) and N <= 4096
            and raw.data_ptr() % 16 == 0)


def dense_norm(x, weight, linear_bias, norm_weight, norm_bias, groups, eps, segments=None):
    """HIP (csrc/dense_norm.hip): the dense layer of a unary block with the GroupNorm statistics of its output taken from the accumulators.
    x: a tensor or a Pending (its stages are applied on load).  -> Pending(raw = T(x) W^T without the bias, [affine of GroupNorm(raw + bias)],
    slope left to the caller via .slopes[-1])."""
    pend = x if isinstance(x, Pending) else Pending(x, [], [], segments)
    raw = _req(pend.raw, torch.float32, 'x')
    if len(pend.affines) > 2:
        raise RuntimeError('dense_norm: at most two pending stages')
    segments = pend.segments if segments is None else segments
    N, K = weight.shape
    rows = raw.numel() // K
    stream = _stream()
    Wp = _linear_weight_pieces(weight, stream)
    nseg = 1 if segments is None else len(segments) - 1
    out = torch.empty(raw.shape[:-1] + (N,), dtype=torch.float32, device=raw.device)
    affine = torch.empty((nseg, 2, N), dtype=torch.float32, device=raw.device)
    ws = _dense_workspace(raw.device, stream, lib().se3_dense_norm_workspace_bytes(int(groups)))
    aa = pend.affines + [None, None]
    sl = pend.slopes + [1.0, 1.0]
    with _timed('dense', 4.0 * rows * (K + N)):           # (bench.py roofline_dense: activations read + product written)
        _check_counters(lib().se3_dense_norm_fwd(raw.data_ptr(), rows, K, aa[0].data_ptr() if aa[0] is not None else None, float(sl[0]),
                                       aa[1].data_ptr() if aa[1] is not None else None, float(sl[1]), Wp.data_ptr(), N,
                                       linear_bias.data_ptr() if linear_bias is not None else None, norm_weight.data_ptr(), norm_bias.data_ptr(),
                                       int(groups), float(eps), _i64_array(segments) if nseg > 1 else None, nseg, out.data_ptr(),
                                       affine.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'se3_dense_norm_fwd', ws)
    return Pending(out, [affine], [1.0], segments)


def dense_stats(x, weight, linear_bias, norm_weight, norm_bias, groups, eps, segments=None):
    """HIP (csrc/dense_norm.hip, statistics-only mode): the affine table (nseg, 2, N) of GroupNorm(T(x) W^T + linear_bias) -- the GEMM runs and
    is reduced to its statistics, the product is never written.  x: a tensor or a Pending (at most two stages)."""
    pend = x if isinstance(x, Pending) else Pending(x, [], [], segments)
    raw = _req(pend.raw, torch.float32, 'x')
    if len(pend.affines) > 2:
        raise RuntimeError('dense_stats: at most two pending stages')
    segments = pend.segments if segments is None else segments
    N, K = weight.shape
    rows = raw.numel() // K
    stream = _stream()
    Wp = _linear_weight_pieces(weight, stream)
    nseg = 1 if segments is None else len(segments) - 1
    affine = torch.empty((nseg, 2, N), dtype=torch.float32, device=raw.device)
    ws = _dense_workspace(raw.device, stream, lib().se3_dense_norm_workspace_bytes(int(groups)))
    aa = pend.affines + [None, None]
    sl = pend.slopes + [1.0, 1.0]
    with _timed('dense', 4.0 * rows * K):                 # (statistics only: the activations read, nothing written)
        _check_counters(lib().se3_dense_norm_fwd(raw.data_ptr(), rows, K, aa[0].data_ptr() if aa[0] is not None else None, float(sl[0]),
                                       aa[1].data_ptr() if aa[1] is not None else None, float(sl[1]), Wp.data_ptr(), N,
                                       linear_bias.data_ptr() if linear_bias is not None else None, norm_weight.data_ptr(), norm_bias.data_ptr(),
                                       int(groups), float(eps), _i64_array(segments) if nseg > 1 else None, nseg, None,
                                       affine.data_ptr(), ws.data_ptr(), ws.numel(), stream), 'se3_dense_norm_fwd (statistics only)', ws)
    return affine


_nonzero_norm_cache = {}


def norm_weight_nonzero(weight):
    """True when no entry of a GroupNorm weight is zero (one host synchronisation per weight version): dense_residual's shortcut form divides
    by the shortcut norm's scale."""
    key = (weight.data_ptr(), weight.device.index, tuple(weight.shape))
    hit = _nonzero_norm_cache.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        hit = (weakref.ref(weight), weight._version, bool((weight.detach().abs() > 1e-30).all()))
        if len(_nonzero_norm_cache) > 256:
            _nonzero_norm_cache.clear()
        _nonzero_norm_cache[key] = hit
    return hit[2]


def dense_residual(x, weight, affine, residual=None, shortcut=None, final_slope=1.0, segments=None):
    """HIP (csrc/dense_norm.hip, final modes): out = lrelu( (T(x) W^T) * scale + shift + R ) with (scale, shift) = `affine` from dense_stats on
    the same operands; R = `residual` (rows.., N) tensor, or the shortcut layer `shortcut` = (x2 tensor, weight2 (N, K2), affine2) evaluated in
    the same kernel, or nothing.  The tail of a bottleneck block with only its output written."""
    pend = x if isinstance(x, Pending) else Pending(x, [], [], segments)
    raw = _req(pend.raw, torch.float32, 'x')
    if len(pend.affines) > 2:
        raise RuntimeError('dense_residual: at most two pending stages')
    segments = pend.segments if segments is None else segments
    N, K = weight.shape
    rows = raw.numel() // K
    stream = _stream()
    Wp = _linear_weight_pieces(weight, stream)
    nseg = 1 if segments is None else len(segments) - 1
    if tuple(affine.shape) != (nseg, 2, N):
        raise RuntimeError('dense_residual: affine table %s for %d segment(s) x %d features' % (tuple(affine.shape), nseg, N))
    out = torch.empty(raw.shape[:-1] + (N,), dtype=torch.float32, device=raw.device)
    x2p = wp2 = aff2 = resp = None
    K2 = 0
    if shortcut is not None:
        x2, w2, aff2t = shortcut
        x2 = _req(x2, torch.float32, 'shortcut input')
        K2 = w2.shape[1]
        if w2.shape[0] != N or x2.numel() != rows * K2 or tuple(aff2t.shape) != (nseg, 2, N) or residual is not None:
            raise RuntimeError('dense_residual: shortcut layer shapes')
        x2p, wp2, aff2 = x2.data_ptr(), _linear_weight_pieces(w2, stream).data_ptr(), aff2t.data_ptr()
    elif residual is not None:
        residual = _req(residual, torch.float32, 'residual')
        if residual.numel() != rows * N:
            raise RuntimeError('dense_residual: residual shape %s' % (tuple(residual.shape),))
        resp = residual.data_ptr()
    aa = pend.affines + [None, None]
    sl = pend.slopes + [1.0, 1.0]
    with _timed('dense', 4.0 * rows * (K + K2 + N + (N if resp is not None else 0))):
        check(lib().se3_dense_residual_fwd(raw.data_ptr(), rows, K, aa[0].data_ptr() if aa[0] is not None else None, float(sl[0]),
                                           aa[1].data_ptr() if aa[1] is not None else None, float(sl[1]), Wp.data_ptr(), affine.data_ptr(),
                                           x2p, K2, wp2, aff2, resp, N, float(final_slope), _i64_array(segments) if nseg > 1 else None, nseg,
                                           out.data_ptr(), stream), 'se3_dense_residual_fwd')
    return out


def group_norm_stats(x, weight, bias, groups, eps, x_bias=None, segments=None):
    """HIP (csrc/rowops.hip): the affine table (nseg, 2, C) of GroupNorm over T(x) (x: tensor, or Pending with ONE stage)."""
    pend = x if isinstance(x, Pending) else Pending(x, [], [], segments)
    raw = _req(pend.raw.contiguous(), torch.float32, 'x')
    if len(pend.affines) > 1:
        raise RuntimeError('group_norm_stats: at most one pending stage')
    segments = pend.segments if segments is None else segments
    C = raw.shape[-1]
    rows = raw.numel() // C
    stream = _stream()
    nseg = 1 if segments is None else len(segments) - 1
    affine = torch.empty((nseg, 2, C), dtype=torch.float32, device=raw.device)
    ws = _dense_workspace(raw.device, stream, lib().se3_group_norm_stats_workspace_bytes(C))
    pa = pend.affines[0] if pend.affines else None
    _check_counters(lib().se3_group_norm_stats(raw.data_ptr(), pa.data_ptr() if pa is not None else None, float(pend.slopes[0]) if pa is not None else 1.0,
                                     x_bias.data_ptr() if x_bias is not None else None, weight.data_ptr(), bias.data_ptr(), rows, C,
                                     int(groups), _i64_array(segments) if nseg > 1 else None, nseg, float(eps), affine.data_ptr(),
                                     ws.data_ptr(), ws.numel(), stream), 'se3_group_norm_stats', ws)
    return affine


def group_norm_apply(x, residual=None, final_slope=1.0, blocked=False):
    """HIP (csrc/rowops.hip): a Pending made concrete: lrelu_f(T_b(T_a(raw)) + R), R = a tensor, a Pending with one stage (slope 1), or None.
    blocked=True (raw (points, 6, C), C % 16 == 0): the result in the fused KPConv's gather layout, as BlockedFeatures."""
    if not x.affines or len(x.affines) > 2:
        raise RuntimeError('group_norm_apply: one or two pending stages')
    raw = _req(x.raw, torch.float32, 'x')
    C = raw.shape[-1]
    rows = raw.numel() // C
    res, res_aff = None, None
    if isinstance(residual, Pending):
        if len(residual.affines) != 1 or residual.slopes[0] != 1.0:
            raise RuntimeError('group_norm_apply: the residual may carry one stage without LeakyReLU')
        res, res_aff = _req(residual.raw, torch.float32, 'residual'), residual.affines[0]
    elif residual is not None:
        res = _req(residual.contiguous(), torch.float32, 'residual')
    if res is not None and res.shape != raw.shape:
        raise RuntimeError('group_norm_apply: residual shape mismatch')
    segments = x.segments
    nseg = 1 if segments is None else len(segments) - 1
    out = torch.empty_like(raw)
    aa = x.affines + [None]
    sl = x.slopes + [1.0]
    stream = _stream()
    amax, amax_tag = _amax_slot(raw.device, stream) if blocked else (None, None)
    check(lib().se3_group_norm_apply_amax(raw.data_ptr(), aa[0].data_ptr(), float(sl[0]), aa[1].data_ptr() if aa[1] is not None else None, float(sl[1]),
                                          res.data_ptr() if res is not None else None, res_aff.data_ptr() if res_aff is not None else None,
                                          float(final_slope), rows, C, _i64_array(segments) if nseg > 1 else None, nseg, int(blocked),
                                          out.data_ptr(), amax.data_ptr() if amax is not None else None, stream),
          'se3_group_norm_apply')
    return BlockedFeatures(out, raw.shape, int(blocked), amax, amax_tag) if blocked else out


_host_table_cache = {}


def _host_table(t, dtype):
    """Constant module tables (kernel points, permutation indices) as host arrays, cached per owning tensor object, view and
    version.  The entry holds a weak reference to the owner (the module's parameter / buffer; for a view, its base): a freed
    tensor's address and id can be handed to another tensor with other values, so neither alone identifies the table."""
    base = t._base if t._base is not None else t
    key = (id(base), base._version, t.data_ptr(), tuple(t.shape), tuple(t.stride()))
    hit = _host_table_cache.get(key)
    if hit is None or hit[1]() is not base:
        hit = (t.detach().to('cpu', dtype).contiguous(), weakref.ref(base))
        if len(_host_table_cache) > 256:
            _host_table_cache.clear()
        _host_table_cache[key] = hit
    return hit[0]


# KPConv path: True / 'auto' = the fused matrix-core kernel (csrc/kpconv_mfma.hip) where the channel counts allow (Cin % 8, Cout % 32),
# 'sums' = the same contraction with the orbit sums written to HBM by a separate gather kernel (A/B runs), False = the round-1 path (slot
# sums G in HBM + library f32 GEMM; also the fallback for other channel counts / slot tables and the operand of the backward pass).
# All paths agree to f32 round-off (tests/test_gpu_ops.py::test_kpconv_matrix_core_path_has_f32_accuracy).
KPCONV_MATRIX_CORE = {'mfma': True, 'fused': True, 'sums': 'sums', 'gemm': False}.get(os.environ.get('SE3_KPCONV_PATH', 'auto'), 'auto')


def _kpconv_use_matrix_core(Cin, Cout, P):
    return True if KPCONV_MATRIX_CORE == 'auto' else KPCONV_MATRIX_CORE


_BUILTIN_KIDX = [[0, 1, 1, 1, 1, 2], [1, 0, 1, 2, 1, 1], [1, 1, 0, 1, 2, 1], [1, 2, 1, 0, 1, 1], [1, 1, 2, 1, 0, 1], [2, 1, 1, 1, 1, 0],
                 [3, 3, 3, 4, 4, 4], [3, 4, 3, 3, 4, 4], [3, 4, 4, 3, 3, 4], [3, 3, 4, 4, 3, 4], [4, 3, 3, 4, 4, 3], [4, 4, 3, 3, 4, 3],
                 [4, 4, 4, 3, 3, 3], [4, 3, 4, 4, 3, 3], [5, 5, 5, 5, 5, 5]]
_BUILTIN_RIDX = [[0, 3, 3, 3, 3, 5], [1, 0, 4, 5, 2, 1], [2, 2, 0, 4, 5, 4], [3, 5, 2, 0, 4, 3], [4, 4, 5, 2, 0, 2], [5, 1, 1, 1, 1, 0]]
def _builtin_slot_tables(kt, rt):
    """True when the module's (k, r) -> s and (a, r) -> t tables are the SE3ET configuration compiled into the matrix-core kernel.  The
    verdict is an attribute of the host copies `_host_table` returns, so it lives and dies with their cache entries (a verdict keyed by the
    host tensors' addresses could outlive them and be inherited by other tables placed at the same addresses)."""
    hit = getattr(kt, '_se3_builtin', None)
    if hit is None or hit[0] is not rt:
        hit = (rt, kt.tolist() == _BUILTIN_KIDX and rt.tolist() == _BUILTIN_RIDX)
        kt._se3_builtin = hit
    return hit[1]


_weight_piece_cache = {}          # (data_ptr, Cin, Cout, device) -> (weakref to the weight tensor, its version counter, pieces)


def _kpconv_weight_pieces(weights, Cin, Cout, stream):
    """f16 hi / lo MFMA fragments of KPConvInterSO3.weights (se3_kpconv_split_weights_f16).  Without autograd (inference) they are kept per
    weight VERSION: torch bumps `_version` on every in-place update (optimizer steps, load_state_dict, copy_), so a stale entry is never
    used as long as the weights are not rewritten behind torch's back (`.data` arithmetic, raw pointers) -- call
    clear_weight_caches() after such an edit.  Under autograd the fragments are rebuilt per call (3 small launches)."""
    w = _req(weights.detach().contiguous(), torch.float32, 'weights', 4)
    cacheable = not torch.is_grad_enabled() and w.data_ptr() == weights.data_ptr()
    key = (w.data_ptr(), Cin, Cout, w.device.index)
    if cacheable:
        hit = _weight_piece_cache.get(key)
        if hit is not None and hit[0]() is weights and hit[1] == weights._version:
            return hit[2].get()[0]
    Wp = torch.empty((lib().se3_kpconv_weight_pieces_bytes(Cin, Cout),), dtype=torch.uint8, device=w.device)
    check(lib().se3_kpconv_split_weights_f16(w.data_ptr(), Cin, Cout, Wp.data_ptr(), stream), 'se3_kpconv_split_weights_f16')
    if cacheable:
        with _TIMING_LOCK:
            if len(_weight_piece_cache) > 256:
                _weight_piece_cache.clear()
            _weight_piece_cache[key] = (weakref.ref(weights), weights._version, _Shared(Wp), _fingerprint(weights))
    return Wp


CACHE_EPOCH = [0]          # bumped whenever cached weight pieces are dropped: part of the key of plans that hold raw pointers into them (cdriver)


def clear_weight_caches():
    CACHE_EPOCH[0] += 1
    _padded_weight_cache.clear()
    _weight_piece_cache.clear()
    _linear_piece_cache.clear()
    _stacked_weight_cache.clear()


def validate_weight_caches():
    """Compares the content fingerprint of every cached weight with the weight's CURRENT values (all sums on the device, one host
    synchronisation in all) and drops the entries that no longer match or whose Parameter is gone.  For code that writes weights behind
    torch's version counter (`p.data.copy_()`, EMA swaps through `.data`, hand-written checkpoint loaders).  Returns the number of entries
    dropped."""
    entries = []
    for cache in (_weight_piece_cache, _linear_piece_cache, _stacked_weight_cache, _padded_weight_cache):
        for key, hit in list(cache.items()):
            w = hit[0]()
            if w is not None and cache is _linear_piece_cache and w.data_ptr() != key[0]:
                # a block of its owner (key: address, N, K, row stride): rebuild the view
                off = (key[0] - w.data_ptr()) // w.element_size()
                w = torch.as_strided(w, (key[1], key[2]), (key[3], 1), w.storage_offset() + off) if 0 <= off < w.numel() else None
            if w is None or w.data_ptr() != key[0]:
                cache.pop(key, None)
                entries.append(None)
            else:
                entries.append((cache, key, hit[3], _fingerprint(w)))
    live = [e for e in entries if e is not None]
    dropped = len(entries) - len(live)
    if live:
        same = (torch.stack([e[2].to(live[0][2].device) for e in live]) == torch.stack([e[3].to(live[0][2].device) for e in live])).tolist()
        for ok, (cache, key, _, _) in zip(same, live):
            if not ok:
                cache.pop(key, None)
                dropped += 1
    if dropped:
        CACHE_EPOCH[0] += 1          # plans holding pointers to the dropped pieces (cdriver._static_plan) are rebuilt
    return dropped


def kpconv_slot_sums(x, q_pts, s_pts, idx, kernel_points, kidx, ridx, sigma):
    """HIP (csrc/kpconv_so3.hip): G (P * 6, 36 Cin), the slot-summed neighbourhood features with out = G @ weights.reshape(36 Cin, Cout)."""
    x = _req(x.contiguous(), torch.float32, 'x', 3)
    q_pts, s_pts = _req(q_pts.contiguous(), torch.float32, 'q_pts', 2), _req(s_pts.contiguous(), torch.float32, 's_pts', 2)
    idx = _req(idx.contiguous(), torch.int64, 'neighb_inds', 2)
    P, NN = idx.shape
    Ns, A, Cin = x.shape
    kp, kt, rt = _host_table(kernel_points, torch.float32), _host_table(kidx, torch.int64), _host_table(ridx, torch.int64)
    G = torch.empty((P * 6, 36 * Cin), dtype=torch.float32, device=x.device)
    check(lib().se3_kpconv_so3_gather(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), x.data_ptr(), kp.data_ptr(),
                                      kt.data_ptr(), rt.data_ptr(), float(sigma), P, Ns, NN, Cin, G.data_ptr(), _stream()),
          'se3_kpconv_so3_gather')
    return G


def kpconv_inter_so3_bwd(grad_out, x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma, need_x=True, need_w=True):
    """Gradients of kpconv_inter_so3 with respect to (x, weights): two library GEMMs on the slot-sum matrix (dW = G^T dout, dG = dout W^T)
    and the HIP transpose of the gather (csrc/kpconv_so3.hip: kpconv_scatter_kernel)."""
    x = _req(x.contiguous(), torch.float32, 'x', 3)
    q_pts, s_pts = _req(q_pts.contiguous(), torch.float32, 'q_pts', 2), _req(s_pts.contiguous(), torch.float32, 's_pts', 2)
    idx = _req(idx.contiguous(), torch.int64, 'neighb_inds', 2)
    P, NN = idx.shape
    Ns, A, Cin = x.shape
    Cout = weights.shape[-1]
    if P == 0:                                                   # no query points: zero gradients (empty tensors have no address for the C side)
        return (torch.zeros_like(x) if need_x else None), (torch.zeros_like(weights) if need_w else None)
    d2 = _req(grad_out.contiguous(), torch.float32, 'grad_out').reshape(P * 6, Cout)
    dx = dw = None
    if need_w:
        G = kpconv_slot_sums(x, q_pts, s_pts, idx, kernel_points, kidx, ridx, sigma)
        dw = mm(G.t(), d2).view(6, 6, Cin, Cout)
        del G
    if need_x:
        kp, kt, rt = _host_table(kernel_points, torch.float32), _host_table(kidx, torch.int64), _host_table(ridx, torch.int64)
        W2 = weights.detach().reshape(36 * Cin, Cout)              # as a dense layer's (N = 36 Cin, K = Cout) weight: dG = dout W2^T
        if Cout % 32 == 0 and d2.data_ptr() % 16 == 0 and P * 6 * max(Cout, 36 * Cin) < 2 ** 31:
            # round 5: on the f16-split streaming kernel (csrc/dense_norm.hip; f32 accurate, ~2x the library's f32-MFMA rate on these shapes);
            # its weight pieces are cached per weight version like every dense layer's (keyed on the view's address / shape, owned by `weights`)
            with torch.no_grad():           # (the view's _base must be `weights` itself -- the cache entry's owner; a view of weights.detach() is owned by a temporary)
                Wv = weights.view(36 * Cin, Cout) if weights.is_contiguous() else W2.contiguous()
            dG = linear_stream(d2, Wv)
        else:
            dG = mm(d2, W2.t())
        if KPCONV_BACKWARD_DETERMINISTIC:
            # order-independent sums (64-bit fixed point at a scale from max |dG|): bit-identical runs (csrc/kpconv_so3.hip)
            # (an upper bound of |dG| = |dout W^T| from the small operands -- max |dout| times the largest absolute row sum of W -- instead of a pass
            # over the (6 P, 36 Cin) product: the scale only has to prevent overflow)
            # (two fused max-magnitude reductions and Cout as the row length -- |dG| <= max |dout| max |W| Cout -- instead of abs / max / abs /
            #  row sums / max: 3 launches for 6 per layer; the looser bound costs the 64-bit sums a few of their ~39 spare bits)
            bound = ((torch.linalg.vector_norm(d2, float('inf')) * torch.linalg.vector_norm(W2, float('inf'))).mul_(float(Cout)).reshape(1)
                     if d2.numel() and W2.numel() else d2.new_zeros(1))
            fixed = torch.zeros(x.shape, dtype=torch.int64, device=x.device)
            dx = torch.empty_like(x)
            check(lib().se3_kpconv_so3_gather_bwd_fixed(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), dG.data_ptr(), kp.data_ptr(),
                                                        kt.data_ptr(), rt.data_ptr(), float(sigma), P, Ns, NN, Cin, bound.data_ptr(),
                                                        fixed.data_ptr(), _stream()), 'se3_kpconv_so3_gather_bwd_fixed')
            check(lib().se3_kpconv_fixed_to_float(fixed.data_ptr(), fixed.numel(), bound.data_ptr(), P, dx.data_ptr(), _stream()),
                  'se3_kpconv_fixed_to_float')
        else:
            dx = torch.zeros_like(x)
            check(lib().se3_kpconv_so3_gather_bwd(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), dG.data_ptr(), kp.data_ptr(),
                                                  kt.data_ptr(), rt.data_ptr(), float(sigma), P, Ns, NN, Cin, dx.data_ptr(), _stream()),
                  'se3_kpconv_so3_gather_bwd')
    return dx, dw


KPCONV_BACKWARD_DETERMINISTIC = os.environ.get('SE3_KPCONV_BWD', 'fixed') != 'float'      # 64-bit fixed-point scatter (bit-identical runs); 'float': hardware float atomics
TRAINING_DETERMINISTIC = KPCONV_BACKWARD_DETERMINISTIC      # the other scatter-adds of the training step (max-pool, row gather, LayerNorm parameter sums) follow the same switch
_neighbor_table_cache = {}


def _kpconv_neighbor_table(q_pts, s_pts, idx, kernel_points, sigma, P, Ns, NN, stream):
    """The neighbour table of the fused KPConv (valid neighbours compacted + 16 orbit weights each): a function of the geometry only, so the
    layers of a pyramid stage (same query / support points, neighbour indices, kernel points and extent) share it.  Kept per stream for the
    LAST geometry seen -- identified by the tensor objects themselves (weak references: a freed tensor's address can be reused) and their
    version counters (in-place changes)."""
    kpd = _req(kernel_points.detach().contiguous(), torch.float32, 'kernel_points', 2)
    # (the kernel points enter by VALUE -- the bytes of their cached host copy: the layers of a stage own different Parameter objects holding
    # the same 15 points, and shared one table only by accident of the cache's depth before round 5: 10 -> 7 table launches per forward)
    kph = _host_table(kernel_points, torch.float32)
    kp_key = getattr(kph, '_se3_bytes', None)
    if kp_key is None:
        kp_key = kph.numpy().tobytes()
        kph._se3_bytes = kp_key
    key = (q_pts.data_ptr(), q_pts._version, s_pts.data_ptr(), s_pts._version, idx.data_ptr(), idx._version, kp_key, float(sigma), P, Ns, NN)
    hit = _neighbor_table_cache.get(stream.value)
    if hit is not None and hit[0] == key and all(r() is t for r, t in zip(hit[1], (q_pts, s_pts, idx))):
        return hit[2]
    nbytes = lib().se3_kpconv_neighbor_table_bytes(P, NN)
    tab = torch.empty((nbytes,), dtype=torch.uint8, device=q_pts.device)
    check(lib().se3_kpconv_neighbor_table(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), kpd.data_ptr(), float(sigma), P, Ns, NN,
                                          tab.data_ptr(), nbytes, stream), 'se3_kpconv_neighbor_table')
    _neighbor_table_cache[stream.value] = (key, tuple(weakref.ref(t) for t in (q_pts, s_pts, idx)), tab)
    return tab


_point_orders = {}               # points.data_ptr() -> (weakref to the points tensor, its version, order (G * 16 int32), G)
_union_plan_cache = {}           # stream -> (key, weakrefs, plan)
_kpconv_union_split_ws = {}


def register_point_order(points, lengths, cell):
    """A spatial order of one stage's stacked points (per cloud: Morton order of floor(p / cell)), kept while the `points` tensor lives: the
    union-staged fused KPConv (csrc/kpconv_union.hip) takes its 16-point tiles along it.  Tile membership only -- no tensor is reordered.
    Called by the pyramid builder (se3et_amd/data.py); lengths: the stage's per-cloud counts (host)."""
    lens = [int(v) for v in (lengths.tolist() if hasattr(lengths, 'tolist') else lengths)]
    n = points.shape[0]
    if not KPCONV_UNION or not points.is_cuda or n == 0 or len(lens) > 32 or sum(lens) != n:
        return None
    points = _req(points, torch.float32, 'points', 2)
    stream = _stream()
    la = _i64_array(lens)
    G = int(lib().se3_point_order_groups(la, len(lens)))
    order = torch.empty((G * 16,), dtype=torch.int32, device=points.device)
    if max(lens) <= 8192:
        # one launch: a workgroup per cloud sorts (Morton code, index) in LDS
        check(lib().se3_point_order(points.data_ptr(), n, la, len(lens), float(cell), order.data_ptr(), stream), 'se3_point_order')
    else:
        keys = torch.empty((n,), dtype=torch.int64, device=points.device)
        check(lib().se3_point_order_keys(points.data_ptr(), n, la, len(lens), float(cell), keys.data_ptr(), stream), 'se3_point_order_keys')
        sk, si = torch.sort(keys, stable=True)
        check(lib().se3_point_order_place(sk.data_ptr(), si.data_ptr(), n, la, len(lens), order.data_ptr(), stream), 'se3_point_order_place')
    with _TIMING_LOCK:
        if len(_point_orders) > 256:
            for k in [k for k, v in _point_orders.items() if v[0]() is None]:
                del _point_orders[k]
        _point_orders[points.data_ptr()] = (weakref.ref(points), points._version, order, G)
    return order


def register_point_orders(points_list, lengths_list, cells):
    """register_point_order for several stages of one pyramid, in ONE launch when every cloud has at most 8192 points."""
    lens = [[int(v) for v in (l.tolist() if hasattr(l, 'tolist') else l)] for l in lengths_list]
    ok = KPCONV_UNION and 1 <= len(points_list) <= 4 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] > 0
                                                             and len(l) <= 32 and sum(l) == p.shape[0] and max(l) <= 8192
                                                             for p, l in zip(points_list, lens))
    if not ok:
        # (clouds beyond 8192 points -- the KITTI configuration -- would take the keys + torch.sort + placement form, whose launches cost more
        # than the four narrow layers gain there: 142 against 141 pairs/s at C3 with the 160-row cap, 138.6 against 140.4 with the 128-row cap
        # and NN = 38 -- SE3_KPCONV_UNION_BIG=1; only on request)
        return [register_point_order(p, l, c) for p, l, c in zip(points_list, lengths_list, cells)] if (KPCONV_UNION_ALL or KPCONV_UNION_BIG) else [None] * len(points_list)
    S = len(points_list)
    las = [_i64_array(l) for l in lens]
    Gs = [int(lib().se3_point_order_groups(la, len(l))) for la, l in zip(las, lens)]
    orders = [torch.empty((G * 16,), dtype=torch.int32, device=p.device) for G, p in zip(Gs, points_list)]
    vp = ctypes.c_void_p
    check(lib().se3_point_order_stages((vp * S)(*[p.data_ptr() for p in points_list]), _i64_array([p.shape[0] for p in points_list]),
                                       (vp * S)(*[ctypes.cast(la, vp).value for la in las]), (ctypes.c_int * S)(*[len(l) for l in lens]),
                                       (ctypes.c_float * S)(*[float(c) for c in cells]), (vp * S)(*[o.data_ptr() for o in orders]), S, _stream()),
          'se3_point_order_stages')
    with _TIMING_LOCK:
        if len(_point_orders) > 256:
            for k in [k for k, v in _point_orders.items() if v[0]() is None]:
                del _point_orders[k]
        for p, o, G in zip(points_list, orders, Gs):
            _point_orders[p.data_ptr()] = (weakref.ref(p), p._version, o, G)
    return orders


def point_order(points):
    hit = _point_orders.get(points.data_ptr())
    if hit is None or hit[0]() is not points or hit[1] != points._version:
        return None
    return hit[2], hit[3]


def _kpconv_union_plan(tab, q_pts, s_pts, idx, order, G, P, NN, stream):
    """Per group of 16 order positions the distinct support rows of its neighbour lists and every list slot's index into them
    (se3_kpconv_union_plan): a function of (order, neighbour table); kept per stream for the last table seen."""
    # (the plan reads the table's compacted lists and counts only -- the geometry, not the layer's kernel points: layers with tables of
    # their own over the same (queries, supports, neighbour indices) share it)
    hit = _union_plan_cache.get(stream.value)
    if hit is not None and hit[1] is order and all(r() is t for r, t in zip(hit[0], (q_pts, s_pts, idx))) and hit[3] == (q_pts._version, s_pts._version, idx._version):
        return hit[2]
    nbytes = lib().se3_kpconv_union_plan_bytes(G, NN)
    plan = torch.empty((nbytes,), dtype=torch.uint8, device=q_pts.device)
    order.record_stream(torch.cuda.current_stream())
    check(lib().se3_kpconv_union_plan(tab.data_ptr(), P, NN, order.data_ptr(), G, plan.data_ptr(), nbytes, stream), 'se3_kpconv_union_plan')
    _union_plan_cache[stream.value] = (tuple(weakref.ref(t) for t in (q_pts, s_pts, idx)), order, plan, (q_pts._version, s_pts._version, idx._version))
    return plan


def kpconv_inter_so3(x, q_pts, s_pts, idx, kernel_points, weights, kidx, ridx, sigma):
    """Fused matrix-core kernel (csrc/kpconv_mfma.hip) for channel counts that are multiples of (8, 32); otherwise the HIP gather of
    the slot-summed neighbourhood features (csrc/kpconv_so3.hip) + one library GEMM with the (36 Cin, Cout) weight matrix.
    x: (Ns, 6, Cin) tensor or BlockedFeatures (inference: written by group_norm_apply(blocked=True))."""
    blocked = isinstance(x, BlockedFeatures)
    xb = x.data if blocked else None
    x_amax = _amax_live(x) if blocked else None
    if blocked:
        Ns, A, Cin = x.shape
    else:
        x = _req(x.contiguous(), torch.float32, 'x', 3)
        Ns, A, Cin = x.shape
    q_pts, s_pts = _req(q_pts.contiguous(), torch.float32, 'q_pts', 2), _req(s_pts.contiguous(), torch.float32, 's_pts', 2)
    idx = _req(idx.contiguous(), torch.int64, 'neighb_inds', 2)
    P, NN = idx.shape
    Cout = weights.shape[-1]
    if A != 6 or tuple(weights.shape[:3]) != (6, 6, Cin) or Ns != s_pts.shape[0]:
        raise RuntimeError('kpconv_inter_so3: inconsistent shapes')
    kp, kt, rt = _host_table(kernel_points, torch.float32), _host_table(kidx, torch.int64), _host_table(ridx, torch.int64)
    path = _kpconv_use_matrix_core(Cin, Cout, P)
    fused_ok = bool(path) and Cin % 8 == 0 and Cout % 32 == 0 and Ns * 6 * Cin < 2 ** 31 and _builtin_slot_tables(kt, rt)
    po = point_order(q_pts) if (fused_ok and path is True and KPCONV_UNION and _kpconv_union_pays(Cin, Cout, q_pts is s_pts)) else None
    want = 0 if not blocked else x.kind
    if blocked and not (fused_ok and path != 'sums' and ((want == 2 and po is not None) or (want == 1 and po is None and Cin % 16 == 0))):
        x, blocked = x.plain().contiguous(), False              # (each fused kernel reads its own blocked layout only; the values, and their amax, stay)
    if fused_ok:
        if blocked:
            x = xb
        stream = _stream()
        Wp = _kpconv_weight_pieces(weights, Cin, Cout, stream)
        out = torch.empty((P, 6, Cout), dtype=torch.float32, device=x.device)
        # neighbour table (valid neighbours + 16 orbit weights each), then ONE kernel in which producer waves form the f16 hi / lo orbit sums
        # of a 16-point tile on the f32 matrix cores into LDS and consumer waves multiply them on the f16 matrix cores (csrc/kpconv_mfma.hip)
        tab = _kpconv_neighbor_table(q_pts, s_pts, idx, kernel_points, sigma, P, Ns, NN, stream)
        if path == 'sums':
            # two launches: the orbit sums as tile images in HBM, then the contraction (kept for A/B runs)
            Hs = torch.empty((lib().se3_kpconv_sums_bytes(P, Cin),), dtype=torch.uint8, device=x.device)
            check(lib().se3_kpconv_so3_gather_sums(x.data_ptr(), tab.data_ptr(), P, Ns, NN, Cin, Hs.data_ptr(), stream),
                  'se3_kpconv_so3_gather_sums')
            check(lib().se3_kpconv_so3_contract_f16(Hs.data_ptr(), Wp.data_ptr(), P, Cin, Cout, out.data_ptr(), stream),
                  'se3_kpconv_so3_contract_f16')
            return out
        if po is not None:
            # union-staged form (csrc/kpconv_union.hip): 16 spatial neighbours per workgroup, their distinct support rows read once
            order, G = po
            plan = _kpconv_union_plan(tab, q_pts, s_pts, idx, order, G, P, NN, stream)
            sbytes = lib().se3_kpconv_union_split_workspace_bytes(G, Cin, Cout) if KPCONV_SPLIT else 0
            sws = _zeroed_workspace(_kpconv_union_split_ws, x.device, stream, sbytes) if sbytes else None
            with _timed('kpconv_fused', 2.0 * 6 * P * 36 * Cin * Cout + 2.0 * P * NN * 16 * 6 * Cin):
                _check_counters(lib().se3_kpconv_so3_union(x.data_ptr(), tab.data_ptr(), plan.data_ptr(), G, P, Ns, NN, Cin, Cout, Wp.data_ptr(),
                                                           out.data_ptr(), sws.data_ptr() if sws is not None else None,
                                                           sws.numel() if sws is not None else 0, 1 if blocked else 0,
                                                           x_amax.data_ptr() if x_amax is not None else None, stream),
                                'se3_kpconv_so3_union', sws)
            return out
        sbytes = lib().se3_kpconv_fused_split_workspace_bytes(P, Cin, Cout) if KPCONV_SPLIT else 0
        sws = _zeroed_workspace(_kpconv_split_ws, x.device, stream, sbytes) if sbytes else None
        # (bench.py: event pair around the launch; algorithmic flops = contraction 2.6P.36Cin.Cout + the gather as a product 2.P.NN.16.6Cin)
        with _timed('kpconv_fused', 2.0 * 6 * P * 36 * Cin * Cout + 2.0 * P * NN * 16 * 6 * Cin):
            _check_counters(lib().se3_kpconv_so3_fused_scaled(x.data_ptr(), tab.data_ptr(), P, Ns, NN, Cin, Cout, Wp.data_ptr(), out.data_ptr(),
                                                              sws.data_ptr() if sws is not None else None, sws.numel() if sws is not None else 0,
                                                              1 if blocked else 0, x_amax.data_ptr() if x_amax is not None else None, stream),
                            'se3_kpconv_so3_fused', sws)
        return out
    G = torch.empty((P * 6, 36 * Cin), dtype=torch.float32, device=x.device)
    check(lib().se3_kpconv_so3_gather(q_pts.data_ptr(), s_pts.data_ptr(), idx.data_ptr(), x.data_ptr(), kp.data_ptr(),
                                      kt.data_ptr(), rt.data_ptr(), float(sigma), P, Ns, NN, Cin, G.data_ptr(), _stream()),
          'se3_kpconv_so3_gather')
    Kg = 36 * Cin
    if not torch.is_grad_enabled() and Kg % 4 == 0 and Kg <= 1024:
        # (the first layer of the backbone, Cin = 1: K = 36.)  The streaming kernel multiplies K rounded up to 32 wide: the rows of G are read
        # past their end into the next row (finite values) against weight columns that are zero; the last row reads zeros (buffer bounds).
        Kp = (Kg + 31) // 32 * 32
        Wt = _padded_transposed_weight(weights, Kg, Kp, Cout)
        out = torch.empty((P * 6, Cout), dtype=torch.float32, device=x.device)
        stream = _stream()
        check(lib().se3_linear_stream(G.data_ptr(), P * 6, Kp, Kg, _linear_weight_pieces(Wt, stream).data_ptr(), None, Cout, 0, out.data_ptr(), Cout,
                                      stream), 'se3_linear_stream (KPConv slot sums)')
        return out.view(P, 6, Cout)
    return mm(G, weights.reshape(Kg, Cout)).view(P, 6, Cout)


_padded_weight_cache = {}


def _padded_transposed_weight(weights, Kg, Kp, Cout):
    """(6, 6, Cin, Cout) KPConv weights as a dense layer's (Cout, Kp) weight: transposed, zero columns from Kg to Kp; kept per weight version."""
    key = (weights.data_ptr(), Kg, Kp, Cout, weights.device.index)
    hit = _padded_weight_cache.get(key)
    if hit is not None and hit[0]() is weights and hit[1] == weights._version:
        return hit[2].get()[0]
    with torch.no_grad():
        Wt = torch.zeros((Cout, Kp), dtype=torch.float32, device=weights.device)
        Wt[:, :Kg] = weights.detach().reshape(Kg, Cout).t()
    with _TIMING_LOCK:
        if len(_padded_weight_cache) > 64:
            _padded_weight_cache.clear()
        _padded_weight_cache[key] = (weakref.ref(weights), weights._version, _Shared(Wt), _fingerprint(weights))
    return Wt


def key_stride(M):
    """Row stride of key-major operands (relative-position logits, transposed values): keys padded to a multiple of 32."""
    return (M + 31) // 32 * 32


def _rows_view(t, name, C=None):
    """([A,] rows, C') float32 GPU view with unit last stride and 16-byte aligned rows -> (tensor3, anchors, rows, row_stride,
    anchor_stride); column blocks / row ranges of a wider tensor are accepted as they are (no copy)."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError('%s must be a float32 GPU tensor' % name)
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.stride(-1) != 1 or t.stride(-2) % 4 or (t.shape[0] > 1 and t.stride(0) % 4) or t.data_ptr() % 16:
        t = t.contiguous()
    if C is not None and t.shape[-1] != C:
        raise RuntimeError('%s: last dim %d != %d' % (name, t.shape[-1], C))
    return t, t.shape[0], t.shape[1], t.stride(1), (t.stride(0) if t.shape[0] > 1 else 0)


def attention(q, k, vt, bias, num_heads, out=None, v_shared=False, tag=None):
    """softmax_m((q.k [+ bias]) / sqrt(d)) v through se3_attention_fwd.  q ([A,] N, C), k ([A,] M, C): strided views are
    fine; vt ([A,] C, >= ceil32(M)) transposed values; bias (A*H, N, Mp) or None; out: optional ([A,] N, C) view with
    contiguous rows.  A = number of output anchors = anchors of vt (q/k with one anchor are broadcast)."""
    H = num_heads
    q3, Aq, N, q_rs, q_sa = _rows_view(q, 'q')
    C = q3.shape[-1]
    k3, Ak, M, k_rs, k_sa = _rows_view(k, 'k', C)
    v3 = vt if vt.dim() == 3 else vt.unsqueeze(0)
    if not v3.is_cuda or v3.dtype != torch.float32 or v3.stride(-1) != 1 or v3.stride(-2) % 4 or v3.data_ptr() % 16:
        v3 = v3.contiguous()
    Av, v_rs = v3.shape[0], v3.stride(1)
    if v3.shape[1] != C or v3.shape[2] < M:
        raise RuntimeError('attention: transposed values %s do not match C=%d, M=%d' % (tuple(v3.shape), C, M))
    A = max(Aq, Ak, Av)
    Mp = key_stride(M)
    if bias is not None and (tuple(bias.shape) != (A * H, N, Mp) or not bias.is_contiguous()):
        raise RuntimeError('attention: bias must be a contiguous (A*H, N, %d) tensor' % Mp)
    if out is None:
        out = torch.empty((A, N, C), dtype=torch.float32, device=q3.device)
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    if tuple(o3.shape) != (A, N, C) or o3.stride(-1) != 1 or o3.stride(-2) != C:
        raise RuntimeError('attention: out must be (A, N, C) with contiguous rows')
    with _timed('attention_kernel' if tag is None else 'attention_kernel@' + tag, 0):
        check(lib().se3_attention_fwd(q3.data_ptr(), k3.data_ptr(), v3.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      A, N, M, C, H, q_rs, k_rs, v_rs, q_sa if Aq > 1 else 0, k_sa if Ak > 1 else 0,
                                      v3.stride(0) if Av > 1 else 0, o3.stride(0) if A > 1 else 0, Mp,
                                      1.0 / math.sqrt(C // H), o3.data_ptr(), _stream()), 'se3_attention_fwd')
    return out


def rpe_bias(qp, qe, emb, eq_emb, num_heads):
    """Relative-position logits (A*H, N, Mp) from the folded queries qp ([A,] N, H*C) [and qe ([A,] N, 4*H)] (strided views
    of one projection are fine: qp and qe must share row / anchor strides) and the embeddings emb (N, M, C), eq_emb (A, N, M, 4)."""
    H = num_heads
    emb = _req(emb, torch.float32, 'embed_qk', 3)
    N, M, C = emb.shape
    qp3, A, Nq, rs, sa = _rows_view(qp, 'qp', H * C)
    if Nq != N:
        raise RuntimeError('rpe_bias: %d folded query rows for an (N=%d, M, C) embedding' % (Nq, N))
    qe_ptr = None
    eq_ptr = None
    if qe is not None:
        qe3, Ae, Ne, rs_e, sa_e = _rows_view(qe, 'qe', 4 * H)
        if (Ae, Ne, rs_e, sa_e) != (A, N, rs, sa):
            raise RuntimeError('rpe_bias: qe must be a column block of the same projection as qp')
        eq_emb = _req(eq_emb, torch.float32, 'embed_eq', 4)
        if tuple(eq_emb.shape) != (A, N, M, 4):
            raise RuntimeError('rpe_bias: equivariant embedding shape %s' % (tuple(eq_emb.shape),))
        qe_ptr, eq_ptr = qe3.data_ptr(), eq_emb.data_ptr()
    if A * H > 32 or C % 16:
        raise RuntimeError('rpe_bias: anchors*heads must be <= 32 and C a multiple of 16')
    Mp = key_stride(M)
    bias = torch.empty((A * H, N, Mp), dtype=torch.float32, device=emb.device)
    # algorithmic bytes of one self-attention call: q, k, v in + out, the embedding, the eq-embedding;
    # booked on this (embedding-streaming) launch, the attention launch of the same call is booked with 0 bytes
    survey_bytes = 4 * (4 * A * N * C + N * M * C + (A * N * M * 4 if qe is not None else 0))
    with _timed('rpe_bias_kernel', survey_bytes, 'eq' if qe is not None else 'inv'):
        check(lib().se3_rpe_bias_fwd(qp3.data_ptr(), qe_ptr, rs, sa, emb.data_ptr(), eq_ptr, N, M, C, A * H, H, Mp,
                                     bias.data_ptr(), _stream()), 'se3_rpe_bias_fwd')
    return bias


def neighbor_table_trim(full, width, pair_row_ends, pair_widths):
    """HIP (csrc/radius_neighbors.hip): full (rows, W) int64 -> (rows, width) with the columns past each pair's own width marked -1."""
    full = _req(full, torch.int64, 'full', 2)
    rows, W = full.shape
    out = torch.empty((rows, width), dtype=torch.int64, device=full.device)
    n = len(pair_row_ends)
    check(lib().se3_neighbor_table_trim(full.data_ptr(), rows, W, int(width), _i64_array(pair_row_ends), (ctypes.c_int * n)(*[int(w) for w in pair_widths]),
                                        n, out.data_ptr(), _stream()), 'se3_neighbor_table_trim')
    return out


def to_device(values, dtype, device):
    """Small host list -> device tensor through pinned memory and an asynchronous copy.  `torch.tensor(values, device=...)`
    copies from pageable memory, which makes the host wait for everything queued on the stream (a full synchronisation per
    index table)."""
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


def _i64_array(values):
    return (ctypes.c_int64 * len(values))(*[int(v) for v in values])


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def stack_bias_offsets(lengths, key_lengths, AH):
    """Float offsets of the per-cloud (A*H, N_c, ceil32(M_c)) logits blocks in one workspace, and its total size."""
    offs, total = [], 0
    for n, m in zip(lengths, key_lengths):
        offs.append(total)
        total += AH * int(n) * key_stride(int(m))
    return offs, total


def _embedding_list(embs, C):
    """(bytes per element, contiguous GPU tensors): the geometric embeddings of a call are all float32 or all bfloat16."""
    dt = embs[0].dtype
    if dt not in (torch.float32, torch.bfloat16) or (dt == torch.bfloat16 and C % 32):
        raise RuntimeError('embed_qk must be float32, or bfloat16 with C a multiple of 32')
    return (4 if dt == torch.float32 else 2), [_req(e, dt, 'embed_qk', 3) for e in embs]


def rpe_bias_stack(qp, qe, embs, eq_embs, starts, lengths, num_heads):
    """Stack mode of rpe_bias: qp ([A,] R, H*C) [qe ([A,] R, 4*H)] hold the folded queries of all clouds (cloud c = rows
    starts[c] .. + lengths[c]); embs[c] (N_c, N_c, C), eq_embs[c] (A, N_c, N_c, 4) or None.  ONE launch; returns the flat
    logits workspace and the per-cloud block offsets (block c: (A*H, N_c, ceil32(N_c)))."""
    H = num_heads
    C = embs[0].shape[-1]
    qp3, A, R, rs, sa = _rows_view(qp, 'qp', H * C)
    has_eq = qe is not None
    qe_ptr = None
    if has_eq:
        qe3, Ae, Re, rs_e, sa_e = _rows_view(qe, 'qe', 4 * H)
        if (Ae, Re, rs_e, sa_e) != (A, R, rs, sa):
            raise RuntimeError('rpe_bias_stack: qe must be a column block of the same projection as qp')
        qe_ptr = qe3.data_ptr()
    if A * H > 32 or C % 16:
        raise RuntimeError('rpe_bias_stack: anchors*heads must be <= 32 and C a multiple of 16')
    esize, embs = _embedding_list(embs, C)
    survey_bytes = 0
    for c, (e, n) in enumerate(zip(embs, lengths)):
        if tuple(e.shape) != (n, n, C) or starts[c] + n > R:
            raise RuntimeError('rpe_bias_stack: cloud %d: embedding %s for %d rows at %d of %d' % (c, tuple(e.shape), n, starts[c], R))
        survey_bytes += 4 * (4 * A * n * C + (A * n * n * 4 if has_eq else 0)) + esize * n * n * C
    eqs = None
    if has_eq:
        eqs = [_req(e, torch.float32, 'embed_eq', 4) for e in eq_embs]
        for e, n in zip(eqs, lengths):
            if tuple(e.shape) != (A, n, n, 4):
                raise RuntimeError('rpe_bias_stack: equivariant embedding shape %s' % (tuple(e.shape),))
    offs, total = stack_bias_offsets(lengths, lengths, A * H)
    bias = torch.empty((total,), dtype=torch.float32, device=qp3.device)
    with _timed('rpe_bias_kernel', survey_bytes, 'eq' if has_eq else 'inv'):
        entry = lib().se3_rpe_bias_stack_fwd if esize == 4 else lib().se3_rpe_bias_stack_bf16_fwd
        check(entry(qp3.data_ptr(), qe_ptr, rs, sa, _ptr_array(embs), _ptr_array(eqs) if has_eq else None, _i64_array(starts),
                    _i64_array(lengths), _i64_array(lengths), _i64_array(offs), len(embs), C, A * H, H, bias.data_ptr(),
                    _stream()), 'se3_rpe_bias_stack_fwd')
    return bias, offs


def attention_stack(q, k, vt, bias, bias_offsets, q_starts, q_lengths, k_starts, k_lengths, num_heads, out, tag=None):
    """Stack mode of attention: q ([A,] R, C), k ([A,] R, C) packed rows, vt ([A,] C, Rk) transposed values addressed by key
    column, out ([A,] R, C) with contiguous rows; cloud c: queries q_starts[c] .. + q_lengths[c], keys k_starts[c] .. +
    k_lengths[c] (k_starts multiples of 4, ceil32(k_lengths[c]) columns readable).  bias: flat workspace of rpe_bias_stack or None."""
    H = num_heads
    q3, Aq, R, q_rs, q_sa = _rows_view(q, 'q')
    C = q3.shape[-1]
    k3, Ak, Rk, k_rs, k_sa = _rows_view(k, 'k', C)
    v3 = vt if vt.dim() == 3 else vt.unsqueeze(0)
    if not v3.is_cuda or v3.dtype != torch.float32 or v3.stride(-1) != 1 or v3.stride(-2) % 4 or v3.data_ptr() % 16:
        v3 = v3.contiguous()
    Av, v_rs = v3.shape[0], v3.stride(1)
    if v3.shape[1] != C:
        raise RuntimeError('attention_stack: transposed values %s do not match C=%d' % (tuple(v3.shape), C))
    A = max(Aq, Ak, Av)
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    if o3.shape[0] != A or o3.shape[2] != C or o3.stride(-1) != 1 or o3.stride(-2) != C or o3.dtype != torch.float32:
        raise RuntimeError('attention_stack: out must be (A, rows, C) float32 with contiguous rows')
    for c in range(len(q_starts)):
        if q_starts[c] + q_lengths[c] > min(R, o3.shape[1]) or k_starts[c] + k_lengths[c] > Rk or \
                k_starts[c] + key_stride(k_lengths[c]) > v3.shape[2]:
            raise RuntimeError('attention_stack: cloud %d exceeds the packed rows / value columns' % c)
    if bias is not None and (not bias.is_cuda or bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise RuntimeError('attention_stack: bias must be a contiguous float32 GPU workspace')
    with _timed('attention_kernel' if tag is None else 'attention_kernel@' + tag, 0):
        check(lib().se3_attention_stack_fwd(q3.data_ptr(), k3.data_ptr(), v3.data_ptr(), bias.data_ptr() if bias is not None else None,
                                            _i64_array(q_starts), _i64_array(q_lengths), _i64_array(k_starts), _i64_array(k_lengths),
                                            _i64_array(bias_offsets) if bias is not None else None, len(q_starts), A, C, H,
                                            q_rs, k_rs, v_rs, q_sa if Aq > 1 else 0, k_sa if Ak > 1 else 0,
                                            v3.stride(0) if Av > 1 else 0, o3.stride(0) if A > 1 else 0,
                                            1.0 / math.sqrt(C // H), o3.data_ptr(), *_attention_pieces(A, k_starts, k_lengths, C, v_rs, q3.device),
                                            _stream()), 'se3_attention_stack_fwd')
    return out


def rpe_self_attention_stack(proj, offs, vt, embs, eq_embs, starts, lengths, num_heads, out):
    """The stack-mode RPE self-attention call: proj ([A,] R, 2C + HC [+ 4H]) is the stacked projection [q | k | W_p^T q | W_eq^T q]
    of the packed rows (column offsets `offs`), vt ([A,] C, R) the transposed values, embs[c] (N_c, N_c, C), eq_embs[c]
    (A, N_c, N_c, 4) or None; out ([A,] R, C) receives the rows of every cloud.  Both kernels (relative-position logits, then
    softmax.V) are launched back to back from one C call."""
    H = num_heads
    C = embs[0].shape[-1]
    p3, A, R, rs, sa = _rows_view(proj, 'proj')
    has_eq = eq_embs is not None and eq_embs[0] is not None
    if offs['qp'] + H * C > p3.shape[-1] or (has_eq and offs['qe'] + 4 * H > p3.shape[-1]):
        raise RuntimeError('rpe_self_attention_stack: projection narrower than its column offsets')
    if A * H > 32 or C % 16:
        raise RuntimeError('rpe_self_attention_stack: anchors*heads must be <= 32 and C a multiple of 16')
    v3 = vt if vt.dim() == 3 else vt.unsqueeze(0)
    if not v3.is_cuda or v3.dtype != torch.float32 or v3.stride(-1) != 1 or v3.stride(-2) % 4 or v3.data_ptr() % 16 or \
            v3.shape[0] != A or v3.shape[1] != C:
        raise RuntimeError('rpe_self_attention_stack: vt must be a float32 (A, C, rows) GPU tensor with 16-byte aligned rows')
    o3 = out if out.dim() == 3 else out.unsqueeze(0)
    if o3.shape[0] != A or o3.shape[2] != C or o3.stride(-1) != 1 or o3.stride(-2) != C or o3.dtype != torch.float32:
        raise RuntimeError('rpe_self_attention_stack: out must be (A, rows, C) float32 with contiguous rows')
    esize, embs = _embedding_list(embs, C)
    survey_bytes, total = 0, 0
    for c, (e, n) in enumerate(zip(embs, lengths)):
        if tuple(e.shape) != (n, n, C) or starts[c] % 4 or starts[c] + n > min(R, o3.shape[1]) or \
                starts[c] + key_stride(n) > v3.shape[2]:
            raise RuntimeError('rpe_self_attention_stack: cloud %d: embedding %s, %d rows at %d of %d' % (c, tuple(e.shape), n, starts[c], R))
        survey_bytes += 4 * (4 * A * n * C + (A * n * n * 4 if has_eq else 0)) + esize * n * n * C
        total += A * H * n * key_stride(n)
    eqs = None
    if has_eq:
        eqs = [_req(e, torch.float32, 'embed_eq', 4) for e in eq_embs]
        for e, n in zip(eqs, lengths):
            if tuple(e.shape) != (A, n, n, 4):
                raise RuntimeError('rpe_self_attention_stack: equivariant embedding shape %s' % (tuple(e.shape),))
    logits = torch.empty((total,), dtype=torch.float32, device=p3.device)
    base = p3.data_ptr()
    col = lambda name: base + 4 * offs[name]
    entry = lib().se3_rpe_self_attention_stack_fwd if esize == 4 else lib().se3_rpe_self_attention_stack_bf16_fwd
    if KERNEL_TIMINGS is not None:       # bench.py pairs these with the library's per-launch HIP events, in call order
        with _TIMING_LOCK:               # (host threads with one stream each: the record and the launches stay in one order)
            KERNEL_TIMINGS.setdefault('rpe_self_attention_calls', []).append((survey_bytes, ('eq' if has_eq else 'inv') + ('' if esize == 4 else '_bf16')))
            check(entry(col('q'), col('k'), v3.data_ptr(), col('qp'), col('qe') if has_eq else None, rs, sa, v3.stride(1),
                        v3.stride(0) if A > 1 else 0, _ptr_array(embs), _ptr_array(eqs) if has_eq else None, _i64_array(starts),
                        _i64_array(lengths), len(embs), A, C, H, logits.data_ptr(), o3.stride(0) if A > 1 else 0, o3.data_ptr(),
                        *_attention_pieces(A, starts, lengths, C, v3.stride(1), p3.device), _stream()), 'se3_rpe_self_attention_stack_fwd')
        return out
    check(entry(col('q'), col('k'), v3.data_ptr(), col('qp'), col('qe') if has_eq else None, rs, sa, v3.stride(1),
                v3.stride(0) if A > 1 else 0, _ptr_array(embs), _ptr_array(eqs) if has_eq else None, _i64_array(starts),
                _i64_array(lengths), len(embs), A, C, H, logits.data_ptr(), o3.stride(0) if A > 1 else 0, o3.data_ptr(),
                *_attention_pieces(A, starts, lengths, C, v3.stride(1), p3.device), _stream()), 'se3_rpe_self_attention_stack_fwd')
    return out


ATTENTION_F16 = True          # False: q.k and P.v on the f32 matrix cores (A/B runs; the f16 hi / lo form has the error of an f32 product)


def _attention_pieces(A, k_starts, k_lengths, C, v_row_stride, device):
    """(workspace pointer, bytes) for the f16 hi / lo pieces of K and V^T of one stack-mode attention call: one buffer per stream."""
    if not ATTENTION_F16:
        return None, 0
    rows = max(int(s) + int(n) for s, n in zip(k_starts, k_lengths))
    nbytes = lib().se3_attention_kv_pieces_bytes(int(A), rows, int(C), int(v_row_stride))
    if nbytes == 0:
        return None, 0
    stream = _stream()
    key = (device, stream.value, 'attn')
    ws = _gn_workspace.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(nbytes, 1 << 22),), dtype=torch.uint8, device=device)
        _gn_workspace[key] = ws
    return ws.data_ptr(), ws.numel()


def cross_attention(q, k, vt, num_heads):
    """HIP: plain cross attention; vt (C, Mp) or (A, C, Mp) transposed values (per-anchor values share the scores)."""
    out = attention(q, k, vt, None, num_heads)
    return out[0] if vt.dim() == 2 else out


def cross_attention_eq(q, k, vt, num_heads, mode, trace_idx):
    """HIP: global anchor-pair statistics (se3_cross_eq_stats) + weighted per-pair softmax.V (se3_cross_eq_apply)."""
    q = _req(q.contiguous(), torch.float32, 'q', 3)
    k = _req(k.contiguous(), torch.float32, 'k', 3)
    vt = _req(vt.contiguous(), torch.float32, 'vt', 3)
    A, N, C = q.shape
    M, H = k.shape[1], num_heads
    scale = 1.0 / math.sqrt(C // H)
    P = (N + 31) // 32
    partial = torch.empty((A * A, P), dtype=torch.float32, device=q.device)
    nparts = ctypes.c_int(0)
    check(lib().se3_cross_eq_stats(q.data_ptr(), k.data_ptr(), A, N, M, C, H, scale, partial.data_ptr(),
                                   ctypes.byref(nparts), _stream()), 'se3_cross_eq_stats')
    if mode not in ('a_soft', 'r_soft'):
        raise RuntimeError('cross_attention_eq: mode %r' % (mode,))
    R = trace_idx.shape[0]
    mix = torch.empty((A, A), dtype=torch.float32, device=q.device)
    ret = torch.empty((A, A) if mode == 'a_soft' else (R,), dtype=torch.float32, device=q.device)
    trace_idx = _req(trace_idx.contiguous(), torch.int64, 'trace_idx', 2)
    check(lib().se3_cross_eq_mix(partial.data_ptr(), P, A, N, M, 0 if mode == 'a_soft' else 1, trace_idx.data_ptr(), R,
                                 mix.data_ptr(), ret.data_ptr(), _stream()), 'se3_cross_eq_mix')
    out = torch.empty((A, N, C), dtype=torch.float32, device=q.device)
    Mp = key_stride(M)
    if tuple(vt.shape) != (A, C, Mp):
        raise RuntimeError('cross_attention_eq: transposed values must be (A, C, %d)' % Mp)
    check(lib().se3_cross_eq_apply(q.data_ptr(), k.data_ptr(), vt.data_ptr(), mix.data_ptr(), A, N, M, C, H,
                                   Mp, scale, out.data_ptr(), _stream()), 'se3_cross_eq_apply')
    return out, ret, mix


# equivariant cross attention of a batch on the bf16 matrix cores at f32 accuracy (SE3_CROSS_EQ=f32 forces the f32 MFMA kernels)
CROSS_EQ_BF16X6 = os.environ.get('SE3_CROSS_EQ', 'bf16x6') != 'f32'
_pair_rows_cache = {}


def _pair_rows(starts, lengths, device):
    """(P * W,) row indices and a (1, P, W, 1) 0/1 mask that cut the packed rows of P clouds into P windows of W rows."""
    key = (tuple(int(s) for s in starts), tuple(int(n) for n in lengths), str(device))
    hit = _pair_rows_cache.get(key)
    if hit is None:
        W = max(key[1])
        idx = to_device([s + min(j, n - 1) for s, n in zip(key[0], key[1]) for j in range(W)], torch.int64, device)
        mask = to_device([1.0 if j < n else 0.0 for n in key[1] for j in range(W)], torch.float32, device)
        hit = (_Shared(idx, mask), W)
        if len(_pair_rows_cache) > 64:
            _pair_rows_cache.clear()
        _pair_rows_cache[key] = hit
    idx, mask = hit[0].get()
    return idx, mask.view(1, len(key[1]), hit[1], 1), hit[1]


GRAM_KERNEL = True            # False: index_select + mask + batched library GEMM (A/B runs, tests)


def _gram_per_pair(x, starts, lengths):
    """x (A, R, C) packed rows -> (A, P, C, C): X_p^T X_p over the rows of every pair (csrc/attention.hip: gram_stack_kernel; other channel
    counts / strided rows: index_select + library GEMM)."""
    A, P, C = x.shape[0], len(starts), x.shape[2]
    if GRAM_KERNEL and C in (128, 256) and P <= 16 and x.dtype == torch.float32 and x.stride(2) == 1 and x.stride(1) == C:
        out = torch.empty((A, P, C, C), dtype=torch.float32, device=x.device)
        check(lib().se3_gram_stack(x.data_ptr(), A, C, x.stride(0) if A > 1 else 0, _i64_array(starts), _i64_array(lengths), P, out.data_ptr(),
                                   _stream()), 'se3_gram_stack')
        return out
    idx, mask, W = _pair_rows(starts, lengths, x.device)
    win = x.index_select(1, idx).view(A, P, W, C)
    win.mul_(mask)                                   # rows past a cloud's length -> 0 (the mask is 0 / 1: masking both operands is masking one)
    win = win.view(A * P, W, C)
    return torch.bmm(win.transpose(1, 2), win).view(A, P, C, C)


def cross_eq_groups(A, q_lengths, num_heads, C, k_starts, vt):
    """Key-anchor groups of cross_attention_eq_stack for this call: with few pairs one workgroup per (128 queries, head, anchor) leaves
    most compute units idle while every wave walks A x key-tiles dependent steps, so the key anchors are divided over 3 (or 2) workgroups
    whose partial results lie side by side along the channels of the output.  1 = the plain form."""
    H = int(num_heads)
    if not CROSS_EQ_BF16X6 or C // H != 64 or C % H or A > 6 or vt.stride(1) % 16 or vt.stride(2) != 1 or any(int(s) % 16 for s in k_starts):
        return 1
    wgs = sum((int(n) + 127) // 128 for n in q_lengths) * H * A
    forced = int(os.environ.get('SE3_EQ_GROUPS', '0'))          # A/B runs: 2 or 3
    if forced in (2, 3) and A % forced == 0:
        return forced
    for G in (3, 2):
        if A % G == 0 and wgs * G <= 320:
            return G
    return 1


def cross_attention_eq_stack(q, k, vt, q_starts, q_lengths, k_starts, k_lengths, num_heads, mode, trace_idx, out, groups=1):
    """HIP: equivariant cross attention of all pairs of a batch (se3_cross_eq_stack_fwd).  q (A, Rq, C), k (A, Rk, C) packed rows,
    vt (A, C, Rk) transposed values, out (A, Rq, C) (rows outside the pairs are left untouched).  Returns the per-pair mixing
    matrices (P, A, A) and weights (P, A*A) ('a_soft') / (P, R) ('r_soft').  groups = G > 1 (cross_eq_groups): out (A, Rq, G * C)
    contiguous, the result is the sum of its G channel blocks (the caller folds it into the next dense layer: stacked_weight)."""
    q = _req(q, torch.float32, 'q', 3)
    k = _req(k, torch.float32, 'k', 3)
    vt = _req(vt, torch.float32, 'vt', 3)
    out = _req(out, torch.float32, 'out', 3)
    A, Rq, C = q.shape
    groups = int(groups)
    if k.shape[0] != A or k.shape[2] != C or tuple(vt.shape[:2]) != (A, C) or tuple(out.shape) != (A, Rq, groups * C):
        raise RuntimeError('cross_attention_eq_stack: shapes q %s k %s vt %s out %s' % (tuple(q.shape), tuple(k.shape), tuple(vt.shape), tuple(out.shape)))
    if groups > 1 and not (CROSS_EQ_BF16X6 and out.is_contiguous()):
        raise RuntimeError('cross_attention_eq_stack: key-anchor groups need the f16 form and a contiguous output')
    if mode not in ('a_soft', 'r_soft'):
        raise RuntimeError('cross_attention_eq_stack: mode %r' % (mode,))
    P = len(q_starts)
    for p in range(P):
        if q_starts[p] + q_lengths[p] > Rq or k_starts[p] + k_lengths[p] > k.shape[1] or k_starts[p] + key_stride(k_lengths[p]) > vt.shape[2]:
            raise RuntimeError('cross_attention_eq_stack: pair %d exceeds the packed rows / value columns' % p)
    trace_idx = _req(trace_idx.contiguous(), torch.int64, 'trace_idx', 2)
    R = trace_idx.shape[0]
    dev = q.device
    # anchor-pair statistics sum_{n,m} (mean_h S[a,e,h,n,m])^2: the head mean of the per-head dot products is the dot product
    # over all C channels, so the sum is (scale / H)^2 <Q_a^T Q_a, K_e^T K_e>_F -- two Gram products per pair instead of the
    # (A*A, N, M) score pass of se3_cross_eq_stats (vanilla_transformer.py:380-389,425-426 computes the scores themselves)
    f = 1.0 / (math.sqrt(C // int(num_heads)) * int(num_heads))
    gq, gk = _gram_per_pair(q, q_starts, q_lengths), _gram_per_pair(k, k_starts, k_lengths)
    partial = torch.empty((P, A, A), dtype=torch.float32, device=dev)
    check(lib().se3_gram_frobenius(gq.data_ptr(), gk.data_ptr(), A, P, C * C, f * f, partial.data_ptr(), _stream()), 'se3_gram_frobenius')
    mix = torch.empty((P, A, A), dtype=torch.float32, device=dev)
    weights = torch.empty((P, A * A if mode == 'a_soft' else R), dtype=torch.float32, device=dev)
    if CROSS_EQ_BF16X6 and q.stride(1) == C and k.stride(1) == C and q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1:
        # bf16 matrix cores at f32 accuracy (csrc/attention.hip: cross_eq_apply_stack_x6_kernel); shapes it does not take are forwarded
        # to the f32 kernels by the entry point itself
        stream = _stream()
        ws_bytes = lib().se3_cross_eq_x6_workspace_bytes(A, Rq, k.shape[1], C, vt.stride(1))
        key = (dev, stream.value, 'x6')
        ws = _gn_workspace.get(key)
        if ws is None or ws.numel() < ws_bytes:
            ws = torch.empty((max(ws_bytes, 1 << 24),), dtype=torch.uint8, device=dev)
            _gn_workspace[key] = ws
        check(lib().se3_cross_eq_stack_x6_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), _i64_array(q_starts), _i64_array(q_lengths),
                                              _i64_array(k_starts), _i64_array(k_lengths), P, A, C, int(num_heads), Rq, k.shape[1],
                                              q.stride(0), k.stride(0), vt.stride(1), vt.stride(0), 0 if mode == 'a_soft' else 1,
                                              trace_idx.data_ptr(), R, 1, partial.data_ptr(), mix.data_ptr(), weights.data_ptr(),
                                              out.data_ptr(), groups, out.stride(0), ws.data_ptr(), ws.numel(), stream),
              'se3_cross_eq_stack_x6_fwd')
        return mix, weights
    if groups > 1:
        raise RuntimeError('cross_attention_eq_stack: key-anchor groups with strided q / k rows')
    check(lib().se3_cross_eq_stack_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), _i64_array(q_starts), _i64_array(q_lengths),
                                       _i64_array(k_starts), _i64_array(k_lengths), P, A, C, int(num_heads), q.stride(0), k.stride(0),
                                       vt.stride(1), vt.stride(0), 0 if mode == 'a_soft' else 1, trace_idx.data_ptr(), R, 1,
                                       partial.data_ptr(), mix.data_ptr(), weights.data_ptr(), out.data_ptr(), _stream()),
          'se3_cross_eq_stack_fwd')
    return mix, weights


_EMB_D_RANGE, _EMB_D_PER_UNIT = 64.0, 64.0        # distance-index table: [0, 64) index units, 64 entries per unit
_EMB_A_PER_UNIT = 32.0                             # angle-index table: a 32-channel slice (418 entries x 256 B) fits in LDS; Hermite error h^4 / 384 |f''''| ~ 2.5e-9 |f''''|
_emb_table_cache = {}


def _embedding_table(weight, bias, div_term, x_max, per_unit):
    """(entries, C, 2): f(x) = W emb(x) + b and f'(x) at x = j / per_unit, j = 0 .. x_max * per_unit, built and VALIDATED ON THE
    DEVICE by se3_embedding_table_refresh in front of every use: the kernel hashes the current weight values and rebuilds the
    table when they differ from the ones it was built from.  The host cache below only holds the buffers (keyed by the
    storage addresses): identity / version bookkeeping of the Parameters plays no role, so `p.data.copy_()`, `module.to()`,
    `load_state_dict` and optimizer steps can never leave a stale table behind.  One table + validation state PER LAUNCH STREAM: the
    refresh kernels count their completed workgroups in the state block, which two streams rebuilding the same table at the same time
    (first use under `bench.py --inflight 2`) would mix -- calls on one stream are ordered, so a per-stream table has one writer."""
    weight, bias, div_term = weight.detach(), bias.detach(), div_term.detach()
    for t, name in ((weight, 'weight'), (bias, 'bias'), (div_term, 'div_term')):
        _req(t, torch.float32, 'embedding ' + name)
    C = weight.shape[0]
    n = int(math.ceil(x_max * per_unit)) + 2
    stream = _stream()
    key = (str(weight.device), stream.value, weight.data_ptr(), bias.data_ptr(), div_term.data_ptr(), C, n, float(per_unit))
    hit = _emb_table_cache.get(key)
    if hit is None:
        hit = (torch.empty((n, C, 2), dtype=torch.float32, device=weight.device),
               torch.zeros((lib().se3_embedding_table_state_bytes(),), dtype=torch.uint8, device=weight.device))
        if len(_emb_table_cache) > 32:
            _emb_table_cache.clear()
        _emb_table_cache[key] = hit
    tab, state = hit
    check(lib().se3_embedding_table_refresh(weight.data_ptr(), bias.data_ptr(), div_term.data_ptr(), C, n, float(per_unit),
                                            tab.data_ptr(), state.data_ptr(), stream), 'se3_embedding_table_refresh')
    return tab


def knn3_stack(points, lengths):
    """HIP (csrc/partition.hip): the 3 nearest other points of every point of several stacked clouds, indices local to the cloud."""
    points = _req(points.contiguous(), torch.float32, 'points', 2)
    if sum(lengths) != points.shape[0]:
        raise RuntimeError('knn3_stack: lengths do not add up to the stacked points')
    knn = torch.empty((points.shape[0], 3), dtype=torch.int64, device=points.device)
    check(lib().se3_knn3_stack(points.data_ptr(), _i64_array(lengths), len(lengths), knn.data_ptr(), _stream()), 'se3_knn3_stack')
    return knn


_emb_ws = {}


def _emb_workspace(device, nbytes):
    """Per-pair record scratch of the embedding kernels, one buffer per launch stream (calls on a stream are ordered)."""
    key = (device, _stream().value)
    ws = _emb_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(nbytes, 1 << 24),), dtype=torch.uint8, device=device)
        _emb_ws[key] = ws
    return ws


def embedding_tables(div_term, w_d, b_d, w_a, b_a, sigma_a):
    """The two validated tables (distance, angle) of a geometric embedding: pass them as `tables=` to several geometric_embedding calls of
    one forward (all clouds of a batch) to validate the weights once instead of once per cloud."""
    return (_embedding_table(w_d, b_d, div_term, _EMB_D_RANGE, _EMB_D_PER_UNIT),
            _embedding_table(w_a, b_a, div_term, 180.0 / sigma_a + 1.0, _EMB_A_PER_UNIT))


def geometric_embedding(points, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, k, wigner_d1=None, dtype=torch.float32, knn=None, tables=None):
    """HIP (csrc/geo_embedding.hip).  Returns emb (N, N, C), or (emb, eq_emb (A, N, N, 4)) when wigner_d1 is given.
    dtype: torch.float32, or torch.bfloat16 for the 'bf16 attention' mode (emb stored rounded; eq_emb stays float32)."""
    points = _req(points.contiguous(), torch.float32, 'points', 2)
    if k != 3:
        raise RuntimeError('geometric_embedding: angle_k must be 3 on the HIP path')
    N, C = points.shape[0], w_d.shape[0]
    # 3 nearest other points (the reference takes top-(k+1) of the distance map and drops the first column)
    if knn is None:
        knn = torch.empty((N, 3), dtype=torch.int64, device=points.device)
        check(lib().se3_knn3(points.data_ptr(), N, knn.data_ptr(), _stream()), 'se3_knn3')
    else:                            # precomputed for all clouds of a batch (knn3_stack)
        knn = _req(knn, torch.int64, 'knn', 2)
        if tuple(knn.shape) != (N, 3):
            raise RuntimeError('geometric_embedding: knn must be (N, 3)')
    tab_d, tab_a = tables if tables is not None else embedding_tables(div_term, w_d, b_d, w_a, b_a, sigma_a)
    if dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError('geometric_embedding: dtype must be float32 or bfloat16')
    emb = torch.empty((N, N, C), dtype=dtype, device=points.device)
    eq = None
    A = 0
    if wigner_d1 is not None:
        A = wigner_d1.shape[0]
        eq = torch.empty((A, N, N, 4), dtype=torch.float32, device=points.device)
        wigner_d1 = wigner_d1.detach().contiguous()
    entry = lib().se3_geo_embedding_fwd if dtype == torch.float32 else lib().se3_geo_embedding_bf16_fwd
    ws = _emb_workspace(points.device, lib().se3_geo_embedding_workspace_bytes(N))
    check(entry(points.data_ptr(), knn.data_ptr(), N, C, tab_d.data_ptr(), tab_d.shape[0], _EMB_D_PER_UNIT, tab_a.data_ptr(),
                tab_a.shape[0], _EMB_A_PER_UNIT, float(sigma_d), float(sigma_a), w_d.data_ptr(), b_d.data_ptr(), w_a.data_ptr(),
                b_a.data_ptr(), div_term.data_ptr(), wigner_d1.data_ptr() if eq is not None else None, A, emb.data_ptr(),
                eq.data_ptr() if eq is not None else None, ws.data_ptr(), ws.numel(), _stream()), 'se3_geo_embedding_fwd')
    return emb if eq is None else (emb, eq)


def geometric_embedding_bwd(grad_emb, points, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, knn, tables=None):
    """Gradients of geometric_embedding with respect to (w_d, b_d, w_a, b_a): the HIP kernel writes the GEMM operands (sinusoid embeddings
    of the four indices, gradient masked by the arg-max angle), the products are two library GEMMs (csrc/geo_embedding.hip)."""
    points = _req(points.contiguous(), torch.float32, 'points', 2)
    g = _req(grad_emb.contiguous(), torch.float32, 'grad_emb', 3)
    N, C = points.shape[0], w_d.shape[0]
    knn = _req(knn, torch.int64, 'knn', 2)
    tab_a = (tables if tables is not None else embedding_tables(div_term, w_d, b_d, w_a, b_a, sigma_a))[1]
    S = torch.empty((4, N * N, C), dtype=torch.float32, device=points.device)
    dEk = torch.empty((3 * N * N, C), dtype=torch.float32, device=points.device)
    check(lib().se3_geo_embedding_bwd_operands(points.data_ptr(), knn.data_ptr(), N, C, tab_a.data_ptr(), tab_a.shape[0], _EMB_A_PER_UNIT,
                                               float(sigma_d), float(sigma_a), w_a.data_ptr(), b_a.data_ptr(), div_term.data_ptr(),
                                               g.data_ptr(), S.data_ptr(), dEk.data_ptr(), _stream()), 'se3_geo_embedding_bwd_operands')
    g2 = g.reshape(N * N, C)
    db = g2.sum(0)
    return mm_tn_splitk(g2, S[0]), db, mm_tn_splitk(dEk, S[1:].reshape(3 * N * N, C)), db.clone()


def mm_tn_splitk(a, b, splits=64):
    """a (K, M)^T @ b (K, N) for K >> M, N (the weight gradients of the geometric embedding: K = N^2 .. 3 N^2 = 1.5e5 .. 4.4e5 rows onto a
    256 x 256 result).  As ONE product the library runs 32 workgroups for 0.6 ms per call (2.4 ms of a 42 ms training step,
    profiles/r04_kernel_stats_train.csv); cut into `splits` row blocks it is a batched product that fills the chip, the partial results summed
    in a fixed order."""
    K, M = a.shape
    k = K // splits
    if k < 512:
        return mm(a.t(), b)
    head = splits * k
    out = torch.bmm(a[:head].view(splits, k, M).transpose(1, 2), b[:head].view(splits, k, b.shape[1])).sum(0)
    if head < K:
        out += mm(a[head:].t(), b[head:])
    return out


def point_to_node_partition(points, nodes, point_limit):
    """HIP (csrc/partition.hip): (point_to_node (N,) int64, node_masks (M,) bool, node_knn_indices (M, K) int64 padded with N,
    node_knn_masks (M, K) bool) -- nearest node per point, the K nearest own points per node."""
    points = _req(points.contiguous(), torch.float32, 'points', 2)
    nodes = _req(nodes.contiguous(), torch.float32, 'nodes', 2)
    N, M, K = points.shape[0], nodes.shape[0], int(point_limit)
    dev = points.device
    p2n = torch.empty((N,), dtype=torch.int64, device=dev)
    masks = torch.empty((M,), dtype=torch.bool, device=dev)
    knn = torch.empty((M, K), dtype=torch.int64, device=dev)
    knn_masks = torch.empty((M, K), dtype=torch.bool, device=dev)
    check(lib().se3_point_to_node_partition(points.data_ptr(), nodes.data_ptr(), N, M, K, p2n.data_ptr(), masks.data_ptr(),
                                            knn.data_ptr(), knn_masks.data_ptr(), _stream()), 'se3_point_to_node_partition')
    return p2n, masks, knn, knn_masks


def point_to_node_partition_stack(points, nodes, point_lengths, node_lengths, point_limit):
    """HIP (csrc/partition.hip), all clouds in one launch per kernel: stacked fine points / superpoints with per-cloud lengths
    (host lists).  GLOBAL indices: (point_to_node (P,), node_masks (M,) bool, node_knn_indices (M, K) padded with P,
    node_knn_masks (M, K) bool)."""
    points = _req(points.contiguous(), torch.float32, 'points', 2)
    nodes = _req(nodes.contiguous(), torch.float32, 'nodes', 2)
    if len(point_lengths) != len(node_lengths) or sum(point_lengths) != points.shape[0] or sum(node_lengths) != nodes.shape[0]:
        raise RuntimeError('point_to_node_partition_stack: lengths do not add up to the stacked arrays')
    P, M, K = points.shape[0], nodes.shape[0], int(point_limit)
    dev = points.device
    p2n = torch.empty((P,), dtype=torch.int64, device=dev)
    masks = torch.empty((M,), dtype=torch.bool, device=dev)
    knn = torch.empty((M, K), dtype=torch.int64, device=dev)
    knn_masks = torch.empty((M, K), dtype=torch.bool, device=dev)
    check(lib().se3_point_to_node_partition_stack(points.data_ptr(), nodes.data_ptr(), _i64_array(point_lengths),
                                                  _i64_array(node_lengths), len(point_lengths), K, p2n.data_ptr(), masks.data_ptr(),
                                                  knn.data_ptr(), knn_masks.data_ptr(), _stream()), 'se3_point_to_node_partition_stack')
    return p2n, masks, knn, knn_masks


def superpoint_scores_stack(feats, node_masks, ref_rows, src_rows, ref_lengths, src_lengths, ref_mask_offsets, src_mask_offsets,
                            dual_normalization):
    """HIP (csrc/matching.hip), all pairs in one launch per kernel: feats (rows, C) unit superpoint features, node_masks bool
    (nodes,); per pair (host lists) the first ref / src row, the counts and the first ref / src entry of node_masks.  Returns
    (B, max_p N_p * M_p) scores: pair p's (N_p, M_p) matrix at the start of row p, -1 for absent nodes and beyond."""
    feats = _req(feats.contiguous(), torch.float32, 'feats', 2)
    node_masks = _req(node_masks.contiguous(), torch.bool, 'node_masks', 1)
    B, C = len(ref_rows), feats.shape[1]
    for p in range(B):
        if ref_rows[p] + ref_lengths[p] > feats.shape[0] or src_rows[p] + src_lengths[p] > feats.shape[0] or \
                ref_mask_offsets[p] + ref_lengths[p] > node_masks.shape[0] or src_mask_offsets[p] + src_lengths[p] > node_masks.shape[0]:
            raise RuntimeError('superpoint_scores_stack: pair %d exceeds the feature rows / node masks' % p)
    stride = max(n * m for n, m in zip(ref_lengths, src_lengths))
    scores = torch.empty((B, stride), dtype=torch.float32, device=feats.device)
    ws = torch.empty((sum(ref_lengths) + sum(src_lengths),), dtype=torch.float32, device=feats.device)
    check(lib().se3_superpoint_scores_stack(feats.data_ptr(), node_masks.data_ptr(), _i64_array(ref_rows), _i64_array(src_rows),
                                            _i64_array(ref_lengths), _i64_array(src_lengths), _i64_array(ref_mask_offsets),
                                            _i64_array(src_mask_offsets), B, C, 1 if dual_normalization else 0, stride,
                                            scores.data_ptr(), ws.data_ptr(), _stream()), 'se3_superpoint_scores_stack')
    return scores


def patch_scores(feats, ref_idx, src_idx, scale):
    """HIP (csrc/matching.hip): fine-matching scores (B, K, K) of all patch pairs, scale * <feats[ref_idx[b, n]], feats[src_idx[b, m]]>, the
    two gathers fused (an index outside [0, rows) selects the reference's zero padding row)."""
    feats = _req(feats, torch.float32, 'feats', 2)
    ref_idx, src_idx = _req(ref_idx.contiguous(), torch.int64, 'ref_idx', 2), _req(src_idx.contiguous(), torch.int64, 'src_idx', 2)
    B, K = ref_idx.shape
    if tuple(src_idx.shape) != (B, K):
        raise RuntimeError('patch_scores: index shapes %s and %s' % (tuple(ref_idx.shape), tuple(src_idx.shape)))
    out = torch.empty((B, K, K), dtype=torch.float32, device=feats.device)
    check(lib().se3_patch_scores(feats.data_ptr(), ref_idx.data_ptr(), src_idx.data_ptr(), B, K, feats.shape[0], feats.shape[1], float(scale),
                                 out.data_ptr(), _stream()), 'se3_patch_scores')
    return out


def patch_scores_ok(feats, K):
    return feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.shape[1] % 64 == 0 and K in (64, 128) and \
        not (torch.is_grad_enabled() and feats.requires_grad)


def anchor_mix_stack(x, mixes, starts, lengths):
    """HIP (csrc/rowops.hip): out[a, r] = sum_e mixes[p][a, e] x[e, r] for the packed rows r of pair p (rows of no pair: zero); x (6, R, C)."""
    x = _req(x, torch.float32, 'x', 3)
    mixes = _req(mixes.contiguous(), torch.float32, 'mixes', 3)
    if x.shape[0] != 6 or tuple(mixes.shape) != (len(starts), 6, 6):
        raise RuntimeError('anchor_mix_stack: x %s, mixes %s for %d pairs' % (tuple(x.shape), tuple(mixes.shape), len(starts)))
    out = torch.empty_like(x)
    check(lib().se3_anchor_mix_stack(x.data_ptr(), x.shape[1], x.shape[2], mixes.data_ptr(), _i64_array(starts), _i64_array(lengths), len(starts),
                                     out.data_ptr(), _stream()), 'se3_anchor_mix_stack')
    return out


def superpoint_scores(ref_feats, src_feats, dual_normalization):
    """HIP (csrc/matching.hip): exp(-||f_r - f_s||^2) on unit features with dual normalisation."""
    ref_feats = _req(ref_feats.contiguous(), torch.float32, 'ref_feats', 2)
    src_feats = _req(src_feats.contiguous(), torch.float32, 'src_feats', 2)
    N, C = ref_feats.shape
    M = src_feats.shape[0]
    scores = torch.empty((N, M), dtype=torch.float32, device=ref_feats.device)
    ws = torch.empty((N + M,), dtype=torch.float32, device=ref_feats.device)
    check(lib().se3_superpoint_scores(ref_feats.data_ptr(), src_feats.data_ptr(), N, M, C, 1 if dual_normalization else 0,
                                      scores.data_ptr(), ws.data_ptr(), _stream()), 'se3_superpoint_scores')
    return scores


def mutual_topk_mask(scores, row_masks, col_masks, k, threshold):
    """HIP (csrc/registration.hip): bool (B, R, C) mask of the entries that are among the k largest of their row and of their
    column, above `threshold`, with valid row and column points."""
    scores = _req(scores.contiguous(), torch.float32, 'scores', 3)
    B, R, C = scores.shape
    rm = row_masks.contiguous()
    cm = col_masks.contiguous()
    if rm.dtype != torch.bool or cm.dtype != torch.bool or tuple(rm.shape) != (B, R) or tuple(cm.shape) != (B, C):
        raise RuntimeError('mutual_topk_mask: masks must be bool (B, R) / (B, C)')
    out = torch.empty((B, R, C), dtype=torch.bool, device=scores.device)
    check(lib().se3_mutual_topk_mask(scores.data_ptr(), rm.data_ptr(), cm.data_ptr(), B, R, C, int(k), float(threshold),
                                     out.data_ptr(), _stream()), 'se3_mutual_topk_mask')
    return out


def weighted_procrustes(src, ref, scores, offsets, gate_transform=None, gate_radius=0.0, eps=1e-5):
    """HIP (csrc/registration.hip): one weighted Kabsch solve per segment of the stacked correspondences -> (S, 4, 4).
    gate_transform: (4, 4) shared by all segments, or (S, 4, 4) one per segment (several pairs at once)."""
    src = _req(src.contiguous(), torch.float32, 'src', 2)
    ref = _req(ref.contiguous(), torch.float32, 'ref', 2)
    scores = _req(scores.contiguous(), torch.float32, 'scores', 1)
    offsets = _req(offsets.contiguous(), torch.int64, 'offsets', 1)
    S = offsets.shape[0] - 1
    T = torch.empty((S, 4, 4), dtype=torch.float32, device=src.device)
    gt = _req(gate_transform.contiguous(), torch.float32, 'gate_transform') if gate_transform is not None else None
    per_segment = gt is not None and gt.dim() == 3
    if per_segment and gt.shape[0] != S:
        raise RuntimeError('weighted_procrustes: %d gate transforms for %d segments' % (gt.shape[0], S))
    check(lib().se3_weighted_procrustes_segments(src.data_ptr(), ref.data_ptr(), scores.data_ptr(), offsets.data_ptr(), S,
                                                 gt.data_ptr() if gt is not None else None, 1 if per_segment else 0,
                                                 float(gate_radius), float(eps), T.data_ptr(), _stream()),
          'se3_weighted_procrustes_segments')
    return T


def count_inliers(src, ref, transforms, radius, range_begin=None, range_end=None):
    """HIP: votes[t] = number of correspondences i (in [range_begin[t], range_end[t]) if given, else all) with
    |ref_i - T_t src_i| < radius."""
    src = _req(src.contiguous(), torch.float32, 'src', 2)
    ref = _req(ref.contiguous(), torch.float32, 'ref', 2)
    transforms = _req(transforms.contiguous(), torch.float32, 'transforms', 3)
    votes = torch.empty((transforms.shape[0],), dtype=torch.int32, device=src.device)
    if range_begin is not None:
        range_begin = _req(range_begin.contiguous(), torch.int64, 'range_begin', 1)
        range_end = _req(range_end.contiguous(), torch.int64, 'range_end', 1)
        if range_begin.shape[0] != transforms.shape[0] or range_end.shape[0] != transforms.shape[0]:
            raise RuntimeError('count_inliers: one range per transform')
    check(lib().se3_count_inliers_ranges(src.data_ptr(), ref.data_ptr(), src.shape[0], transforms.data_ptr(), transforms.shape[0],
                                         range_begin.data_ptr() if range_begin is not None else None,
                                         range_end.data_ptr() if range_end is not None else None, float(radius),
                                         votes.data_ptr(), _stream()), 'se3_count_inliers_ranges')
    return votes
