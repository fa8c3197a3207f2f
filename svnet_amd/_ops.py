"""torch.autograd.Function wrappers over the C ABI (include/svnet_hip.h).

PyTorch is plumbing here: it owns device memory (every output / workspace is a torch tensor), the
current HIP stream and the autograd graph.  All arithmetic happens in libsvnet_hip.so.  Every entry
point refuses CPU tensors — there is deliberately no fallback path.
"""
import ctypes

import torch

from . import _lib, config
from ._lib import BinHeadDesc, GemmDesc, call

BN_EPS = 1e-5
RED_SLICES = 16          # SVNET_RED_SLICES (include/svnet_hip.h): slices of the fused backward preludes' batch sums
BN_MOMENTUM = 0.1

# Test instrumentation (None in production): a dict {"knn": [], "signs": [], "pools": []} that records the DISCRETE decisions of a
# forward - every neighbour list, the sign / non-zero / STE bit planes of every binarized activation and the arg-max of every
# max-pool, in call order - so that the parity
# tests can replay them into the oracle and certify each disagreement as a knife edge (tests/decisions.py).  Recording only reads
# what the kernels write anyway; no arithmetic changes.
TAP = None


# ----------------------------------------------------------------------------- helpers

def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("svnet_amd: expected a HIP (cuda) tensor, got %s — the product path has no CPU fallback" % t.device)


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError("svnet_amd: expected float32, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def gemm(M, N, K, B, b_rs, b_cs, C, ldc, A=None, a_rs=0, a_cs=0, a_scale=None, a_planes=None, b_exact=False, c_cs=1,
         alpha=1.0, col_scale=None, bias=None, mask=None, col_sum=None, split_k=0, accumulate=False, tern_tile_mask=0):
    """C(i,j) = epilogue(sum_k A(i,k) B(k,j)); see svnet_gemm_desc (bit-planes are row-sliced)."""
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.a_rs, d.a_cs = _p(A), a_rs, a_cs
    d.a_scale = _p(a_scale)
    if a_planes is not None:
        d.a_sign, d.a_nz = _p(a_planes[0]), _p(a_planes[1])
    d.B, d.b_rs, d.b_cs = _p(B), b_rs, b_cs
    d.b_exact = int(b_exact)
    d.C, d.ldc, d.c_cs = _p(C), ldc, c_cs
    d.alpha = alpha
    d.col_scale, d.bias = _p(col_scale), _p(bias)
    d.mask = _p(mask)
    d.col_sum = _p(col_sum)
    d.split_k, d.accumulate = split_k, int(accumulate)
    d.tern_tile_mask = tern_tile_mask
    ws = None
    # big product against sign weights: let the library pack B once (K >= 64 with more than 128 columns: the LDS-tiled rows kernel,
    # mfma_rows2_kernel, whose A tile arrives in coalesced row pieces whatever the row alignment - conv5's linear2, K = 83)
    if b_exact and A is not None and M >= 8192 and N >= 64 and (K >= 128 or (K >= config.ROWS2_MIN_K and N > 128)):
        nbytes = _lib.lib().svnet_gemm_workspace_bytes(N, K)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=C.device)
        d.workspace, d.workspace_bytes = _p(ws), nbytes
    elif (not b_exact and A is not None and a_cs == 1 and c_cs == 1 and M >= 1024 and K >= 8 and N >= 8 and a_scale is None and col_scale is None
          and mask is None and col_sum is None):
        # many rows against general fp32 weights (the fp layers): three exact bf16 pieces of B, packed by the library (matrix cores)
        nbytes = 3 * _lib.lib().svnet_gemm_workspace_bytes(N, K)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=C.device)
        d.workspace, d.workspace_bytes = _p(ws), nbytes
    call("svnet_gemm_f32", ctypes.byref(d), _stream())


def _words(K):
    return (K + 63) // 64


# ----------------------------------------------------------------------------- k-NN and edge features

def _check_finite(name, *tensors):
    """config.DEBUG_FINITE (SVNET_DEBUG_FINITE=1): the k-NN kernels clamp their output ids so that a NaN feature can never make the
    next gather fault the GPU - which also turns such a NaN into silently wrong neighbours.  In debug mode the inputs
    are checked instead (one device reduction + a host sync per call: not for timed runs, not inside a graph capture)."""
    if config.DEBUG_FINITE:
        for t in tensors:
            if not bool(torch.isfinite(t).all()):
                raise FloatingPointError("svnet_amd: %s received non-finite features (%d NaN, %d Inf of %d)"
                                         % (name, int(torch.isnan(t).sum()), int(torch.isinf(t).sum()), t.numel()))


def knn(x, k):
    """x: [B,C,N] (any strides, as sv_util.knn receives it) -> idx [B,N,k] int64, cloud-local, nearest first."""
    _hip(x)
    if x.dtype != torch.float32 or x.dim() != 3:
        raise TypeError("knn expects a float32 [B,C,N] tensor")
    _check_finite("knn", x)
    B, C, N = x.shape
    xx_mode = 1 if (x.stride(1) == 1 and C > 1) else 0
    if xx_mode == 0 and not x.is_contiguous():
        x = x.contiguous()
    L = _lib.lib()
    nbytes = L.svnet_knn_workspace_bytes(B, N, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    idx = torch.empty((B, N, k), dtype=torch.int64, device=x.device)
    call("svnet_knn_f32", _p(x), B, N, C, x.stride(0), x.stride(2), x.stride(1), xx_mode, int(k), _p(idx), _p(ws), nbytes,
                          _stream())
    if TAP is not None:
        TAP["knn"].append(idx)
    return idx


def _block_tail(entry, P, N, E, Os, Ov, stat1, stat_v, sc1, g1, b1, rm1, rv1, g2, b2, rm2, rv2, training, coef, nbt1, nbt2, job, hi, lo, mv, mvn,
                s_out, v_out, slot, kws):
    """svnet_{edgeblock,xyzblock}_tail_f32: coefficients + gate MLP + apply (+ the k-NN table kws) of a fused level as ONE launch.
    False = not taken (switch off / shape not supported): the caller issues the separate launches."""
    if not config.FUSE_BLOCK_TAIL or not _lib.lib().svnet_block_tail_supported(P, N, Os, Ov, 1 if kws is not None else 0):
        return False
    d = _lib.BlockTailDesc()
    d.stat1, d.stat_v, d.E, d.Os, d.Ov, d.scale1 = _p(stat1), _p(stat_v), E, Os, Ov, _p(sc1)
    d.gamma1, d.beta1, d.running_mean1, d.running_var1 = _p(g1), _p(b1), _p(rm1), _p(rv1)
    d.gamma2, d.beta2, d.running_mean2, d.running_var2 = _p(g2), _p(b2), _p(rm2), _p(rv2)
    d.training, d.eps, d.momentum = int(training), BN_EPS, BN_MOMENTUM
    d.coef, d.num_batches_tracked1, d.num_batches_tracked2 = _p(coef), _p(nbt1), _p(nbt2)
    d.gate = job
    d.hi, d.lo, d.mv, d.mvn, d.P, d.N, d.slope = _p(hi), _p(lo), _p(mv), _p(mvn), P, N, 0.2
    d.s_out, d.v_out = _p(s_out), _p(v_out)
    if slot is not None:
        d.s_cat, d.s_ld, d.v_cat, d.v_ld = slot
    if kws is not None:
        d.knn_workspace, d.knn_workspace_bytes = _p(kws), kws.numel()
    call(entry, ctypes.byref(d), _stream())
    return True


class knn_table_ahead:
    """`with knn_table_ahead():` around a fused level whose pooled (s, v) is read next by get_graph_feature_sv (sv_dgcnn_cls.py:55-65):
    the level's apply pass then also writes the candidate table of that k-NN (svnet_*_apply_knn_f32: the k-NN is on the forward's
    critical path and its first kernel only re-read, squared and transposed those rows), and knn_sv on exactly those tensors skips
    its preparation.  Anything else - other tensors, a shape the fused producer does not take - runs the plain path."""
    active = 0
    table = None                            # (s.data_ptr(), v.data_ptr(), workspace, B, N, C) of the last prepared level

    def __enter__(self):
        knn_table_ahead.active += 1
        return self

    def __exit__(self, *exc):
        knn_table_ahead.active -= 1
        return False

    @staticmethod
    def workspace(B, N, Os, Ov, dev):
        """The workspace the apply pass should fill, or None (context off, switch off, unsupported shape)."""
        if not (knn_table_ahead.active and config.KNN_TABLE_AHEAD):
            return None
        L = _lib.lib()
        if not L.svnet_knn_table_fusable(B, N, Os + 3 * Ov):
            return None
        return torch.empty(L.svnet_knn_workspace_bytes(B, N, Os + 3 * Ov), dtype=torch.uint8, device=dev)

    @staticmethod
    def take(s, v):
        t, knn_table_ahead.table = knn_table_ahead.table, None
        if t is None or t[0] != s.data_ptr() or t[1] != v.data_ptr() or t[3:] != (s.shape[0], s.shape[1], s.shape[2] + 3 * v.shape[-1]):
            return None
        return t[2]


def knn_sv(s, v, k):
    """Feature-space graph of get_graph_feature_sv (sv_util.py:100-101): knn(cat[s, v.view(B,N,3Cv)].transpose(-1,-2), k) without
    the concatenated copy.  s [B,N,Cs], v [B,N,3,Cv] -> idx [B,N,k] int64."""
    _hip(s, v)
    s, v = _f32c(s.detach()), _f32c(v.detach())
    _check_finite("knn (feature space)", s, v)
    B, N, Cs = s.shape
    Cv3 = 3 * v.shape[-1]
    idx = torch.empty((B, N, k), dtype=torch.int64, device=s.device)
    ws = knn_table_ahead.take(s, v)
    if ws is not None:                       # the producer of (s, v) has prepared the table
        call("svnet_knn_from_table_f32", _p(ws), ws.numel(), B, N, Cs + Cv3, int(k), _p(idx), _stream())
    else:
        nbytes = _lib.lib().svnet_knn_workspace_bytes(B, N, Cs + Cv3)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
        call("svnet_knn_sv_f32", _p(s), Cs, _p(v), Cv3, B, N, int(k), _p(idx), _p(ws), nbytes, _stream())
    if TAP is not None:
        TAP["knn"].append(idx)
    return idx


def edge_xyz(x, idx, mode):
    """x: [B,3m,N]; idx [B,N,k] cloud-local; mode 0 plain / 1 first / 2 cross -> [B,N,k,3,W]."""
    _hip(x, idx)
    x = _f32c(x.detach())
    B, C, N = x.shape
    m = C // 3
    k = idx.shape[-1]
    W = (3 if mode == 2 else 2) * m
    out = torch.empty((B, N, k, 3, W), dtype=torch.float32, device=x.device)
    call("svnet_edge_xyz_f32", _p(x), _p(idx.contiguous()), B, N, k, m, mode, _p(out), _stream())
    return out


class EdgeDiffcat(torch.autograd.Function):
    """table [B,N,G,F] -> [B,N,k,G,2F] = [t_j - t_i, t_i]   (sv_util.py:106-114)."""

    @staticmethod
    def forward(ctx, table, idx, idx_is_global, k):
        _hip(table, idx)
        table = _f32c(table)
        B, N, G, F = table.shape
        idx = idx.contiguous()
        out = torch.empty((B, N, k, G, 2 * F), dtype=torch.float32, device=table.device)
        call("svnet_edge_diffcat_fwd_f32", _p(table), _p(idx), int(idx_is_global), B, N, k, G, F, _p(out), _stream())
        ctx.save_for_backward(idx)
        ctx.meta = (B, N, G, F, k, int(idx_is_global))
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, G, F, k, glob = ctx.meta
        g = _f32c(g)
        d_table = _zeros((B, N, G, F), torch.float32, g.device)
        call("svnet_edge_diffcat_bwd_f32", _p(g), _p(idx), glob, B, N, k, G, F, _p(d_table), _stream())
        return d_table, None, None, None


# ----------------------------------------------------------------------------- dense layers

class FpLinear(torch.autograd.Function):
    """y = x W^T (+ b)   (sv_layers.py:30-31 fast path, nn.Linear)."""

    @staticmethod
    def forward(ctx, x, W, bias):
        _hip(x, W, bias)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        W = _f32c(W)
        M, K = x2.shape
        O = W.shape[0]
        y = torch.empty((M, O), dtype=torch.float32, device=x.device)
        gemm(M, O, K, A=x2, a_rs=K, a_cs=1, B=W, b_rs=1, b_cs=K, C=y, ldc=O, bias=bias)
        ctx.save_for_backward(x2, W)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return y.view(x.shape[:-1] + (O,))

    @staticmethod
    def backward(ctx, g):
        x2, W = ctx.saved_tensors
        M, K = x2.shape
        O = W.shape[0]
        g2 = _f32c(g).reshape(M, O)
        dx = dW = db = None
        if M <= 64 and O * K <= (1 << 22) and O <= 4096 and ctx.needs_input_grad[1]:
            # a few rows (the classifier's output layer): dx, dW and db in ONE launch instead of three launch-bound products
            dx = torch.empty((M, K), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
            dW = torch.empty((O, K), dtype=torch.float32, device=g.device)
            db = torch.empty((O,), dtype=torch.float32, device=g.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            call("svnet_fplinear_small_bwd_f32", _p(g2), _p(x2), _p(W), M, K, O, _p(dx), _p(dW), _p(db), _stream())
            return (dx.view(ctx.xshape) if dx is not None else None), dW, db
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=g.device)
            gemm(M, K, O, A=g2, a_rs=O, a_cs=1, B=W, b_rs=K, b_cs=1, C=dx, ldc=K)
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dW = _zeros((O, K), torch.float32, g.device)
            gemm(O, K, M, A=g2, a_rs=1, a_cs=O, B=x2, b_rs=K, b_cs=1, C=dW, ldc=K, accumulate=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = pool_raw(g2, 1, M, O, 1)[0].view(O) * float(M)
        return dx, dW, db


def _binweight_grad(GX, W, sc, O, K, training, gx_sliced=False, sum_buf=None, sum_len=0):
    """STE chain rule to (W, scale) of a bw layer; eval mode binarizes with a bare sign(): no gradient reaches W
    (sv_layers.py:44-45).  gx_sliced: GX is a sliced accumulator (its slices are added up by the kernel); sum_buf / sum_len: another
    sliced accumulator of the same backward whose totals the launch leaves in its first sum_len elements (dL/dbeta)."""
    dev = GX.device
    dsc = torch.empty((O,), dtype=torch.float32, device=dev)
    dW = torch.empty((O, K), dtype=torch.float32, device=dev) if training else torch.zeros((O, K), dtype=torch.float32, device=dev)
    call("svnet_binweight_grad_f32", _p(GX), _p(W), _p(sc), O, K, _p(dW) if training else None, _p(dsc), 0, int(bool(gx_sliced)),
         _p(sum_buf), int(sum_len), _stream())
    return dW, dsc


class _ZeroArena:
    """Zero-filled scratch of ONE training / inference step from a single fill: every accumulator of the step (BatchNorm sums,
    atomically accumulated gradients, ...) is a slice of one persistent buffer that `begin()` clears with one launch.  The slices
    are handed out in call order (a bump allocator), so a step that repeats asks for the same slices; what lies beyond the extent
    cleared at `begin()` (first step, a larger shape) falls back to torch.zeros and enlarges the extent for the next step.
    Slices stay valid until the next begin() - long enough for gradients, which the bucket copies before that.
    Every TrainStep / ForwardStep owns ITS arena (svnet_amd.train), and a step captured into a HIP graph pins it: the graph has
    the buffer's addresses baked in (the fill and every accumulator slice), so the buffer is never replaced afterwards and lives
    as long as the step object - another step, model or batch size cannot free or outgrow it under the graph's feet.
    Outside begin()/end() (tests calling single ops, ...) nothing is pooled."""

    def __init__(self):
        self.buf, self.zeroed, self.off, self.active, self.pinned = None, 0, 0, False, False

    def begin(self, dev, pin=False):
        need = (self.off + 4095) // 4096 * 4096
        if need and not self.pinned and (self.buf is None or self.buf.numel() < need or self.buf.device != torch.device(dev)):
            self.buf = torch.empty((need + need // 4 + (1 << 20),), dtype=torch.uint8, device=dev)
        self.zeroed = min(need, self.buf.numel()) if self.buf is not None else 0      # (pinned and outgrown: the rest comes from torch.zeros)
        if self.zeroed:
            self.buf[:self.zeroed].zero_()
        self.off, self.active = 0, True
        self.pinned = self.pinned or pin

    def end(self):
        self.active = False

    def take(self, nbytes, dev):
        if not self.active:
            return None
        o = self.off
        self.off += (nbytes + 255) // 256 * 256
        if self.buf is not None and self.off <= self.zeroed and self.buf.device == torch.device(dev):
            return self.buf[o:o + nbytes]
        return None


_DEFAULT_ARENA = _ZeroArena()
ARENA = _DEFAULT_ARENA          # the arena of the step that is running (begin_step / end_step)


def _zeros(shape, dtype, dev):
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    buf = ARENA.take(nbytes, dev) if nbytes else None
    if buf is None:
        return torch.zeros(tuple(shape), dtype=dtype, device=dev)
    return buf.view(dtype).view(tuple(shape))


def _zeros_pool(dev, *specs):
    """One zero-filled allocation (one fill launch, none inside a step: _ZeroArena) carved into tensors: specs are (shape, dtype) pairs."""
    offs, total = [], 0
    for shape, dtype in specs:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        offs.append((total, nbytes))
        total += (nbytes + 255) // 256 * 256
    buf = ARENA.take(max(total, 1), dev)
    if buf is None:
        buf = torch.zeros((max(total, 1),), dtype=torch.uint8, device=dev)
    return [buf[o:o + nb].view(dtype).view(shape) for (o, nb), (shape, dtype) in zip(offs, specs)]


def _sliced_len(L):
    """SVNET_SLICED_LEN(L) (include/svnet_hip.h): [L result | RED_SLICES x L slices | arrival counter] - the zero-filled accumulator the
    grid-wide reductions add to; consumers read the first L elements."""
    return (RED_SLICES + 1) * L + 2


class _PlaneCache:
    """Packed forms of the binarized weights (sign / non-zero bit planes, +-1 values, scale*sign, the fused edge kernels' permuted
    planes and MFMA-fragment sign weights): sv_layers.py:44-48 re-derives sign(W) inside every forward; here a packed form is
    rebuilt only when its parameters CHANGED since it was packed - autograd's version counter (torch.optim steps, load_state_dict,
    in-place ops under no_grad) or the data address (re-homing into a flat buffer); writers that bypass both (the flat optimizer
    kernels of svnet_amd.train, anything going through .data or a raw pointer) call invalidate().  An eager step rebuilds the
    stale forms for all layers together on the side stream at its start (begin()), off the forward's critical path; a step
    replayed as a captured graph keeps the packing OUTSIDE the graph (begin(external=True) while capturing, refresh() before
    every replay).  Entries are created lazily the first time a layer asks (built inline that once).  Keys are the Parameter
    objects; the rebuild closures read their CURRENT data.  Outside begin()/end() every layer packs on the fly."""

    def __init__(self):
        self.entries, self.active, self.event, self.waited = {}, False, None, None      # waited: ids of the streams that have joined
        self.sig, self.dirty = {}, False        # (version counter, address) of every entry's parameters at its last (re)build

    @staticmethod
    def _signature(params):
        return tuple((p._version, p.data_ptr()) for p in params)

    def invalidate(self):
        """The weights were changed behind autograd's back (a raw kernel on a flat parameter buffer: svnet_amd.train's optimizers call
        this; so must anything else that writes through .data or a raw pointer): rebuild everything at the next step."""
        self.dirty = True

    def _stale(self):
        """Entries whose parameters changed since they were packed (autograd version counter or address), or all after invalidate()."""
        out = []
        for key in list(self.entries):
            refs, _, rebuild = self.entries[key]
            ps = [r() for r in refs]
            if any(p is None for p in ps):
                del self.entries[key]                 # the parameters are gone (another model was built)
                self.sig.pop(key, None)
            elif self.dirty or self.sig.get(key) != self._signature(ps):
                out.append((key, ps, rebuild))
        return out

    def refresh(self, dev):
        """Re-pack what is stale, on the CURRENT stream.  For a step that is replayed as a captured graph (the packing launches are
        not part of the graph: they depend on whether an optimizer ran in between) - svnet_amd.train calls it before every replay."""
        for key, ps, rebuild in self._stale():
            rebuild()
            self.sig[key] = self._signature(ps)
        self.dirty = False

    def rebuild_all(self, run=None):
        """Every live entry re-packed now, whatever its state (run(rebuild) issues one; default: on the current stream) - what a
        captured optimizer step records behind its update kernel (svnet_amd.train).  Returns the entries' keys."""
        keys = []
        for key in list(self.entries):
            refs, _, rebuild = self.entries[key]
            if any(r() is None for r in refs):
                continue
            (run or (lambda f: f()))(rebuild)
            keys.append(key)
        return keys

    def mark_fresh(self, keys):
        """The given entries have just been re-packed from the parameters' current values by someone else (a replayed graph)."""
        for key in keys:
            e = self.entries.get(key)
            if e is not None:
                ps = [r() for r in e[0]]
                if all(p is not None for p in ps):
                    self.sig[key] = self._signature(ps)

    def _join(self):
        if self.waited is not None:
            st = torch.cuda.current_stream()
            if st.cuda_stream not in self.waited:
                st.wait_event(self.event)
                self.waited.add(st.cuda_stream)

    def begin(self, dev, external=False):
        """external: the step is being captured into a graph - its packed forms are kept fresh by refresh() before every replay."""
        self.active, self.waited = True, None
        if external:
            return
        stale = self._stale()
        if stale:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for key, ps, rebuild in stale:
                    rebuild()
                    self.sig[key] = self._signature(ps)
                self.event = side.record_event()
            self.waited = {side.cuda_stream}
        self.dirty = False

    def end(self):
        if self.active:                          # (also when nothing looked anything up: graph capture needs the side stream joined)
            self._join()
        self.active = False

    def get(self, kind, params, build):
        """build() -> (outputs, rebuild): packs now; rebuild() re-packs into the same output tensors."""
        if not self.active:
            return build()[0]
        key = (kind,) + tuple(id(p) for p in params)
        e = self.entries.get(key)
        if e is not None and all(r() is p for r, p in zip(e[0], params)):
            self._join()
            return e[1]
        import weakref
        out, rebuild = build()
        self.entries[key] = (tuple(weakref.ref(p) for p in params), out, rebuild)
        self.sig[key] = self._signature(params)
        return out


PLANES = _PlaneCache()


def begin_step(dev, planes_external=False, arena=None):
    """Start of a train / inference step (svnet_amd.train): one fill for the step's zero-initialised scratch (`arena`: the step
    object's own _ZeroArena; `planes_external` = the step is being captured into a HIP graph, which also pins the arena), and the
    packed weight forms of the layers whose weights changed since they were last packed rebuilt on the side stream."""
    global ARENA
    ARENA = arena if arena is not None else _DEFAULT_ARENA
    ARENA.begin(dev, pin=planes_external)
    PLANES.begin(dev, planes_external)


def end_step():
    global ARENA, _FUSED_COLSUMS, _FUSED_VSTATS
    knn_table_ahead.table = None                          # (a prepared k-NN table nobody asked for is not held across steps)
    _FUSED_VSTATS = None
    _FUSED_COLSUMS = None                                 # (the producer's [M, O] output and its arena slice are not held across steps)
    ARENA.end()
    ARENA = _DEFAULT_ARENA
    PLANES.end()
    DEFERRED.active = False
    DEFERRED.keep.clear()
    DEFERRED.seen.clear()


class _Deferred:
    """Weight-gradient work that nothing in the backward pass waits for (svnet_amd.train.TrainStep only: it gathers the parameter
    gradients ONCE, after the backward).  While `active`, a fused edge layer's backward leaves its weight-gradient chain (linear1's
    product, linear2's, the parameter epilogue) on the side stream WITHOUT joining it into the main stream, which goes on with the next
    layer's backward; `join()` (called before the gradients are packed) is the one join.  `keep` holds every tensor those kernels
    touch until then: the caching allocator would otherwise hand their blocks - freed when the backward function returns - to the
    main stream's next allocation while the side stream still reads them.  Off (plain autograd use of the layers): every backward
    joins before it returns, as before."""

    def __init__(self):
        self.active = False
        self.keep = []
        self.seen = set()

    def first_use(self, *params):
        """True while none of `params` (a layer's weights) has been handed an unwritten gradient in this backward.  A module applied
        twice in one forward - or one parameter read by two Functions - gets two gradients, which autograd ADDS on the main stream:
        the second backward must therefore take the joined schedule (its final join also covers the first, deferred chain).  This is
        the run-time half of TrainStep._deferral_is_safe(), which can only see parameters registered under two names."""
        keys = [p.data_ptr() for p in params if p is not None]
        dup = any(k in self.seen for k in keys)
        self.seen.update(keys)
        return not dup

    def join(self, dev):
        if self.keep:
            torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
            self.keep.clear()
        self.seen.clear()


DEFERRED = _Deferred()


class defer_rows_wgrad:
    """`with _ops.defer_rows_wgrad():` around the FORWARD of a big rows layer marks it: inside a TrainStep its weight-gradient chain may go
    to the tail of the side stream (BinLinear.backward).  Opt-in per call site, because it only pays where launch-bound kernels follow the
    layer's backward on the main stream (conv5 of sv_dgcnn_cls: 4.29 -> 4.23 ms); applied to every rows layer it cost sv_pointnet_cls
    --binary 0.38 ms per step (5.66 -> 6.04: the deferred products queue up in front of the next blocks' vector paths)."""
    on = False

    def __enter__(self):
        self.prev, defer_rows_wgrad.on = defer_rows_wgrad.on, True
        return self

    def __exit__(self, *exc):
        defer_rows_wgrad.on = self.prev


def _binweight(W, scale, i8=False):
    """{w_sign, w_nz (row-major 64-bit plane words), w_b (+-1/0 values), w_eff (scale*sign(W)), w_i8 (int8 MFMA operand, only for
    callers that ask: i8=True)} of a bw layer's weight [O,K].  The re-pack closure kept by the cache holds the parameters WEAKLY."""
    import weakref
    O, K, dev = W.shape[0], W[0].numel(), W.device
    w_ref, s_ref = weakref.ref(W), (None if scale is None else weakref.ref(scale))

    def build():
        out = {"w_sign": torch.empty((O, _words(K)), dtype=torch.int64, device=dev), "w_nz": torch.empty((O, _words(K)), dtype=torch.int64, device=dev),
               "w_b": torch.empty((O, K), dtype=torch.float32, device=dev),
               "w_eff": torch.empty((O, K), dtype=torch.float32, device=dev) if s_ref is not None else None, "w_i8": None}

        def rebuild():
            Wp, sp = w_ref(), (None if s_ref is None else s_ref())
            if Wp is None or (s_ref is not None and sp is None):
                return                                    # the layer is gone: _PlaneCache._stale() drops the entry
            Wc = _f32c(Wp.detach()).view(O, K)
            sc = None if sp is None else _f32c(sp.detach()).view(-1)
            call("svnet_binweight_prepare_f32", _p(Wc), _p(sc), O, K, _p(out["w_sign"]), _p(out["w_nz"]), _p(out["w_b"]), _p(out["w_eff"]), _stream())
            if out["w_i8"] is not None:
                call("svnet_binweight_pack_i8", _p(Wc), O, K, _p(out["w_i8"]), _stream())
        rebuild()
        return out, rebuild
    out = PLANES.get("bw", (W,) if scale is None else (W, scale), build)
    if i8 and out["w_i8"] is None:                        # first use on the matrix-core path: packed now, re-packed with the rest from then on
        out["w_i8"] = torch.empty((_lib.lib().svnet_binweight_i8_bytes(O, K),), dtype=torch.int8, device=dev)
        call("svnet_binweight_pack_i8", _p(_f32c(W.detach()).view(O, K)), O, K, _p(out["w_i8"]), _stream())
    return out


class BwLinear(torch.autograd.Function):
    """y = (x sign(W)^T) * scale with fp32 activations (sv_layers.py:44-49 with bw only: linear2, v2s.linear, svfuse)."""

    @staticmethod
    def forward(ctx, x, W, scale, training=True, vstats=False):
        """vstats: x is [..., 3, K] vectors and a VectorBN in training mode consumes the output next (SVBlock's linear2): the product
        also leaves the VectorBN's batch sums (csrc/vlinear.hip) in _FUSED_VSTATS, where VBN.forward finds them - no statistics pass."""
        global _FUSED_VSTATS
        _hip(x, W, scale)
        ctx.training = bool(training)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        W_in, W = W, _f32c(W)
        M, K = x2.shape
        O = W.shape[0]
        sc = _f32c(scale).view(-1)
        w_b = _binweight(W_in, scale)["w_b"]
        y = torch.empty((M, O), dtype=torch.float32, device=x.device)
        _FUSED_VSTATS = None
        if (vstats and training and config.FUSE_VBN_STATS and x.dim() >= 2 and x.shape[-2] == 3 and M % 3 == 0 and M > 0
                and K <= 96 and O <= 256):
            sums = _zeros((_sliced_len(2 * O),), torch.float64, x.device)
            call("svnet_vlinear_stats_f32", _p(x2), M // 3, K, _p(w_b), _p(sc), O, _p(y), _p(sums), _stream())
            _FUSED_VSTATS = (y, y._version, sums)
        else:
            gemm(M, O, K, A=x2, a_rs=K, a_cs=1, B=w_b, b_rs=1, b_cs=K, b_exact=True, C=y, ldc=O, col_scale=sc)
        ctx.save_for_backward(x2, W, sc, w_b)
        ctx.xshape, ctx.sshape = x.shape, scale.shape
        return y.view(x.shape[:-1] + (O,))

    @staticmethod
    def backward(ctx, g):
        x2, W, sc, w_b = ctx.saved_tensors
        M, K = x2.shape
        O = W.shape[0]
        g2 = _f32c(g).reshape(M, O)
        dx = dW = dsc = None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        # many rows: the weight-gradient product runs beside the input-gradient product (its three phases - stage B, MFMA, write C -
        # are in lockstep over the whole chip, one row block per workgroup, so each leaves the other two resources idle)
        beside = _Beside(g.device, config.DW_BESIDE and need_w and ctx.needs_input_grad[0] and M >= config.TWO_STREAM_MIN_ROWS)
        if need_w:
            GX = _zeros((O, K), torch.float32, g.device)
            with beside:
                gemm(O, K, M, A=g2, a_rs=1, a_cs=O, B=x2, b_rs=K, b_cs=1, C=GX, ldc=K, accumulate=True)
                dW, dsc = _binweight_grad(GX, W, sc, O, K, ctx.training)
                dsc = dsc.view(ctx.sshape)
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=g.device)
            gemm(M, K, O, A=g2, a_rs=O, a_cs=1, a_scale=sc, B=w_b, b_rs=K, b_cs=1, b_exact=True, C=dx, ldc=K)
            dx = dx.view(ctx.xshape)
        beside.join(dW, dsc)
        return dx, dW, dsc, None, None


class BinLinear(torch.autograd.Function):
    """y = (sign(x+beta) sign(W)^T) * scale (+ b): ternary XNOR/popcount (sv_layers.py:35-51 with bw and ba)."""

    @staticmethod
    def forward(ctx, x, W, beta, scale, bias, training=True):
        global _FUSED_COLSUMS
        _hip(x, W, beta, scale, bias)
        ctx.training = bool(training)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        W_in, W = W, _f32c(W).reshape(W.shape[0], -1)          # [O,K] (a Conv1d weight [O,K,1] is the same memory)
        M, K = x2.shape
        O = W.shape[0]
        KW = _words(K)
        dev = x.device
        need_grad = any(ctx.needs_input_grad)
        sc = _f32c(scale).view(-1)
        bt = _f32c(beta).view(-1)
        use_mfma = bool(config.BINLINEAR_MFMA and M >= 1024 and O >= 64)
        packed = _binweight(W_in, scale, i8=use_mfma)
        w_sign, w_nz, w_b = packed["w_sign"], packed["w_nz"], packed["w_b"]
        planes = ([torch.empty(((M + 63) // 64, K), dtype=torch.int64, device=dev) for _ in range(3)] if (need_grad or TAP is not None)
                  else [None] * 3)
        y = torch.empty((M, O), dtype=torch.float32, device=dev)
        if use_mfma:
            # many rows: int8 ternary operands on the matrix cores (same integer counts: identical outputs and planes).  In training
            # the kernel also leaves the column sums of y for the BatchNorm that follows (sv_layers.py:189): _batch_stats picks them
            # up instead of reading y again
            sums = _zeros((_sliced_len(2 * O),), torch.float64, dev) if (training and config.FUSE_BN_STATS and K <= 46340) else None
            call("svnet_binlinear_i8_fwd_f32", _p(x2), K, _p(bt), _p(packed["w_i8"]), _p(sc), _p(bias), M, K, O, _p(y),
                 _p(planes[0]), _p(planes[1]), _p(planes[2]), _p(sums), _stream())
            _FUSED_COLSUMS = (y, y._version, sums) if sums is not None else None
        else:
            _FUSED_COLSUMS = None                         # (a stale record would keep the previous producer's output alive)
            call("svnet_binlinear_fwd_f32", _p(x2), K, _p(bt), _p(w_sign), _p(w_nz), _p(sc), _p(bias), M, K, O, _p(y),
                 _p(planes[0]), _p(planes[1]), _p(planes[2]), _stream())
        if TAP is not None:
            TAP["signs"].append(("rows", M, K, planes))
        if need_grad:
            ctx.save_for_backward(W, sc, w_b, *planes)
        ctx.meta = (M, K, O, KW, x.shape, beta.shape, scale.shape, bias is not None, W_in.shape)
        ctx.defer_ok = defer_rows_wgrad.on
        return y.view(x.shape[:-1] + (O,))

    @staticmethod
    def backward(ctx, g):
        W, sc, w_b, x_sign, x_nz, x_ste = ctx.saved_tensors
        M, K, O, KW, xshape, bshape, sshape, has_bias, wshape = ctx.meta
        dev = g.device
        g2 = _f32c(g).reshape(M, O)
        dx = dW = dbeta = dsc = dbias = None
        need_x, need_w = ctx.needs_input_grad[0] or ctx.needs_input_grad[2], ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        beside = _Beside(dev, config.DW_BESIDE and need_x and need_w and ctx.training and M >= config.TWO_STREAM_MIN_ROWS)    # (see BwLinear.backward)
        def wgrad(sum_buf=None, sum_len=0):
            # GX[o,k] = sum_m g[m,o] x_b[m,k], computed as (x_b^T g)[k,o] with the ternary operand on the A side
            GX = _zeros((O, K), torch.float32, dev)           # (accumulate onto zeros from the step's arena: no zero-fill launch of its own)
            gemm(K, O, M, a_planes=(x_sign, x_nz), B=g2, b_rs=O, b_cs=1, C=GX, ldc=1, c_cs=K, accumulate=True)
            dW_, dsc_ = _binweight_grad(GX, W, sc, O, K, ctx.training, sum_buf=sum_buf, sum_len=sum_len)
            return dW_.view(wshape), dsc_.view(sshape)

        if need_w and beside.on:                              # helper-stream form: the weight gradient is issued first, beside the input gradient
            with beside:
                dW, dsc = wgrad()
        # TrainStep (svnet_amd.train) gathers the parameter gradients ONCE, after the backward: a big layer's weight-gradient chain then goes
        # to the TAIL of the side stream, unjoined (_Deferred) - the main stream carries on with what the layer in front waits for (dx) and
        # the chain fills the chip under the launch-bound kernels that follow it there (conv5 of the classifier: gate MLP, mean / Vector2Scalar
        # backward, the next layer's prelude)
        defer = (need_w and not beside.on and need_x and ctx.training and DEFERRED.active and config.DEFER_ROWS_WGRAD and ctx.defer_ok
                 and M >= config.TWO_STREAM_MIN_ROWS and DEFERRED.first_use(W, sc))
        dbuf = None
        if need_x:
            # (otherwise) the input gradient FIRST: its column sums (dL/dbeta, a sliced accumulator) are then totalled by the weight-gradient
            # epilogue launch below instead of by a launch of their own
            dbuf = _zeros((_sliced_len(K),), torch.float32, dev)
            if ctx.training:
                dx = torch.empty((M, K), dtype=torch.float32, device=dev)
                gemm(M, K, O, A=g2, a_rs=O, a_cs=1, a_scale=sc, B=w_b, b_rs=K, b_cs=1, b_exact=True, C=dx, ldc=K, mask=x_ste, col_sum=dbuf)
            else:   # eval: bare sign() has zero gradient (sv_layers.py:38-39)
                dx = torch.zeros((M, K), dtype=torch.float32, device=dev)
            dx = dx.view(xshape)
            dbeta = dbuf[:K].view(bshape)
        pending = need_x and ctx.training                     # the slices of dbuf still have to be added up
        if defer:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            ready = main.record_event()                       # (g2, the planes and dbuf's slices are complete on this stream)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                dW, dsc = wgrad(dbuf, K)                      # (its epilogue launch also totals dL/dbeta: a parameter gradient, like dW)
            DEFERRED.keep.append((g2, x_sign, x_nz, W, sc, dbuf))     # (NOT the returned gradients: see EdgeBlock.backward)
            pending = False
        elif need_w and not beside.on:
            dW, dsc = wgrad(dbuf if pending else None, K if pending else 0)
            pending = False
        if pending:
            call("svnet_slices_sum_f32", _p(dbuf), K, _stream())
        beside.join(dW, dsc)
        if has_bias and ctx.needs_input_grad[4]:
            dbias = pool_raw(g2, 1, M, O, 1)[0].view(O) * float(M)
        return dx, dW, dbeta, dsc, dbias, None


def _binweight_cols(W, scale, ranges, i8=False):
    """_binweight() of the columns W[:, a:b] for (a, b) in `ranges`, side by side, of a bw layer's weight [O,K] (a contiguous copy is
    packed; cached per (parameter, column set) like the whole-matrix forms and re-packed with them when the parameter changes)."""
    import weakref
    O, dev = W.shape[0], W.device
    ranges = tuple((int(a), int(b)) for a, b in ranges if b > a)
    Kb = sum(b - a for a, b in ranges)
    w_ref, s_ref = weakref.ref(W), (None if scale is None else weakref.ref(scale))

    def build():
        out = {"w_sign": torch.empty((O, _words(Kb)), dtype=torch.int64, device=dev), "w_nz": torch.empty((O, _words(Kb)), dtype=torch.int64, device=dev),
               "w_b": torch.empty((O, Kb), dtype=torch.float32, device=dev),
               "w_i8": torch.empty((_lib.lib().svnet_binweight_i8_bytes(O, Kb),), dtype=torch.int8, device=dev) if i8 else None}

        def rebuild():
            Wp = w_ref()
            if Wp is None or (s_ref is not None and s_ref() is None):
                return
            W2 = Wp.detach().reshape(O, -1)
            Wc = (W2[:, ranges[0][0]:ranges[0][1]] if len(ranges) == 1 else torch.cat([W2[:, a:b] for a, b in ranges], dim=1)).contiguous()
            call("svnet_binweight_prepare_f32", _p(Wc), None, O, Kb, _p(out["w_sign"]), _p(out["w_nz"]), _p(out["w_b"]), None, _stream())
            if out["w_i8"] is not None:
                call("svnet_binweight_pack_i8", _p(Wc), O, Kb, _p(out["w_i8"]), _stream())
        rebuild()
        return out, rebuild
    return PLANES.get("bwcols:%s:%d" % (",".join("%d-%d" % r for r in ranges), int(i8)), (W,) if scale is None else (W, scale), build)


def _cloud_planes_as_rows(planes, B, N, K):
    """Row-sliced planes [ceil(B/64), K] of B per-cloud rows -> the planes [ceil(B*N/64), K] of the B*N rows in which every cloud's row
    is repeated N times (test instrumentation only: what the binarized broadcast half of a concatenation looks like to the decision tap)."""
    dev = planes[0].device
    out = []
    shifts = torch.arange(64, device=dev, dtype=torch.int64)
    for pl in planes:
        bits = ((pl.view(-1, 1, K) >> shifts.view(1, 64, 1)) & 1).reshape(-1, K)[:B]                # [B,K]
        rows = bits.repeat_interleave(N, dim=0)                                                     # [B*N,K]
        pad = (-rows.shape[0]) % 64
        if pad:
            rows = torch.cat([rows, torch.zeros((pad, K), dtype=torch.int64, device=dev)], dim=0)
        out.append((rows.view(-1, 64, K) << shifts.view(1, 64, 1)).sum(dim=1))                      # disjoint bits: sum == or (wraps at bit 63)
    return out


class BinLinearCloud(torch.autograd.Function):
    """Linear(bw, ba) / Conv1d(binary) (sv_layers.py:35-51, :55-78) on rows whose leading Kc columns are CONSTANT over each cloud's N
    rows - the layer applied to cat[expand(x_cloud), x_point] (sv_dgcnn_partseg.py:115-121: conv8 on [glob | pooled | label] repeated
    over the points + the per-point feature) WITHOUT the concatenation:

        y[b,n,o] = scale[o] * (count(x_point[b,n] ; W[:, Kc:]) + count(x_cloud[b] ; W[:, :Kc]))

    The XNOR-popcount sum over a row is the sum over its two column blocks, so the outputs (and the BatchNorm sums the kernel leaves
    behind) are IDENTICAL to the product over the materialised [B*N, Kc+Kp] rows; the per-cloud block costs B rows instead of B*N
    (conv8 of sv_dgcnn_partseg: 1 600 of 2 144 columns).  Backward: dL/dy is summed over each cloud's rows once; the per-point block's
    input / weight gradients are the usual products over Kp columns, the per-cloud block's are products over B rows."""

    @staticmethod
    def forward(ctx, x_cloud, x_point, W, beta, scale, training=True, cloud_at=0):
        """cloud_at: first column of the per-cloud block in the layer's row (0: sv_dgcnn_partseg's conv8, whose per-cloud features come first;
        512: conv_fuse.linear1 of sv_pointnet_cls, cat[s, expand(pooled s), Vector2Scalar(v)]); the per-point columns keep their order."""
        global _FUSED_COLSUMS
        _hip(x_cloud, x_point, W, beta, scale)
        ctx.training = bool(training)
        B, N, Kp = x_point.shape
        Kc = x_cloud.shape[-1]
        k0 = int(cloud_at)
        xp = _f32c(x_point).reshape(B * N, Kp)
        xc = _f32c(x_cloud).reshape(B, Kc)
        W_in, Wc = W, _f32c(W).reshape(W.shape[0], -1)
        O, K = Wc.shape
        if K != Kc + Kp or not 0 <= k0 <= Kp:
            raise ValueError("BinLinearCloud: weight has %d columns, inputs %d + %d (cloud block at %d)" % (K, Kc, Kp, k0))
        M = B * N
        dev = xp.device
        need_grad = any(ctx.needs_input_grad)
        sc, bt = _f32c(scale).view(-1), _f32c(beta).view(-1)
        bt_c = bt[k0:k0 + Kc]
        bt_p = bt[Kc:] if k0 == 0 else torch.cat([bt[:k0], bt[k0 + Kc:]])
        pk_c = _binweight_cols(W_in, scale, [(k0, k0 + Kc)])
        pk_p = _binweight_cols(W_in, scale, [(0, k0), (k0 + Kc, K)], i8=True)
        keep = need_grad or TAP is not None
        pl_c = [torch.empty(((B + 63) // 64, Kc), dtype=torch.int64, device=dev) for _ in range(3)] if keep else [None] * 3
        pl_p = [torch.empty(((M + 63) // 64, Kp), dtype=torch.int64, device=dev) for _ in range(3)] if keep else [None] * 3
        # the per-cloud block: integer counts of the B rows (scale 1: the values are exact integers in fp32)
        ones = torch.ones((O,), dtype=torch.float32, device=dev)
        n_cloud = torch.empty((B, O), dtype=torch.float32, device=dev)
        call("svnet_binlinear_fwd_f32", _p(xc), Kc, _p(bt_c), _p(pk_c["w_sign"]), _p(pk_c["w_nz"]), _p(ones), None, B, Kc, O, _p(n_cloud),
             _p(pl_c[0]), _p(pl_c[1]), _p(pl_c[2]), _stream())
        y = torch.empty((M, O), dtype=torch.float32, device=dev)
        sums = _zeros((_sliced_len(2 * O),), torch.float64, dev) if (training and config.FUSE_BN_STATS and K <= 46340) else None
        call("svnet_binlinear_i8_cloud_fwd_f32", _p(xp), Kp, _p(bt_p), _p(pk_p["w_i8"]), _p(sc), None, M, Kp, O, _p(y),
             _p(pl_p[0]), _p(pl_p[1]), _p(pl_p[2]), _p(sums), _p(n_cloud), N, _stream())
        _FUSED_COLSUMS = (y, y._version, sums) if sums is not None else None
        if TAP is not None:      # the planes of the full [M, K] rows, as the layer over the materialised concatenation would have recorded them
            full = [torch.cat([b[:, :k0], a, b[:, k0:]], dim=1).contiguous() for a, b in zip(_cloud_planes_as_rows(pl_c, B, N, Kc), pl_p)]
            TAP["signs"].append(("rows", M, K, full))
        if need_grad:
            ctx.save_for_backward(Wc, sc, pk_c["w_b"], pk_p["w_b"], *pl_c, *pl_p)
        ctx.meta = (B, N, Kc, Kp, O, x_cloud.shape, x_point.shape, beta.shape, scale.shape, W_in.shape, k0)
        return y.view(B, N, O)

    @staticmethod
    def backward(ctx, g):
        Wc, sc, wb_c, wb_p, cs, cz, cq, ps, pz, pq = ctx.saved_tensors
        B, N, Kc, Kp, O, cshape, pshape, bshape, sshape, wshape, k0 = ctx.meta
        K, M = Kc + Kp, B * N
        dev = g.device
        g2 = _f32c(g).reshape(M, O)
        # dL/dy summed over each cloud's rows: all the per-cloud block ever sees of it
        gc = pool_raw(g2, B, N, O, 1)[0]
        gc = gc * float(N)
        need_w = ctx.needs_input_grad[2] or ctx.needs_input_grad[4]
        dxc = dxp = dbeta = dW = dsc = None
        dbuf_c = _zeros((_sliced_len(Kc),), torch.float32, dev)
        dbuf_p = _zeros((_sliced_len(Kp),), torch.float32, dev)
        if ctx.training:
            dxp = torch.empty((M, Kp), dtype=torch.float32, device=dev)
            gemm(M, Kp, O, A=g2, a_rs=O, a_cs=1, a_scale=sc, B=wb_p, b_rs=Kp, b_cs=1, b_exact=True, C=dxp, ldc=Kp, mask=pq, col_sum=dbuf_p)
            dxc = torch.empty((B, Kc), dtype=torch.float32, device=dev)
            gemm(B, Kc, O, A=gc, a_rs=O, a_cs=1, a_scale=sc, B=wb_c, b_rs=Kc, b_cs=1, b_exact=True, C=dxc, ldc=Kc, mask=cq, col_sum=dbuf_c)
            call("svnet_slices_sum_f32", _p(dbuf_p), Kp, _stream())
            call("svnet_slices_sum_f32", _p(dbuf_c), Kc, _stream())
        else:       # eval: bare sign() has zero gradient (sv_layers.py:38-39)
            dxp = torch.zeros((M, Kp), dtype=torch.float32, device=dev)
            dxc = torch.zeros((B, Kc), dtype=torch.float32, device=dev)
        dbeta = torch.cat([dbuf_p[:k0], dbuf_c[:Kc], dbuf_p[k0:Kp]]).view(bshape)
        if need_w:
            # GX[o, k] = sum_m g[m,o] x_b[m,k]: the per-cloud columns from the B summed rows, the per-point columns from all rows
            if k0 == 0:
                GX = _zeros((O, K), torch.float32, dev)
                gemm(Kc, O, B, a_planes=(cs, cz), B=gc, b_rs=O, b_cs=1, C=GX, ldc=1, c_cs=K, accumulate=True)
                gemm(Kp, O, M, a_planes=(ps, pz), B=g2, b_rs=O, b_cs=1, C=GX[:, Kc:], ldc=1, c_cs=K, accumulate=True)
            else:       # (the per-point columns are not one run of the row: products into their own buffers, one small cat)
                GXc, GXp = _zeros((O, Kc), torch.float32, dev), _zeros((O, Kp), torch.float32, dev)
                gemm(Kc, O, B, a_planes=(cs, cz), B=gc, b_rs=O, b_cs=1, C=GXc, ldc=1, c_cs=Kc, accumulate=True)
                gemm(Kp, O, M, a_planes=(ps, pz), B=g2, b_rs=O, b_cs=1, C=GXp, ldc=1, c_cs=Kp, accumulate=True)
                GX = torch.cat([GXp[:, :k0], GXc, GXp[:, k0:]], dim=1)
            dW, dsc = _binweight_grad(GX, Wc, sc, O, K, ctx.training)
            dW, dsc = dW.view(wshape), dsc.view(sshape)
        return dxc.view(cshape), dxp.view(pshape), dW, dbeta, dsc, None, None


class BinLinearBNAct(torch.autograd.Function):
    """act(BatchNorm1d(Linear(bw, ba)(x))) over M <= 64 rows (the classifier heads: sv_dgcnn_cls.py:76-78, sv_pointnet_cls.py:59-60) in
    one packing pass + ONE fused pass forward and two passes backward (csrc/head.hip) instead of ~12 launch-bound kernels per layer.
    Same integer counts, statistics and gradient formulas as BinLinear + BNAct; training = batch statistics + STE backward,
    eval = running statistics (forward only: callers that need eval-mode gradients take the layer-wise ops)."""

    @staticmethod
    def supported(M, K, O):
        return 1 <= M <= 64 and K <= 3776 and O >= 1

    @staticmethod
    def forward(ctx, x, W, beta, scale, gamma, bn_beta, running_mean, running_var, training, act, slope, nbt=None, eps=BN_EPS,
                momentum=BN_MOMENTUM, want_grad=True):
        _hip(x, W, beta, scale, gamma, bn_beta)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        W_in, Wc = W, _f32c(W).reshape(W.shape[0], -1)
        M, K = x2.shape
        O = Wc.shape[0]
        KW = _words(K)
        dev = x.device
        need_grad = bool(want_grad) and any(ctx.needs_input_grad)      # (want_grad: the caller's grad mode - forward() itself always runs under no_grad)
        if need_grad and not training:
            raise RuntimeError("svnet_amd: BinLinearBNAct has no eval-mode backward (use BinLinear + BNAct)")
        packed = _binweight(W_in, scale)
        sc, bt = _f32c(scale).view(-1), _f32c(beta).view(-1)
        keep = need_grad or TAP is not None
        rowp = torch.empty((3, M, KW), dtype=torch.int64, device=dev)            # sign | non-zero | STE, row-major words
        colp = torch.empty((3, K), dtype=torch.int64, device=dev) if keep else None   # the same planes as column words (bit m = row m)
        call("svnet_binhead_pack_f32", _p(x2), _p(bt), M, K, _p(rowp[0]), _p(rowp[1]), _p(rowp[2]),
             _p(colp[0]) if keep else None, _p(colp[1]) if keep else None, _p(colp[2]) if keep else None, _stream())
        y = torch.empty((M, O), dtype=torch.float32, device=dev)
        out = torch.empty((M, O), dtype=torch.float32, device=dev)
        stats = torch.empty((2, O), dtype=torch.float32, device=dev)
        d = BinHeadDesc()
        d.M, d.K, d.O = M, K, O
        d.w_sign, d.w_nz, d.wld = _p(packed["w_sign"]), _p(packed["w_nz"]), KW
        d.scale, d.gamma, d.bn_beta = _p(sc), _p(gamma), _p(bn_beta)
        d.running_mean, d.running_var, d.nbt = _p(running_mean), _p(running_var), _p(nbt)
        d.training, d.eps, d.momentum, d.act, d.slope = int(bool(training)), eps, momentum, int(act), slope
        d.x_sign, d.x_nz = _p(rowp[0]), _p(rowp[1])
        d.y, d.mean, d.invstd, d.out = _p(y), _p(stats[0]), _p(stats[1]), _p(out)
        call("svnet_binhead_fwd_f32", ctypes.byref(d), _stream())
        if TAP is not None:
            TAP["signs"].append(("rows", M, K, [colp[0].view(1, K), colp[1].view(1, K), colp[2].view(1, K)]))
            _tap_act(gamma, act, out)
        if need_grad:
            ctx.save_for_backward(Wc, sc, packed["w_b"], rowp, colp, y, stats, gamma, bn_beta)
        ctx.meta = (M, K, O, KW, x.shape, beta.shape, scale.shape, W_in.shape, int(act), slope)
        return out.view(x.shape[:-1] + (O,))

    @staticmethod
    def backward(ctx, g):
        Wc, sc, w_b, rowp, colp, y, stats, gamma, bn_beta = ctx.saved_tensors
        M, K, O, KW, xshape, bshape, sshape, wshape, act, slope = ctx.meta
        dev = g.device
        g2 = _f32c(g).reshape(M, O)
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        need_w = ctx.needs_input_grad[1]
        dnT = torch.empty((O, 64), dtype=torch.float32, device=dev)
        dW = torch.empty((O, K), dtype=torch.float32, device=dev) if need_w else None
        small = torch.empty((3, O), dtype=torch.float32, device=dev)             # dscale | dgamma | dbeta(bn)
        dx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_x else None
        dbeta = torch.empty((K,), dtype=torch.float32, device=dev) if need_x else None
        d = BinHeadDesc()
        d.M, d.K, d.O = M, K, O
        d.W, d.w_b, d.wld = _p(Wc), _p(w_b), KW
        d.scale, d.gamma, d.bn_beta = _p(sc), _p(gamma), _p(bn_beta)
        d.training, d.act, d.slope = 1, act, slope
        d.x_ste, d.xc_sign, d.xc_nz = _p(rowp[2]), _p(colp[0]), _p(colp[1])
        d.y, d.mean, d.invstd = _p(y), _p(stats[0]), _p(stats[1])
        d.g, d.dnT, d.dW = _p(g2), _p(dnT), _p(dW)
        d.dscale, d.dgamma, d.dbn_beta = _p(small[0]), _p(small[1]), _p(small[2])
        d.dx, d.dbeta_in = _p(dx), _p(dbeta)
        call("svnet_binhead_bwd_f32", ctypes.byref(d), _stream())
        return (dx.view(xshape) if dx is not None else None, dW.view(wshape) if dW is not None else None,
                dbeta.view(bshape) if dbeta is not None else None, small[0].view(sshape), small[1], small[2],
                None, None, None, None, None, None, None, None, None)


# ----------------------------------------------------------------------------- Vector2Scalar

class V2S(torch.autograd.Function):
    """Vector2Scalar (sv_layers.py:111-129). Returns (s [...,C*J], z [...,3,J])."""

    @staticmethod
    def forward(ctx, v, W, scale, training=True):
        _hip(v, W, scale)
        ctx.training = bool(training)
        ctx.set_materialize_grads(False)
        v3 = _f32c(v).reshape(-1, 3, v.shape[-1])
        W_in, W = W, _f32c(W)
        M, _, C = v3.shape
        J = W.shape[0]
        if scale is not None:
            sc = _f32c(scale).view(-1)
            w_eff = _binweight(W_in, scale)["w_eff"]
        else:
            sc, w_eff = None, W
        s = torch.empty((M, C * J), dtype=torch.float32, device=v.device)
        z = torch.empty((M, 3, J), dtype=torch.float32, device=v.device)
        call("svnet_v2s_fwd_f32", _p(v3), _p(w_eff), M, C, J, _p(s), _p(z), _stream())
        ctx.save_for_backward(v3, W, w_eff, sc)
        ctx.vshape = v.shape
        ctx.sshape = None if scale is None else scale.shape
        lead = v.shape[:-2]
        return s.view(lead + (C * J,)), z.view(lead + (3, J))

    @staticmethod
    def backward(ctx, gs, gz):
        v3, W, w_eff, sc = ctx.saved_tensors
        M, _, C = v3.shape
        J = W.shape[0]
        L = _lib.lib()
        gs2 = torch.zeros((M, C * J), dtype=torch.float32, device=v3.device) if gs is None else _f32c(gs).reshape(M, C * J)
        gz2 = None if gz is None else _f32c(gz).reshape(M, 3, J)
        dv = torch.empty_like(v3)
        gxb = _zeros((_sliced_len(J * C),), torch.float32, v3.device)      # sliced accumulator: 2 048 workgroups add to it
        call("svnet_v2s_bwd_f32", _p(v3), _p(w_eff), _p(gs2), _p(gz2), M, C, J, _p(dv), _p(gxb), _stream())
        if sc is not None:
            dW, dsc = _binweight_grad(gxb, W, sc, J, C, ctx.training, gx_sliced=True)      # (adds the slices up on the way in)
            dsc = dsc.view(ctx.sshape)
        else:
            call("svnet_slices_sum_f32", _p(gxb), J * C, _stream())
            dW, dsc = gxb[:J * C].view(J, C), None
        return dv.view(ctx.vshape), dW, dsc, None


class V2SCat(torch.autograd.Function):
    """cat[s, Vector2Scalar(v)] along the last axis - the input of an SVBlock's linear1 (sv_layers.py:187-188) - written in place by
    the Vector2Scalar kernel (svnet_v2s_cat_fwd_f32): no [.., C*J] intermediate and no cat pass; the backward hands the s part of
    the gradient on as a view and reads the Vector2Scalar part where it lies (svnet_v2s_bwd_ld_f32)."""

    @staticmethod
    def forward(ctx, s, v, W, scale, training=True, clouds=0, gate_W0=None, gate_W2=None):
        """clouds > 0: also returns mean(s) over each cloud's rows [clouds, Cs] (the gate's input, sv_layers.py:179): s then has ONE
        consumer in the autograd graph, and the backward writes its gradient once - the cat gradient's s columns plus the mean's
        broadcast - instead of a broadcast pass and a strided add of two gradients in front of the layer's backward.
        With gate_W0 / gate_W2 (the block's gate MLP, sv_layers.py:156-161): returns (cat, gate) - the column sums of s come out of the
        concatenation kernel's own copy of s (fp64, svnet_v2s_cat_sum_fwd_f32) and the MLP starts from them: no pooling pass over s."""
        _hip(s, v, W, scale)
        ctx.training = bool(training)
        v3 = _f32c(v).reshape(-1, 3, v.shape[-1])
        M, _, C = v3.shape
        s2 = _f32c(s).reshape(M, s.shape[-1])
        Cs = s2.shape[1]
        W_in, W = W, _f32c(W)
        J = W.shape[0]
        if scale is not None:
            sc = _f32c(scale).view(-1)
            w_eff = _binweight(W_in, scale)["w_eff"]
        else:
            sc, w_eff = None, W
        out = torch.empty((M, Cs + C * J), dtype=torch.float32, device=v.device)
        ctx.meta = (M, C, J, Cs, s.shape, v.shape, None if scale is None else scale.shape)
        ctx.clouds = int(clouds)
        ctx.set_materialize_grads(False)
        cat = out.view(s.shape[:-1] + (Cs + C * J,))
        ctx.with_gate = False
        if clouds and gate_W0 is not None and _lib.lib().svnet_v2s_cat_sum_supported(M, C, Cs, M // clouds):
            rows = M // clouds
            s_sum = _zeros((clouds, Cs), torch.float64, v.device)
            call("svnet_v2s_cat_sum_fwd_f32", _p(v3), _p(w_eff), _p(s2), Cs, M, C, J, _p(out), Cs + C * J, _p(s_sum), rows, _stream())
            W0c, W2c = _f32c(gate_W0), _f32c(gate_W2)
            H, Ov = W0c.shape[0], W2c.shape[0]
            pooled = torch.empty((clouds, Cs), dtype=torch.float32, device=v.device)        # = float(s_sum): the MLP scales it by 1 / rows
            h = torch.empty((clouds, H), dtype=torch.float32, device=v.device)
            gate = torch.empty((clouds, Ov), dtype=torch.float32, device=v.device)
            call("svnet_gate_mlp_fwd_f32", None, _p(s_sum), _p(pooled), 1.0 / float(rows), _p(W0c), _p(W2c), clouds, Cs, H, Ov, _p(h), _p(gate),
                 None, 0, _stream())
            _tap_act(gate_W0, 2, h)
            ctx.save_for_backward(v3, W, w_eff, sc, pooled, W0c, W2c, h, gate)
            ctx.with_gate = True
            ctx.gate_in_scale = 1.0 / float(rows)
            return cat, gate
        call("svnet_v2s_cat_fwd_f32", _p(v3), _p(w_eff), _p(s2), Cs, M, C, J, _p(out), Cs + C * J, _stream())
        ctx.save_for_backward(v3, W, w_eff, sc)
        if clouds:
            s_mean, _ = pool_raw(s2, clouds, M // clouds, Cs, 1)
            if gate_W0 is not None:                          # (shape outside the summing kernel's: the mean by its own pass, the MLP here)
                W0c, W2c = _f32c(gate_W0), _f32c(gate_W2)
                H, Ov = W0c.shape[0], W2c.shape[0]
                h = torch.empty((clouds, H), dtype=torch.float32, device=v.device)
                gate = torch.empty((clouds, Ov), dtype=torch.float32, device=v.device)
                call("svnet_gate_mlp_fwd_f32", _p(s_mean), None, None, 1.0, _p(W0c), _p(W2c), clouds, Cs, H, Ov, _p(h), _p(gate), None, 0, _stream())
                _tap_act(gate_W0, 2, h)
                ctx.save_for_backward(v3, W, w_eff, sc, s_mean, W0c, W2c, h, gate)
                ctx.with_gate = True
                ctx.gate_in_scale = 1.0
                return cat, gate
            return cat, s_mean
        return cat

    @staticmethod
    def backward(ctx, g, g_mean=None):
        dW0 = dW2 = None
        if ctx.with_gate:        # the second output was the gate: its MLP's backward first - g_mean becomes dL/d(mean of s)
            v3, W, w_eff, sc, pooled, W0g, W2g, hg, gateg = ctx.saved_tensors
            if g_mean is not None:
                Bc, Cin = pooled.shape
                H, Ov = W0g.shape[0], W2g.shape[0]
                zb = _zeros((H * Cin + Ov * H,), torch.float32, pooled.device)
                dW0, dW2 = zb[:H * Cin].view(H, Cin), zb[H * Cin:].view(Ov, H)
                dmean = torch.empty_like(pooled)
                call("svnet_gate_mlp_bwd_f32", _p(_f32c(g_mean)), _p(gateg), _p(hg), _p(pooled), ctx.gate_in_scale, _p(W0g), _p(W2g), Bc, Cin, H, Ov,
                     1.0, _p(dmean), _p(dW0), _p(dW2), _stream())
                g_mean = dmean
        else:
            v3, W, w_eff, sc = ctx.saved_tensors
        M, C, J, Cs, sshape, vshape, scshape = ctx.meta
        if g is None:                                   # (only the mean was used)
            g = torch.zeros((M, Cs + C * J), dtype=torch.float32, device=v3.device)
        g2 = _f32c(g).reshape(M, Cs + C * J)
        ds = None
        if ctx.needs_input_grad[0]:
            if g_mean is not None:
                ds = torch.empty((M, Cs), dtype=torch.float32, device=g2.device)
                call("svnet_pool_mean_bwd_add_f32", _p(_f32c(g_mean).reshape(ctx.clouds, Cs)), _p(g2), Cs + C * J, ctx.clouds, M // ctx.clouds, Cs,
                     _p(ds), _stream())
                ds = ds.reshape(sshape)
            else:
                ds = g2[:, :Cs].reshape(sshape)          # (a strided view: its consumers read it in place)
        dv = torch.empty_like(v3)
        gxb = _zeros((_sliced_len(J * C),), torch.float32, v3.device)
        call("svnet_v2s_bwd_ld_f32", _p(v3), _p(w_eff), _p(g2[:, Cs:]), Cs + C * J, None, M, C, J, _p(dv), _p(gxb), _stream())
        if sc is not None:
            dW, dsc = _binweight_grad(gxb, W, sc, J, C, ctx.training, gx_sliced=True)
            dsc = dsc.view(scshape)
        else:
            call("svnet_slices_sum_f32", _p(gxb), J * C, _stream())
            dW, dsc = gxb[:J * C].view(J, C), None
        return ds, dv.view(vshape), dW, dsc, None, None, dW0, dW2


class VProject(torch.autograd.Function):
    """s[..., c*J+j] = sum_i v[..., i, c] * z[..., i, j] for a given per-row frame z (the back-projection einsum
    'bimj,bijk->bimk' of sv_pointnet_partseg.py:89, flattened)."""

    @staticmethod
    def forward(ctx, v, z):
        _hip(v, z)
        v3 = _f32c(v).reshape(-1, 3, v.shape[-1])
        z3 = _f32c(z).reshape(-1, 3, z.shape[-1])
        M, _, C = v3.shape
        J = z3.shape[-1]
        s = torch.empty((M, C * J), dtype=torch.float32, device=v.device)
        call("svnet_vproject_fwd_f32", _p(v3), _p(z3), M, C, J, _p(s), _stream())
        ctx.save_for_backward(v3, z3)
        ctx.shapes = (v.shape, z.shape)
        return s.view(v.shape[:-2] + (C * J,))

    @staticmethod
    def backward(ctx, gs):
        v3, z3 = ctx.saved_tensors
        M, _, C = v3.shape
        J = z3.shape[-1]
        gs2 = _f32c(gs).reshape(M, C * J)
        dv, dz = torch.empty_like(v3), torch.empty_like(z3)
        call("svnet_vproject_bwd_f32", _p(v3), _p(z3), _p(gs2), M, C, J, _p(dv), _p(dz), _stream())
        return dv.view(ctx.shapes[0]), dz.view(ctx.shapes[1])


# ----------------------------------------------------------------------------- normalisation

# Column sums a producer left for the BatchNorm over its output: (the [M, C] tensor itself - held, so that its memory cannot be handed to
# another tensor while the record lives -, its version at the time, sums [2C] double).  Consumed once; any later producer replaces it.
_FUSED_COLSUMS = None
_FUSED_VSTATS = None          # (y, version, sums) of the last BwLinear(vstats=True): the VectorBN sums of ITS output


def _batch_stats(x, M, C, kind, running_mean, running_var, training, momentum, eps, nbt=None):
    global _FUSED_COLSUMS
    L = _lib.lib()
    dev = x.device
    mean = torch.empty((C,), dtype=torch.float32, device=dev)
    invstd = torch.empty((C,), dtype=torch.float32, device=dev)
    if training:
        rec, _FUSED_COLSUMS = _FUSED_COLSUMS, None
        if (kind == 0 and rec is not None and rec[0].data_ptr() == x.data_ptr() and tuple(rec[0].shape) == (M, C)
                and rec[0]._version == rec[1] and x.is_contiguous()):
            sums = rec[2]                                       # the producing kernel's sums (exact integer counts, svnet_binlinear_i8_fwd_f32)
        else:
            sums = _zeros((_sliced_len(2 * C),), torch.float64, dev)
            call("svnet_colstats_f64", _p(x), M, C, kind, _p(sums), _stream())
        call("svnet_bn_finalize_f32", _p(sums), M, C, eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var),
                                      _p(nbt), _stream())
    else:
        call("svnet_bn_eval_stats_f32", _p(running_mean), _p(running_var), C, eps, _p(mean), _p(invstd), _stream())
    return mean, invstd


def _tap_act(gamma, act, out):
    """Decision tap (tests): which side of its kink every ReLU / LeakyReLU output of a BatchNorm + activation layer lies on, keyed by
    the BatchNorm's weight (tests/decisions.py maps it to the parameter's name); the hidden layer of a gate MLP likewise, keyed by its
    first weight."""
    if TAP is not None and "acts" in TAP and act in (1, 2):
        TAP["acts"].append((gamma.data_ptr(), (out > 0).reshape(-1, out.shape[-1])))


class BNAct(torch.autograd.Function):
    """BatchNorm1d over rows (+ LeakyReLU / ReLU): sv_layers.py:189-190, sv_dgcnn_cls.py:76-78."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, act, slope, nbt=None, eps=BN_EPS, momentum=BN_MOMENTUM):
        _hip(x, gamma, beta)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        M, C = x2.shape
        mean, invstd = _batch_stats(x2, M, C, 0, running_mean, running_var, training, momentum, eps, nbt)
        y = torch.empty_like(x2)
        call("svnet_bn_act_fwd_f32", _p(x2), _p(mean), _p(invstd), _p(gamma), _p(beta), M, C, act, slope, _p(y), _stream())
        _tap_act(gamma, act, y)
        ctx.save_for_backward(x2, mean, invstd, gamma, beta)
        ctx.meta = (M, C, act, slope, bool(training), x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, invstd, gamma, beta = ctx.saved_tensors
        M, C, act, slope, training, xshape = ctx.meta
        L = _lib.lib()
        g2 = _f32c(g).reshape(M, C)
        red = _zeros((_sliced_len(2 * C),), torch.float32, g.device)
        call("svnet_bn_act_bwd_reduce_f32", _p(g2), _p(x2), _p(mean), _p(invstd), _p(gamma), _p(beta), M, C, act, slope, _p(red),
                                            _stream())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            call("svnet_bn_act_bwd_apply_f32", _p(g2), _p(x2), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(red), M, C, act, slope,
                                               int(training), _p(dx), _stream())
            dx = dx.view(xshape)
        else:
            call("svnet_slices_sum_f32", _p(red), 2 * C, _stream())      # (the apply pass leaves the totals in red[0:2C]; without it: here)
        return dx, red[C:2 * C], red[:C], None, None, None, None, None, None, None, None


class VBN(torch.autograd.Function):
    """VectorBN (+ gate): out = v * BN(|v|+eps) / (|v|+eps) * gate   (sv_layers.py:86-102, :194)."""

    @staticmethod
    def forward(ctx, v, gamma, beta, running_mean, running_var, gate, rows_per_batch, training, nbt=None, eps=BN_EPS, momentum=BN_MOMENTUM):
        _hip(v, gamma, beta, gate)
        v3 = _f32c(v).reshape(-1, 3, v.shape[-1])
        M, _, C = v3.shape
        gate2 = None if gate is None else _f32c(gate).reshape(-1, C)
        out = torch.empty_like(v3)
        global _FUSED_VSTATS
        rec, _FUSED_VSTATS = _FUSED_VSTATS, None
        if training and M > 0:
            # statistics pass, then ONE kernel that finalises them per thread and applies them (no one-workgroup launch in between)
            if (rec is not None and rec[0].data_ptr() == v3.data_ptr() and rec[0].numel() == v3.numel() and rec[0]._version == rec[1]
                    and tuple(rec[0].shape) == (3 * M, C)):
                sums = rec[2]                                   # the producing product's sums (csrc/vlinear.hip)
            else:
                sums = _zeros((_sliced_len(2 * C),), torch.float64, v3.device)
                call("svnet_colstats_f64", _p(v3), M, C, 1, _p(sums), _stream())
            mean = torch.empty((C,), dtype=torch.float32, device=v3.device)
            invstd = torch.empty((C,), dtype=torch.float32, device=v3.device)
            call("svnet_vbn_fwd_stats_f32", _p(v3), _p(sums), eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt),
                 _p(gamma), _p(beta), _p(gate2), rows_per_batch, M, C, _p(out), _stream())
        else:
            mean, invstd = _batch_stats(v3, M, C, 1, running_mean, running_var, training, momentum, eps, nbt)
            call("svnet_vbn_fwd_f32", _p(v3), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(gate2), rows_per_batch, M, C, _p(out),
                                               _stream())
        ctx.save_for_backward(v3, mean, invstd, gamma, beta, gate2)
        ctx.meta = (M, C, rows_per_batch, bool(training), v.shape, None if gate is None else gate.shape)
        return out.view(v.shape)

    @staticmethod
    def backward(ctx, g):
        v3, mean, invstd, gamma, beta, gate2 = ctx.saved_tensors
        M, C, rpb, training, vshape, gshape = ctx.meta
        L = _lib.lib()
        g3 = _f32c(g).reshape(M, 3, C)
        if gate2 is None:
            red, dgate = _zeros((_sliced_len(2 * C),), torch.float32, g.device), None
        else:
            red, dgate = _zeros_pool(g.device, ((_sliced_len(2 * C),), torch.float32), (tuple(gate2.shape), torch.float32))
        call("svnet_vbn_bwd_reduce_f32", _p(g3), _p(v3), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(gate2), rpb, M, C, _p(red),
                                         _p(dgate), _stream())
        dv = None
        if ctx.needs_input_grad[0]:
            dv = torch.empty_like(v3)
            call("svnet_vbn_bwd_apply_f32", _p(g3), _p(v3), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(gate2), _p(red), rpb, M, C,
                                            int(training), _p(dv), _stream())
            dv = dv.view(vshape)
        else:
            call("svnet_slices_sum_f32", _p(red), 2 * C, _stream())
        if dgate is not None:
            dgate = dgate.view(gshape)
        return dv, red[C:2 * C], red[:C], None, None, dgate, None, None, None, None, None


# ----------------------------------------------------------------------------- pooling / activations / loss

def pool_raw(x, outer, R, inner, mode, out=None):
    """x contiguous viewed as [outer,R,inner] -> (out [outer,inner], argmax int32 or None).  `out`: an [outer, inner] column slice
    of a wider row-major tensor (stride(0) = its row length) to write into instead of a fresh tensor."""
    if out is None:
        out = torch.empty((outer, inner), dtype=torch.float32, device=x.device)
    arg = torch.empty((outer, inner), dtype=torch.int32, device=x.device) if mode == 0 else None
    nbytes = _lib.lib().svnet_pool_workspace_bytes(outer, R, inner, mode)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device) if nbytes else None
    call("svnet_pool_fwd_f32", _p(x), outer, R, inner, mode, _p(out), out.stride(0), _p(arg), _p(ws), nbytes, _stream())
    return out, arg


def pool_maxmean_raw(x, outer, R, inner, out_max, out_mean):
    """[max | mean] of x viewed as [outer,R,inner] into two [outer, inner] column slices of one row-major tensor (same row stride);
    returns the arg-max.  One pass over x when the reduced axis is long, two pool_raw calls otherwise."""
    L = _lib.lib()
    nb0, nb1 = L.svnet_pool_workspace_bytes(outer, R, inner, 0), L.svnet_pool_workspace_bytes(outer, R, inner, 1)
    if R >= 256 and nb0 and nb1 and out_max.stride(0) == out_mean.stride(0):
        arg = torch.empty((outer, inner), dtype=torch.int32, device=x.device)
        ws = torch.empty((nb0 + nb1,), dtype=torch.uint8, device=x.device)
        call("svnet_pool_maxmean_fwd_f32", _p(x), outer, R, inner, _p(out_max), _p(out_mean), out_max.stride(0), _p(arg), _p(ws), nb0 + nb1,
             _stream())
        return arg
    _, arg = pool_raw(x, outer, R, inner, 0, out=out_max)
    pool_raw(x, outer, R, inner, 1, out=out_mean)
    return arg


class Pool(torch.autograd.Function):
    """max (first index on ties) or mean over one axis (sv_util.py:125-131)."""

    @staticmethod
    def forward(ctx, x, dim, mode):
        _hip(x)
        x = _f32c(x)
        dim = dim % x.dim()
        outer = 1
        for d in x.shape[:dim]:
            outer *= d
        R = x.shape[dim]
        inner = 1
        for d in x.shape[dim + 1:]:
            inner *= d
        out, arg = pool_raw(x, outer, R, inner, mode)
        if arg is not None:
            ctx.save_for_backward(arg)
            if TAP is not None:
                TAP["pools"].append(arg)
        ctx.meta = (outer, R, inner, mode, x.shape)
        return out.view(x.shape[:dim] + x.shape[dim + 1:])

    @staticmethod
    def backward(ctx, g):
        outer, R, inner, mode, xshape = ctx.meta
        arg = ctx.saved_tensors[0] if mode == 0 else None
        g2 = _f32c(g).reshape(outer, inner)
        dx = torch.empty(xshape, dtype=torch.float32, device=g.device)
        call("svnet_pool_bwd_f32", _p(g2), _p(arg), outer, R, inner, mode, _p(dx), _stream())
        return dx, None, None


class PoolMaxParts(torch.autograd.Function):
    """max over the points of cat[a, b] along the channels, computed from the two PARTS where they lie (the maximum of a concatenation is
    the concatenation of the maxima): a [B,N,Ca], b [B,N,Cb] -> [B, Ca+Cb] - sv_dgcnn_partseg.py:107-108 pools svfuse3's [B,N,1016]
    feature at once, so that concatenation is never built (nor its gradient sliced).  One decision record, as the pooled concatenation's."""

    @staticmethod
    def forward(ctx, a, b):
        _hip(a, b)
        a, b = _f32c(a), _f32c(b)
        B, N, Ca = a.shape
        Cb = b.shape[-1]
        out = torch.empty((B, Ca + Cb), dtype=torch.float32, device=a.device)
        _, arg_a = pool_raw(a, B, N, Ca, 0, out=out[:, :Ca])
        _, arg_b = pool_raw(b, B, N, Cb, 0, out=out[:, Ca:])
        ctx.save_for_backward(arg_a, arg_b)
        if TAP is not None:
            TAP["pools"].append(torch.cat([arg_a, arg_b], dim=1))
        ctx.meta = (B, N, Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, Ca, Cb = ctx.meta
        arg_a, arg_b = ctx.saved_tensors
        g = _f32c(g)
        ga, gb = g[:, :Ca].contiguous(), g[:, Ca:].contiguous()
        da = torch.empty((B, N, Ca), dtype=torch.float32, device=g.device)
        db = torch.empty((B, N, Cb), dtype=torch.float32, device=g.device)
        call("svnet_pool_bwd_f32", _p(ga), _p(arg_a), B, N, Ca, 0, _p(da), _stream())
        call("svnet_pool_bwd_f32", _p(gb), _p(arg_b), B, N, Cb, 0, _p(db), _stream())
        return da, db


class PoolMaxMean(torch.autograd.Function):
    """cat(max, mean) over one axis of x (the global pooling of the classifiers, sv_dgcnn_cls.py:72-74): the two reductions
    share the backward pass, so the gradient of x is written once instead of written twice and added."""

    @staticmethod
    def forward(ctx, x, dim):
        _hip(x)
        x = _f32c(x)
        dim = dim % x.dim()
        outer = 1
        for d in x.shape[:dim]:
            outer *= d
        R = x.shape[dim]
        inner = 1
        for d in x.shape[dim + 1:]:
            inner *= d
        out = torch.empty((outer, 2 * inner), dtype=torch.float32, device=x.device)
        arg = pool_maxmean_raw(x, outer, R, inner, out[:, :inner], out[:, inner:])
        ctx.save_for_backward(arg)
        if TAP is not None:
            TAP["pools"].append(arg)
        ctx.meta = (outer, R, inner, x.shape)
        return out.view(x.shape[:dim] + (2 * inner,)) if x.dim() - dim == 2 else out

    @staticmethod
    def backward(ctx, g):
        outer, R, inner, xshape = ctx.meta
        (arg,) = ctx.saved_tensors
        g2 = _f32c(g).reshape(outer, 2 * inner)
        dx = torch.empty(xshape, dtype=torch.float32, device=g.device)
        call("svnet_pool_maxmean_bwd_f32", _p(g2), _p(g2[:, inner:]), 2 * inner, _p(arg), outer, R, inner, _p(dx), _stream())
        return dx, None


class GlobalMaxMeanPool(torch.autograd.Function):
    """The classifiers' global pooling of cat[s, s_v] over the points (sv_dgcnn_cls.py:69-74: adaptive max pool | adaptive avg pool
    of the [B,N,1022] feature) computed from the two PARTS of that feature where they lie: out [B, 2*(Ca+Cb)] =
    [max a | max b | mean a | mean b].  Four pooling launches write straight into column slices of `out` and two backward launches
    read column slices of its gradient: no concatenated feature, no slice / cat / zero-fill / add glue in either direction."""

    @staticmethod
    def forward(ctx, a, b):
        _hip(a, b)
        a, b = _f32c(a), _f32c(b)
        B, N, Ca = a.shape
        Cb = b.shape[-1]
        C = Ca + Cb
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=a.device)
        main, side = torch.cuda.current_stream(a.device), _side_stream(a.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):                                   # the two parts are independent: b beside a
            arg_b = pool_maxmean_raw(b, B, N, Cb, out[:, Ca:C], out[:, C + Ca:])
            arg_b.record_stream(main)
        arg_a = pool_maxmean_raw(a, B, N, Ca, out[:, :Ca], out[:, C:C + Ca])
        main.wait_stream(side)
        ctx.save_for_backward(arg_a, arg_b)
        if TAP is not None:
            TAP["pools"].append(torch.cat([arg_a, arg_b], dim=1))
        ctx.meta = (B, N, Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, Ca, Cb = ctx.meta
        arg_a, arg_b = ctx.saved_tensors
        C = Ca + Cb
        g = _f32c(g)
        da = torch.empty((B, N, Ca), dtype=torch.float32, device=g.device)
        db = torch.empty((B, N, Cb), dtype=torch.float32, device=g.device)
        main, side = torch.cuda.current_stream(g.device), _side_stream(g.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            call("svnet_pool_maxmean_bwd_f32", _p(g[:, Ca:]), _p(g[:, C + Ca:]), 2 * C, _p(arg_b), B, N, Cb, _p(db), _stream())
        call("svnet_pool_maxmean_bwd_f32", _p(g), _p(g[:, C:]), 2 * C, _p(arg_a), B, N, Ca, _p(da), _stream())
        main.wait_stream(side)
        return da, db


class GlobalMaxMeanPoolBN(torch.autograd.Function):
    """GlobalMaxMeanPool(batch_norm_act(bn, y, LeakyReLU), b) without the activated tensor: the classifier's conv5 output is only
    ever pooled over the points (sv_dgcnn_cls.py:69-74), so BatchNorm + LeakyReLU run inside the pooling pass over the pre-BN
    y [B,N,Ca] (svnet_bn_pool_fwd_f32) and the backward forms the activated tensor's gradient - (point == arg-max ? g_max : 0) +
    g_mean / N - on the fly from the pooled gradient (svnet_bn_pool_bwd_f32): one [B*N, Ca] tensor less in each direction.
    out [B, 2*(Ca+Cb)] = [max a | max b | mean a | mean b] with a = lrelu(bn(y)); values identical to the unfused chain."""

    @staticmethod
    def forward(ctx, y, b, gamma, beta, running_mean, running_var, training, act, slope, nbt=None, eps=BN_EPS, momentum=BN_MOMENTUM):
        _hip(y, b, gamma, beta)
        y, b = _f32c(y), _f32c(b)
        B, N, Ca = y.shape
        Cb = b.shape[-1]
        C = Ca + Cb
        L = _lib.lib()
        y2 = y.reshape(B * N, Ca)
        mean, invstd = _batch_stats(y2, B * N, Ca, 0, running_mean, running_var, training, momentum, eps, nbt)
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=y.device)
        main, side = torch.cuda.current_stream(y.device), _side_stream(y.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):                                   # the two parts are independent: b beside a
            arg_b = pool_maxmean_raw(b, B, N, Cb, out[:, Ca:C], out[:, C + Ca:])
            arg_b.record_stream(main)
        nb = L.svnet_pool_workspace_bytes(B, N, Ca, 0) + L.svnet_pool_workspace_bytes(B, N, Ca, 1)
        ws = _zeros((nb,), torch.uint8, y.device)          # (zero keys from the step's one fill: no memset launch in front of the pass)
        arg_a = torch.empty((B, Ca), dtype=torch.int32, device=y.device)
        call("svnet_bn_pool_fwd_f32", _p(y2), _p(mean), _p(invstd), _p(gamma), _p(beta), B, N, Ca, act, slope, _p(out), _p(out[:, C:]),
             2 * C, _p(arg_a), _p(ws), nb, 1, _stream())
        main.wait_stream(side)
        ctx.save_for_backward(y2, mean, invstd, gamma, beta, arg_a, arg_b)
        if TAP is not None:
            TAP["pools"].append(torch.cat([arg_a, arg_b], dim=1))
            if "acts" in TAP and act in (1, 2):      # the kink decisions of the BatchNorm + activation the pooling pass evaluates (same expression)
                TAP["acts"].append((gamma.data_ptr(), ((y2 - mean) * invstd * gamma + beta) > 0))
        ctx.meta = (B, N, Ca, Cb, act, slope, bool(training))
        return out

    @staticmethod
    def supported(B, N, Ca):
        L = _lib.lib()
        return N >= 256 and L.svnet_pool_workspace_bytes(B, N, Ca, 0) > 0 and L.svnet_pool_workspace_bytes(B, N, Ca, 1) > 0

    @staticmethod
    def backward(ctx, g):
        y2, mean, invstd, gamma, beta, arg_a, arg_b = ctx.saved_tensors
        B, N, Ca, Cb, act, slope, training = ctx.meta
        C = Ca + Cb
        g = _f32c(g)
        dev = g.device
        db = torch.empty((B, N, Cb), dtype=torch.float32, device=dev)
        dy = torch.empty((B, N, Ca), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        red = _zeros((_sliced_len(2 * Ca),), torch.float32, dev)
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            call("svnet_pool_maxmean_bwd_f32", _p(g[:, Ca:]), _p(g[:, C + Ca:]), 2 * C, _p(arg_b), B, N, Cb, _p(db), _stream())
        call("svnet_bn_pool_bwd_f32", _p(g), _p(g[:, C:]), 2 * C, _p(arg_a), _p(y2), _p(mean), _p(invstd), _p(gamma), _p(beta), B, N, Ca,
             act, slope, int(training), _p(red), _p(dy), _stream())
        main.wait_stream(side)
        return dy, db, red[Ca:2 * Ca], red[:Ca], None, None, None, None, None, None, None, None


class GlobalMaxMeanPoolBNV(torch.autograd.Function):
    """The classifier's whole tail behind conv5's two products (sv_dgcnn_cls.py:68-74), from what they leave in HBM:
        y     [B,N,Ca]    linear1's output BEFORE bn1 + LeakyReLU          (as GlobalMaxMeanPoolBN)
        v_lin [B,N,3,C]   linear2's output BEFORE VectorBN and the gate
    out [B, 2*(Ca+3C)] = [max a | max b | mean a | mean b] over the points, a = act(bn1(y)), b = svfuse's Vector2Scalar of
    VectorBN(v_lin) * gate.  The scalar half is GlobalMaxMeanPoolBN's; the vector half is ONE pass over v_lin each way
    (csrc/vtail.hip): VectorBN's output v5, the scalars s_v [B,N,3C] and their gradient are never written - layer by layer the half
    was three passes forward (VectorBN, Vector2Scalar, pooling: 335 MB) and three backward (401 MB before VectorBN's apply pass)."""

    @staticmethod
    def supported(B, N, Ca, C):
        return C <= 192 and GlobalMaxMeanPoolBN.supported(B, N, Ca)

    @staticmethod
    def forward(ctx, y, v_lin, gate, g1, b1, rm1, rv1, g2, b2, rm2, rv2, Wz, scz, training, act, slope, nbt1=None, nbt2=None,
                eps=BN_EPS, momentum=BN_MOMENTUM):
        global _FUSED_VSTATS
        _hip(y, v_lin, gate, g1, b1, g2, b2, Wz, scz)
        y, v_lin = _f32c(y), _f32c(v_lin)
        B, N, Ca = y.shape
        C = v_lin.shape[-1]
        Cb, P = 3 * C, B * N
        Ct = Ca + Cb
        L = _lib.lib()
        dev = y.device
        y2, v3 = y.reshape(P, Ca), v_lin.reshape(P, 3, C)
        gate2 = None if gate is None else _f32c(gate).reshape(B, C)
        W_in, Wzc = Wz, _f32c(Wz)
        if scz is not None:
            sczf = _f32c(scz).view(-1)
            w_eff = _binweight(W_in, scz)["w_eff"]
        else:
            sczf, w_eff = None, Wzc
        mean1, invstd1 = _batch_stats(y2, P, Ca, 0, rm1, rv1, training, momentum, eps, nbt1)
        out = torch.empty((B, 2 * Ct), dtype=torch.float32, device=dev)
        mean2 = torch.empty((C,), dtype=torch.float32, device=dev)
        invstd2 = torch.empty((C,), dtype=torch.float32, device=dev)
        arg_b = torch.empty((B, Cb), dtype=torch.int32, device=dev)
        nbw = L.svnet_vtail_workspace_bytes(B, N, C)
        wsb = _zeros((nbw,), torch.uint8, dev)              # (zero keys from the step's one fill: no memset launch in front of the pass)
        rec, _FUSED_VSTATS = _FUSED_VSTATS, None
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)                                          # (the gate and the statistics buffers come from this stream)
        with torch.cuda.stream(side):                                   # the vector half beside the scalar half
            sums = None
            if training:
                if (rec is not None and rec[0].data_ptr() == v3.data_ptr() and rec[0].numel() == v3.numel() and rec[0]._version == rec[1]):
                    sums = rec[2]                                       # the producing product's sums (csrc/vlinear.hip)
                else:
                    sums = _zeros((_sliced_len(2 * C),), torch.float64, dev)
                    call("svnet_colstats_f64", _p(v3), P, C, 1, _p(sums), _stream())
            else:
                call("svnet_bn_eval_stats_f32", _p(rm2), _p(rv2), C, eps, _p(mean2), _p(invstd2), _stream())
            call("svnet_vtail_fwd_f32", _p(v3), _p(sums), eps, momentum, _p(mean2), _p(invstd2), _p(rm2) if training else None,
                 _p(rv2) if training else None, _p(nbt2) if training else None, _p(g2), _p(b2), _p(gate2), _p(w_eff), B, N, C,
                 _p(out[:, Ca:]), _p(out[:, Ct + Ca:]), 2 * Ct, _p(arg_b), _p(wsb), nbw, 1, _stream())
        nb = L.svnet_pool_workspace_bytes(B, N, Ca, 0) + L.svnet_pool_workspace_bytes(B, N, Ca, 1)
        ws = _zeros((nb,), torch.uint8, dev)
        arg_a = torch.empty((B, Ca), dtype=torch.int32, device=dev)
        call("svnet_bn_pool_fwd_f32", _p(y2), _p(mean1), _p(invstd1), _p(g1), _p(b1), B, N, Ca, act, slope, _p(out), _p(out[:, Ct:]),
             2 * Ct, _p(arg_a), _p(ws), nb, 1, _stream())
        main.wait_stream(side)
        ctx.save_for_backward(y2, mean1, invstd1, g1, b1, arg_a, arg_b, v3, mean2, invstd2, g2, b2, gate2, w_eff, Wzc, sczf)
        if TAP is not None:
            TAP["pools"].append(torch.cat([arg_a, arg_b], dim=1))
            if "acts" in TAP and act in (1, 2):
                TAP["acts"].append((g1.data_ptr(), ((y2 - mean1) * invstd1 * g1 + b1) > 0))
        ctx.meta = (B, N, Ca, C, act, slope, bool(training), v_lin.shape, None if gate is None else gate.shape,
                    None if scz is None else scz.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        y2, mean1, invstd1, g1, b1, arg_a, arg_b, v3, mean2, invstd2, g2, b2, gate2, w_eff, Wzc, sczf = ctx.saved_tensors
        B, N, Ca, C, act, slope, training, vshape, gshape, scshape = ctx.meta
        Cb, P = 3 * C, B * N
        Ct = Ca + Cb
        g = _f32c(g)
        dev = g.device
        F = torch.float32
        dy = torch.empty((B, N, Ca), dtype=F, device=dev) if ctx.needs_input_grad[0] else None
        red1, red2, dgate, gxb = _zeros_pool(dev, ((_sliced_len(2 * Ca),), F), ((_sliced_len(2 * C),), F), ((B, C), F), ((_sliced_len(3 * C),), F))
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            # (allocated ON the side stream: every kernel that touches them runs there - this pass, VectorBN's apply pass, linear2's backward,
            #  which autograd runs where its forward ran - so the allocator may hand their blocks on as soon as they are freed; allocated on the
            #  main stream, dv's block went back to the MAIN stream's pool while linear2's products were still reading it)
            dv = torch.empty((P, 3, C), dtype=F, device=dev)
            recompute = bool(config.FUSE_VTAIL_APPLY)
            g5 = None if recompute else torch.empty((P, 3, C), dtype=F, device=dev)
            call("svnet_vtail_bwd_f32", _p(v3), _p(mean2), _p(invstd2), _p(g2), _p(b2), _p(gate2), _p(w_eff), _p(g[:, Ca:]), _p(g[:, Ct + Ca:]),
                 2 * Ct, _p(arg_b), B, N, C, _p(red2), _p(dgate) if gate2 is not None else None, _p(gxb), _p(g5), _stream())
            if recompute:   # the apply pass recomputes dL/d(VectorBN's output) from the product instead of reading a stored copy (134 MB less)
                call("svnet_vtail_bwd_apply_f32", _p(v3), _p(mean2), _p(invstd2), _p(g2), _p(b2), _p(gate2), _p(w_eff), _p(g[:, Ca:]),
                     _p(g[:, Ct + Ca:]), 2 * Ct, _p(arg_b), B, N, C, _p(red2), int(training), _p(dv), _stream())
            else:
                call("svnet_vbn_bwd_apply_f32", _p(g5), _p(v3), _p(mean2), _p(invstd2), _p(g2), _p(b2), _p(gate2), _p(red2), N, P, C,
                     int(training), _p(dv), _stream())
            if sczf is not None:
                dWz, dscz = _binweight_grad(gxb, Wzc, sczf, 3, C, training, gx_sliced=True)
                dscz = dscz.view(scshape)
            else:
                call("svnet_slices_sum_f32", _p(gxb), 3 * C, _stream())
                dWz, dscz = gxb[:3 * C].view(3, C), None
        call("svnet_bn_pool_bwd_f32", _p(g), _p(g[:, Ct:]), 2 * Ct, _p(arg_a), _p(y2), _p(mean1), _p(invstd1), _p(g1), _p(b1), B, N, Ca,
             act, slope, int(training), _p(red1), _p(dy), _stream())
        # (main waits for the WHOLE vector half here.  Waiting for the gate's gradient only - all the main stream goes on to read - measured
        #  4.219 against 4.227 ms and a captured step's first replay then held a stale weight gradient of linear2: not kept)
        main.wait_stream(side)
        for t in (dWz, dscz):
            if t is not None:
                t.record_stream(main)
        # forward args: y, v_lin, gate, g1, b1, rm1, rv1, g2, b2, rm2, rv2, Wz, scz, training, act, slope, nbt1, nbt2, eps, momentum
        return (dy, dv.view(vshape), dgate.view(gshape) if gate2 is not None else None, red1[Ca:2 * Ca], red1[:Ca], None, None,
                red2[C:2 * C], red2[:C], None, None, dWz, dscz, None, None, None, None, None, None, None)


class Act(torch.autograd.Function):
    """kind 1 relu, 2 sigmoid, 3 leaky-relu(0.2) (sv_layers.py:156-161)."""

    @staticmethod
    def forward(ctx, x, kind):
        _hip(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        call("svnet_act_fwd_f32", _p(x), x.numel(), kind, _p(y), _stream())
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(y)
        call("svnet_act_bwd_f32", _p(g), _p(y), y.numel(), ctx.kind, _p(dx), _stream())
        return dx, None


class GateMLP(torch.autograd.Function):
    """gate = sigmoid(W2 . relu(W0 . pooled)) of an SVBlock (sv_layers.py:156-161,179-183), one launch each way."""

    @staticmethod
    def forward(ctx, pooled, W0, W2):
        pooled, W0c, W2c = _f32c(pooled), _f32c(W0), _f32c(W2)
        B, Cin = pooled.shape
        H, Ov = W0c.shape[0], W2c.shape[0]
        h = torch.empty((B, H), device=pooled.device, dtype=torch.float32)
        gate = torch.empty((B, Ov), device=pooled.device, dtype=torch.float32)
        call("svnet_gate_mlp_fwd_f32", _p(pooled), None, None, 1.0, _p(W0c), _p(W2c), B, Cin, H, Ov, _p(h), _p(gate), None, 0, _stream())
        _tap_act(W0, 2, h)
        ctx.save_for_backward(pooled, W0c, W2c, h, gate)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        pooled, W0, W2, h, gate = ctx.saved_tensors
        dgate = _f32c(dgate)
        B, Cin = pooled.shape
        H, Ov = W0.shape[0], W2.shape[0]
        zb = _zeros((H * Cin + Ov * H,), torch.float32, pooled.device)
        dW0, dW2 = zb[:H * Cin].view(H, Cin), zb[H * Cin:].view(Ov, H)
        dpooled = torch.empty_like(pooled) if ctx.needs_input_grad[0] else None
        call("svnet_gate_mlp_bwd_f32", _p(dgate), _p(gate), _p(h), _p(pooled), 1.0, _p(W0), _p(W2), B, Cin, H, Ov, 1.0,
             _p(dpooled) if dpooled is not None else None, _p(dW0), _p(dW2), _stream())
        return dpooled, dW0, dW2


class GateMLPRows(torch.autograd.Function):
    """gate = sigmoid(W2 . relu(W0 . mean_n s[b,n,:])) of an SVBlock on rows (sv_layers.py:179-183) with the mean over the cloud's rows
    formed INSIDE the MLP's launch (Cin <= 256): the pooling pass in front of it was two launches (split, finish), its backward a
    third - here the gradient of s is handed to autograd as the broadcast view it is."""

    @staticmethod
    def supported(s):
        return s.is_cuda and s.dim() == 3 and s.shape[-1] <= 256 and s.shape[1] * s.shape[2] <= (1 << 18) and s.shape[1] > 0

    @staticmethod
    def forward(ctx, s, W0, W2):
        s, W0c, W2c = _f32c(s), _f32c(W0), _f32c(W2)
        B, Rr, Cin = s.shape
        H, Ov = W0c.shape[0], W2c.shape[0]
        pooled = torch.empty((B, Cin), device=s.device, dtype=torch.float32)
        h = torch.empty((B, H), device=s.device, dtype=torch.float32)
        gate = torch.empty((B, Ov), device=s.device, dtype=torch.float32)
        call("svnet_gate_mlp_fwd_f32", None, None, _p(pooled), 1.0, _p(W0c), _p(W2c), B, Cin, H, Ov, _p(h), _p(gate), _p(s), Rr, _stream())
        _tap_act(W0, 2, h)
        ctx.save_for_backward(pooled, W0c, W2c, h, gate)
        ctx.rows = Rr
        return gate

    @staticmethod
    def backward(ctx, dgate):
        pooled, W0, W2, h, gate = ctx.saved_tensors
        dgate = _f32c(dgate)
        B, Cin = pooled.shape
        H, Ov = W0.shape[0], W2.shape[0]
        zb = _zeros((H * Cin + Ov * H,), torch.float32, pooled.device)
        dW0, dW2 = zb[:H * Cin].view(H, Cin), zb[H * Cin:].view(Ov, H)
        dpooled = torch.empty_like(pooled) if ctx.needs_input_grad[0] else None
        call("svnet_gate_mlp_bwd_f32", _p(dgate), _p(gate), _p(h), _p(pooled), 1.0, _p(W0), _p(W2), B, Cin, H, Ov, 1.0 / float(ctx.rows),
             _p(dpooled) if dpooled is not None else None, _p(dW0), _p(dW2), _stream())
        ds = None if dpooled is None else dpooled.unsqueeze(1).expand(B, ctx.rows, Cin)     # d mean / d s = 1 / R on every row
        return ds, dW0, dW2


class _UnitGrad:
    """One cached scalar 1.0 per device: a train step seeds its backward with it (`torch.autograd.backward(loss, UNIT_GRAD.get(dev))`)
    instead of `loss.backward()`'s fresh ones_like, and SmoothCE.backward recognises it BY IDENTITY - the fill kernel and the
    multiplication by one were two launches (~10 us) on the critical path between the forward and the backward."""

    def __init__(self):
        self.t = {}

    def get(self, dev):
        dev = torch.device(dev)
        key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
        if key not in self.t:
            self.t[key] = torch.ones((), dtype=torch.float32, device=dev)
        return self.t[key]


UNIT_GRAD = _UnitGrad()


class SmoothCE(torch.autograd.Function):
    """Label-smoothed cross entropy, mean over rows (utils.py:33-50 cal_loss)."""

    @staticmethod
    def forward(ctx, logits, target, eps):
        _hip(logits, target)
        lg = _f32c(logits)
        R, C = lg.shape
        tg = target.contiguous().view(-1)
        loss = torch.empty((1,), dtype=torch.float32, device=lg.device)
        dlog = torch.empty_like(lg)
        ws = torch.empty((1024,), dtype=torch.float32, device=lg.device)
        call("svnet_smooth_ce_f32", _p(lg), _p(tg), R, C, eps, _p(loss), _p(dlog), _p(ws), 1024, _stream())
        ctx.save_for_backward(dlog)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        if g is UNIT_GRAD.get(dlog.device):      # the step's own seed (train.TrainStep: a cached 1.0): dL/dlogits is dlog itself
            return dlog, None, None
        return dlog * g, None, None


# ----------------------------------------------------------------------------- fused edge block (tier 2)

def _act_raw(x, kind):
    y = torch.empty_like(x)
    call("svnet_act_fwd_f32", _p(x), x.numel(), kind, _p(y), _stream())
    return y


def _two_sources(g, gc, P, shape_tail):
    """(g pointer tensor or None, second-source tensor or None, its row stride) for a fused layer's backward: `g` is the gradient of the
    layer's own output (contiguous after _f32c, or None when nothing but the concatenation consumed it), `gc` the gradient of its
    CatSink slice - normally a column slice of the concatenation's gradient, read where it lies."""
    n = 1
    for d in shape_tail:
        n *= d
    g = None if g is None else _f32c(g).reshape((P,) + tuple(shape_tail))
    if gc is None:
        return g, None, 0
    if gc.dtype != torch.float32:
        raise TypeError("svnet_amd: expected float32 gradients")
    W = shape_tail[-1]
    ld = gc.stride(-2)
    rows = gc.numel() // W
    ok = gc.stride(-1) == 1 and ld >= W and all(gc.stride(i) == gc.stride(i + 1) * gc.shape[i + 1] for i in range(gc.dim() - 2))
    if not ok:                                   # (an unexpected layout: make it a plain second source)
        gc = gc.contiguous()
        ld = W
    return g, gc, ld


class CatSink:
    """svcat([x1, .., xn]) of a pyramid of pooled levels (sv_dgcnn_cls.py:68, sv_dgcnn_partseg.py) WITHOUT a concatenation pass: while the
    sink is active, the apply kernel of every fused edge layer writes its pooled (s, v) a second time, as the next column slice of
    the pre-allocated concatenations.  `result(levels)` returns the concatenation as an autograd function of the levels (backward =
    the column slices of torch.cat's backward); levels that did not come from a fused layer - or a sink that was not filled in
    order - fall back to torch.cat.

        sink = CatSink(widths_s, widths_v)
        with sink:
            ... svpool(block(...)) for every level, in order ...
        s_cat, v_cat = sink.result(pyramid)
    """

    def __init__(self, widths_s, widths_v):
        self.ws, self.wv = [int(w) for w in widths_s], [int(w) for w in widths_v]
        self.s = self.v = None
        self.filled = []                  # (s_out, v_out) addresses of the slots written so far, in order
        self.views = []                   # the slices' autograd-tracked views (outputs of the fused layers), in order
        self.prev = None

    def __enter__(self):
        global _SINK
        self.prev, _SINK = _SINK, self
        return self

    def __exit__(self, *exc):
        global _SINK
        _SINK = self.prev

    def slot(self, B, N, Os, Ov, dev):
        """(s_ptr, s_ld, v_ptr, v_ld) of the next slice if it is (Os, Ov) wide, else None (and the sink is abandoned)."""
        if self.filled is None:           # abandoned by an earlier level: every later one keeps its own output, result() concatenates
            return None
        i = len(self.filled)
        if i >= len(self.ws) or (self.ws[i], self.wv[i]) != (Os, Ov):
            self.filled = None
            return None
        if self.s is None:
            self.s = torch.empty((B, N, sum(self.ws)), dtype=torch.float32, device=dev)
            self.v = torch.empty((B, N, 3, sum(self.wv)), dtype=torch.float32, device=dev)
        if tuple(self.s.shape[:2]) != (B, N):
            self.filled = None
            return None
        so, vo = sum(self.ws[:i]), sum(self.wv[:i])
        return (ctypes.c_void_p(self.s.data_ptr() + 4 * so), sum(self.ws), ctypes.c_void_p(self.v.data_ptr() + 4 * vo), sum(self.wv))

    def wrote(self, s_out, v_out):
        """Called inside the layer's forward; returns the (s, v) column-slice views of the concatenations it has just filled."""
        if self.filled is None:
            return None, None
        i = len(self.filled)
        self.filled.append((s_out.data_ptr(), v_out.data_ptr()))
        so, vo = sum(self.ws[:i]), sum(self.wv[:i])
        return self.s[..., so:so + self.ws[i]], self.v[..., vo:vo + self.wv[i]]

    def tracked(self, s_view, v_view):
        """Called by the layer's caller with the views as autograd returned them (outputs of the layer's Function)."""
        if s_view is not None:
            self.views.append((s_view, v_view))

    def result(self, levels):
        ok = (self.filled is not None and len(self.filled) == len(self.ws) == len(levels) == len(self.views)
              and all(torch.is_tensor(b[0]) and torch.is_tensor(b[1]) and a[0] == b[0].data_ptr() and a[1] == b[1].data_ptr()
                      for a, b in zip(self.filled, levels)))
        if not ok:
            return torch.cat([x[0] for x in levels], dim=-1), torch.cat([x[1] for x in levels], dim=-1)
        # the concatenation depends on the layers through their SLICE outputs: its gradient reaches every layer's backward as a second
        # source next to the gradient of the pooled output the next layer consumed - no strided add / copy kernels in between
        flat = [t for vw in self.views for t in vw]
        return _CatFilled.apply(self.s, self.v, len(levels), *flat)


_SINK = None


class _CatFilled(torch.autograd.Function):
    """The concatenations a CatSink's kernels have already filled, as a function of the layers' slice outputs (autograd only: no kernel)."""

    @staticmethod
    def forward(ctx, s_cat, v_cat, n, *flat):
        ctx.ws = [flat[2 * i].shape[-1] for i in range(n)]
        ctx.wv = [flat[2 * i + 1].shape[-1] for i in range(n)]
        return s_cat.view(s_cat.shape), v_cat.view(v_cat.shape)

    @staticmethod
    def backward(ctx, gs, gv):
        out, so, vo = [], 0, 0
        for ws, wv in zip(ctx.ws, ctx.wv):
            out.append(gs[..., so:so + ws] if gs is not None else None)
            out.append(gv[..., vo:vo + wv] if gv is not None else None)
            so, vo = so + ws, vo + wv
        return (None, None, None) + tuple(out)


class EdgeBlock(torch.autograd.Function):
    """get_graph_feature_sv -> binarized SVBlock -> svpool(max, mean) in one pass over the edges
    (csrc/edgeblock.hip).  Inputs are the POINT tables; no edge tensor is materialised in either direction."""

    @staticmethod
    def forward(ctx, s, v, idx, k, training, Wz, scz, W1, beta1, scale1, g1, b1, rm1, rv1, W2, sc2, g2, b2, rm2, rv2, Wg0, Wg2,
                nbt1=None, nbt2=None):
        _hip(s, v)
        from ._lib import EdgeBlockDesc
        s = _f32c(s)
        v = _f32c(v)
        B, N, Cs = s.shape
        Cv = v.shape[-1]
        Os, Ov = W1.shape[0], W2.shape[0]
        P, E = B * N, B * N * k
        dev = s.device
        f32 = dict(dtype=torch.float32, device=dev)
        # `idx` is the neighbour-id tensor or the lazy edge handle (sv_util.EdgeFeatures): with a handle whose graph is not
        # computed yet, the k-NN kernels are launched HERE on the main stream, after the fork point of the side stream that takes
        # the point-level GEMMs below (they need v and the weights, not the graph)
        W1c, W2c, Wzc, Wg0c, Wg2c = _f32c(W1), _f32c(W2), _f32c(Wz), _f32c(Wg0), _f32c(Wg2)
        sc1, sc2f, sczf = _f32c(scale1).reshape(-1), _f32c(sc2).reshape(-1), _f32c(scz).reshape(-1)

        # per-point pieces of the two linear maps on v_e = [v_j - v_i, v_i]: ut = [U | T], zz = [Zp | Zq]
        import weakref
        refs = tuple(weakref.ref(p_) for p_ in (W1, beta1, W2, sc2, Wz, scz))

        def build():
            out = {"wv": torch.empty((2 * Ov + 6, Cv), **f32), "scv": torch.empty((2 * Ov + 6,), **f32),
                   "w_sign": torch.empty((Os, 5), dtype=torch.int64, device=dev), "w_nz": torch.empty((Os, 5), dtype=torch.int64, device=dev),
                   "beta_perm": torch.empty((5 * 64,), **f32), "wbt": torch.empty((320 * ((Os + 15) // 16 * 16),), dtype=torch.int16, device=dev),
                   "w_dense": torch.zeros((1,), dtype=torch.int32, device=dev)}       # 1 = no exact zero in W1 (svnet_edgeblock_wbt_bf16 sets it from the planes)

            def rebuild():                       # (holds the parameters weakly: a dead model's entry is dropped by _PlaneCache._stale())
                ps = [r() for r in refs]
                if any(p_ is None for p_ in ps):
                    return
                W1p, beta1p, W2p, sc2p, Wzp, sczp = ps
                call("svnet_edgeblock_prepare_vec_f32", _p(_f32c(W2p.detach())), _p(_f32c(sc2p.detach()).reshape(-1)), _p(_f32c(Wzp.detach())),
                     _p(_f32c(sczp.detach()).reshape(-1)), Ov, Cv, _p(out["wv"]), _p(out["scv"]), _stream())
                call("svnet_edgeblock_prepare_f32", _p(_f32c(W1p.detach())), _p(_f32c(beta1p.detach())), Os, Cs, Cv, _p(out["w_sign"]), _p(out["w_nz"]),
                     _p(out["beta_perm"]), _stream())
                call("svnet_edgeblock_wbt_bf16", _p(out["w_sign"]), _p(out["w_nz"]), Os, Cs, Cv, _p(out["wbt"]), _p(out["w_dense"]), _stream())
            rebuild()
            return out, rebuild
        packed = PLANES.get("edge", (W1, beta1, W2, sc2, Wz, scz), build)
        wv, scv, w_sign, w_nz, beta_perm, wbt = (packed[n] for n in ("wv", "scv", "w_sign", "w_nz", "beta_perm", "wbt"))
        # fork AFTER the packed weights exist on this stream (they may just have been built on it) and BEFORE the k-NN is launched.
        # The side stream's outputs are allocated BEFORE the fork: a block that the main stream frees after the fork (the k-NN's
        # workspace) may be handed out again at once, and the side stream - which does not wait for the k-NN - would write into it
        zz = torch.empty((P * 3, 6), **f32)
        ut = torch.empty((P * 3, 2 * Ov), **f32)
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)
        idx = (idx if torch.is_tensor(idx) else idx.idx).contiguous()
        _hip(idx)
        with torch.cuda.stream(side):
            gemm(3 * P, 6, Cv, A=v, a_rs=Cv, a_cs=1, B=wv[2 * Ov:], b_rs=1, b_cs=Cv, b_exact=True, C=zz, ldc=6, col_scale=scv[2 * Ov:])
            gemm(3 * P, 2 * Ov, Cv, A=v, a_rs=Cv, a_cs=1, B=wv, b_rs=1, b_cs=Cv, b_exact=True, C=ut, ldc=2 * Ov, col_scale=scv)
        main.wait_stream(side)

        n_max = torch.empty((P, Os), dtype=torch.int32, device=dev)
        n_min = torch.empty((P, Os), dtype=torch.int32, device=dev)
        slot_max = torch.empty((P, Os), dtype=torch.uint8, device=dev)
        slot_min = torch.empty((P, Os), dtype=torch.uint8, device=dev)
        mv = torch.empty((P, 3, Ov), **f32)
        mvn = torch.empty((P, 3, Ov), **f32)
        if training:
            stat_n, stat_v, gate_sum = _zeros_pool(dev, ((RED_SLICES * 2 * Os,), torch.int64), ((RED_SLICES * 2 * Ov,), torch.float64),
                                                     ((B, 2 * Cs), torch.float64))      # batch sums in slices (SVNET_RED_SLICES)
        else:
            stat_n = stat_v = None
            gate_sum = _zeros((B, 2 * Cs), torch.float64, dev)
        d = EdgeBlockDesc()
        d.B, d.N, d.k = B, N, k
        d.Cs, d.Cv, d.Os, d.Ov = Cs, Cv, Os, Ov
        d.s, d.v, d.idx, d.zz, d.ut = _p(s), _p(v), _p(idx), _p(zz), _p(ut)
        d.w_sign, d.w_nz, d.beta_perm = _p(w_sign), _p(w_nz), _p(beta_perm)
        d.w_dense = _p(packed["w_dense"]) if config.EDGE_DENSE_WEIGHTS else None
        d.n_max, d.n_min, d.slot_max, d.slot_min = _p(n_max), _p(n_min), _p(slot_max), _p(slot_min)
        d.mv, d.mvn, d.stat_n, d.stat_v, d.gate_sum = _p(mv), _p(mvn), _p(stat_n), _p(stat_v), _p(gate_sum)
        # kept for the backward instead of any fp32 edge tensor: the integer sum n (2 B per edge-channel) and the sign /
        # non-zero / STE bit planes of the binarized edge feature (120 B per edge)
        keep = (training and any(ctx.needs_input_grad)) or TAP is not None
        n16 = torch.empty((E, Os), dtype=torch.int16, device=dev) if keep else None
        planes = torch.empty((E, 15), dtype=torch.int64, device=dev) if keep else None
        d.n16, d.planes = _p(n16), _p(planes)
        call("svnet_edgeblock_fwd_f32", ctypes.byref(d), _stream())
        if TAP is not None:
            TAP["signs"].append(("edges", E, (Cs, Cv), planes))

        # gate MLP on the mean edge scalar (sv_layers.py:156-161,179-183): one workgroup per cloud
        H = Wg0.shape[0]
        h = torch.empty((B, H), **f32)
        gate = torch.empty((B, Ov), **f32)
        gin = torch.empty((B, 2 * Cs), **f32)
        # (the MLP runs in extra workgroups of the coefficient launch below: the two only share their inputs)
        job = _lib.GateFwdJob(None, _p(gate_sum), _p(gin), 1.0 / float(N * k), _p(Wg0c), _p(Wg2c), B, 2 * Cs, H, Ov, _p(h), _p(gate))

        coef = torch.empty((4 * Os + 4 * Ov,), **f32)
        s_out = torch.empty((B, N, Os), **f32)
        v_out = torch.empty((B, N, 3, Ov), **f32)
        slot = _SINK.slot(B, N, Os, Ov, dev) if _SINK is not None else None
        kws = knn_table_ahead.workspace(B, N, Os, Ov, dev)
        if _block_tail("svnet_edgeblock_tail_f32", P, N, E, Os, Ov, stat_n, stat_v, sc1, g1, b1, rm1, rv1, g2, b2, rm2, rv2, training, coef,
                       nbt1, nbt2, job, n_max, n_min, mv, mvn, s_out, v_out, slot, kws):
            pass        # coefficients + gate MLP + apply (+ the next k-NN's table) in one launch
        else:
            call("svnet_edgeblock_coeffs_f32", _p(stat_n), _p(stat_v), E, Os, Ov, _p(sc1), _p(g1), _p(b1), _p(rm1), _p(rv1),
                 _p(g2), _p(b2), _p(rm2), _p(rv2), int(training), BN_EPS, BN_MOMENTUM, _p(coef), _p(nbt1), _p(nbt2), ctypes.byref(job), _stream())
            if kws is not None:
                call("svnet_edgeblock_apply_knn_f32", _p(n_max), _p(n_min), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out),
                     _p(v_out), *(slot if slot is not None else (None, 0, None, 0)), _p(kws), kws.numel(), _stream())
            else:
                call("svnet_edgeblock_apply_f32", _p(n_max), _p(n_min), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out),
                     _p(v_out), *(slot if slot is not None else (None, 0, None, 0)), _stream())
        if kws is not None:
            knn_table_ahead.table = (s_out.data_ptr(), v_out.data_ptr(), kws, B, N, Os + 3 * Ov)
        _tap_act(Wg0, 2, h)
        s_view, v_view = _SINK.wrote(s_out, v_out) if slot is not None else (None, None)
        if TAP is not None:      # the pooled slot: BatchNorm + LeakyReLU is increasing (slope coef[o] >= 0: max_k n) or decreasing (min_k n)
            TAP["pools"].append(torch.where(coef[:Os].view(1, Os) >= 0, slot_max, slot_min))
            # ... and the kink decision of bn1 + LeakyReLU on EVERY edge (sv_layers.py:189-190): the layer's pre-activation is
            # A1[o] * n + B1[o] with the kept integer n - the expression the apply kernel evaluates at the pooled edge
            if "acts" in TAP and n16 is not None:
                TAP["acts"].append((g1.data_ptr(), (coef[:Os].view(1, Os) * n16.float() + coef[Os:2 * Os].view(1, Os)) > 0))
        ctx.save_for_backward(v, idx, zz, ut, w_sign, w_nz, n16, planes, n_max, n_min, slot_max, slot_min, mv, mvn, coef, gate, h,
                              gin, wv, scv, W1c, sc1, W2c, sc2f, Wzc, sczf, g1, g2, Wg0c, Wg2c, wbt)
        ctx.meta = (B, N, k, Cs, Cv, Os, Ov, bool(training), scale1.shape, sc2.shape, scz.shape)
        ctx.set_materialize_grads(False)        # (an output nothing consumed arrives as None, not as a zero tensor + fill launch)
        return s_out, v_out, s_view, v_view

    @staticmethod
    def backward(ctx, gs, gv, gsc=None, gvc=None):
        from ._lib import EdgeBlockBwdDesc
        (v, idx, zz, ut, w_sign, w_nz, n16, planes, n_max, n_min, slot_max, slot_min, mv, mvn, coef, gate, h, gin, wv, scv,
         W1, sc1, W2, sc2, Wz, scz, g1, g2, Wg0, Wg2, wbt) = ctx.saved_tensors
        B, N, k, Cs, Cv, Os, Ov, training, sh1, sh2, shz = ctx.meta
        P, E = B * N, B * N * k
        dev = v.device
        if n16 is None:
            raise RuntimeError("EdgeBlock.backward: the forward ran in eval mode (no STE gradient exists, sv_layers.py:38-45)")
        f32 = dict(dtype=torch.float32, device=dev)
        F = torch.float32
        gs, gs2, gs2_ld = _two_sources(gs, gsc, P, (Os,))
        gv, gv2, gv2_ld = _two_sources(gv, gvc, P, (3, Ov))
        if gs is None and gs2 is None:
            gs = torch.zeros((P, Os), **f32)
        if gv is None and gv2 is None:
            gv = torch.zeros((P, 3, Ov), **f32)
        gv_sum = torch.empty((P, 3, Ov), **f32) if gv2 is not None else None
        H = Wg0.shape[0]
        K1, R = 2 * Cs + 6 * Cv, 2 * Ov + 6

        # every accumulator of this backward from ONE zero fill
        (red, redv, dgate, dWg0, dWg2, ds_acc, dv_acc, dzc, dbeta_perm, GXp, GXc, ovf_count) = _zeros_pool(
            dev, ((RED_SLICES * 2 * Os,), F), ((RED_SLICES * 2 * Ov,), F), ((B, Ov), F), ((H, 2 * Cs), F), ((Ov, H), F), ((P, Cs), F), ((P, 3, Cv), F),
            ((P, 3, 3), F), ((64, 320), F), ((Os, 320), F), ((R, Cv), F), ((1,), torch.int32))        # dbeta_perm: SVNET_DBETA_SLICES x 320

        # Two streams (forked / joined with events, so the pattern is captured into the hipGraph as parallel branches): the side
        # stream builds the reverse neighbour lists while the main stream runs the point-level prelude
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        rev_range = torch.empty((2 * P,), dtype=torch.int32, device=dev)
        rev_edge = torch.empty((E,), dtype=torch.int32, device=dev)
        rev_src = torch.empty((E,), dtype=torch.int32, device=dev)
        # reverse lists longer than GATHER_CHUNK entries are summed in pieces (feature-space graphs have hubs: a wave per whole list
        # made the gather as slow as its longest list)
        # (the list lengths scale with k - their mean IS k: k = 40, part-seg, 15.86 -> 15.81 ms with 256 entries per wave, profiles/r05_ab_partseg_switches.log)
        GATHER_CHUNK = int(config.GATHER_CHUNK) * max(1, (k + 19) // 20)
        ovf_items = torch.empty((2 * (2 * E // GATHER_CHUNK + 1),), dtype=torch.int32, device=dev) if GATHER_CHUNK > 0 else None
        if ovf_items is None:
            ovf_count = None
        side.wait_stream(main)
        # (the reverse lists are built on the side stream BEHIND the vector path, see there: the message sums are their only reader)

        # ---- point-level prelude: BatchNorm reductions, gate gradient
        gy = torch.empty((P, Os), **f32)
        call("svnet_edgeblock_bwd_prelude_f32", _p(gs), _p(gv), _p(n_max), _p(n_min), _p(mv), _p(mvn), _p(coef), _p(sc1), _p(gate),
             P, N, Os, Ov, 0.2, _p(gy), _p(red), _p(redv), _p(dgate), _p(gs2), gs2_ld, _p(gv2), gv2_ld, _p(gv_sum), _stream())
        if gv_sum is not None:
            gv = gv_sum                          # (what the kernels below read: the summed vector gradient)
        bcoef = torch.empty((8 * Os + 2 * Ov + 4,), **f32)
        dg1, db1 = torch.empty((Os,), **f32), torch.empty((Os,), **f32)
        dg2, db2 = torch.empty((Ov,), **f32), torch.empty((Ov,), **f32)
        # ---- gate MLP backward (dW0, dW2 and the per-edge constant of the gate path, workgroups per cloud) in extra workgroups of the
        # coefficient launch: both start from the prelude's sums only
        gconst = torch.empty((B, 2 * Cs), **f32)
        inv_nk = 1.0 / float(N * k)
        job = _lib.GateBwdJob(_p(dgate), _p(gate), _p(h), _p(gin), inv_nk, _p(Wg0), _p(Wg2), B, 2 * Cs, H, Ov, inv_nk, _p(gconst), _p(dWg0), _p(dWg2))
        call("svnet_edgeblock_bwd_coeffs_f32", _p(red), _p(redv), _p(coef), _p(g1), _p(g2), E, Os, Ov, int(training), _p(sc1), _p(bcoef),
             _p(dg1), _p(db1), _p(dg2), _p(db2), ctypes.byref(job), _stream())
        coeffs_done = main.record_event() if config.VEC_EARLY else None      # (what the vector path waits for)

        # ---- the edge pass
        affine = k >= 8        # the weight-gradient GEMM recomputes dL/dy_pre from n16: the tile kernel then writes no fp32 [E,Os] tensor
        dn_out = None if affine else torch.empty((E, Os), **f32)
        x_sign = torch.empty(((E + 63) // 64, 320), dtype=torch.int64, device=dev)
        x_nz = torch.empty(((E + 63) // 64, 320), dtype=torch.int64, device=dev)
        d = EdgeBlockBwdDesc()
        d.B, d.N, d.k = B, N, k
        d.Cs, d.Cv, d.Os, d.Ov = Cs, Cv, Os, Ov
        d.v, d.idx, d.zz, d.ut = _p(v), _p(idx), _p(zz), _p(ut)
        d.n16, d.planes, d.w1bt, d.scale1 = _p(n16), _p(planes), _p(wbt), _p(sc1)
        d.slot_max, d.slot_min, d.coef, d.gate = _p(slot_max), _p(slot_min), _p(coef), _p(gate)
        d.gy, d.bcoef, d.gv, d.gconst = _p(gy), _p(bcoef), _p(gv), _p(gconst)
        d.dn_out, d.x_sign32, d.x_nz32 = _p(dn_out), _p(x_sign), _p(x_nz)
        # the neighbour's share of every edge goes to a message row and is summed over the reverse neighbour lists (gather):
        # float atomics are executed at the memory side on this part and were the bottleneck of the scatter formulation
        msg = torch.empty((E, _lib.lib().svnet_edgeblock_msg_stride(Cs, Cv, Ov)), **f32)
        dvc = torch.empty((P, 3, Ov), **f32)
        ub_tab, ge_tab = torch.empty((P, 3, Ov), **f32), torch.empty((P, 3, Ov), **f32)
        d.ub_tab, d.ge_tab = _p(ub_tab), _p(ge_tab)
        d.msg, d.ds_acc, d.dv_acc, d.dvc, d.dzc, d.dbeta_perm = _p(msg), _p(ds_acc), _p(dv_acc), _p(dvc), _p(dzc), _p(dbeta_perm)
        d.debug = _p(DEBUG_BUFFER)
        # the vector path (wave per point) and the scalar path (32-edge tiles) are independent: two streams, so that the
        # register/LDS-bound tile kernel and the light vector kernel share the CUs
        if coeffs_done is not None:
            side.wait_event(coeffs_done)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            d.parts = 1
            call("svnet_edgeblock_bwd_f32", ctypes.byref(d), _stream())
            # the reverse neighbour lists (one 1024-thread workgroup per cloud, 60 - 70 us) used to be issued first thing, beside the
            # prelude: issued here - behind the vector path, under the tile kernel - the step is 1.4 % shorter (4.66 -> 4.60 ms, six
            # alternating runs): the prelude and the vector path, which the critical path waits for, no longer share the CUs with them
            call("svnet_knn_reverse_i32", _p(idx), B, N, k, _p(rev_range), _p(rev_edge), _p(rev_src), GATHER_CHUNK, _p(ovf_items), _p(ovf_count),
                 _stream())
            vec_done = side.record_event() if DEFERRED.active else None     # (also covers the reverse lists: same stream; unused on the joined schedule)
        d.parts = 2
        call("svnet_edgeblock_bwd_f32", ctypes.byref(d), _stream())

        # ---- neighbour sums -> gradient rows of the collapsed per-point products [U | T | Zp | Zq], ds, dv; dbeta in the
        # reference's feature order
        Rp = (R + 3) // 4 * 4                    # padded row stride: 16-byte fragment loads in the two GEMMs below
        acat = torch.empty((3 * P, Rp), **f32)
        dbeta1 = torch.empty((1, K1), **f32)
        # linear1's weight-gradient product GXp = dy^T . x_b (MFMA, ternary planes, fused column order) only needs the tile
        # kernel's outputs: it keeps the main stream while the side stream (joined with main first) sums the messages
        side.wait_stream(main)
        used = 0                      # 32-column tiles of the 5 x 64 fused columns that hold features (the rest is padding)
        for ct in range(10):
            if ((Cs if ct < 4 else 2 * Cv) > 32 * (ct & 1)):
                used |= 1 << ct

        def wgrad1():
            if affine:
                chc_off = (3 * Os + 2 * Ov + 3) & ~3
                call("svnet_edgeblock_wgrad_f32", _p(n16), _p(slot_max), _p(slot_min), _p(gy), _p(bcoef[chc_off:]), _p(x_sign), _p(x_nz), E, k, Os,
                     _p(GXp), used, _stream())
            else:
                gemm(320, Os, E, a_planes=(x_sign, x_nz), B=dn_out, b_rs=Os, b_cs=1, C=GXp, ldc=1, c_cs=320, accumulate=True,
                     tern_tile_mask=used)

        if DEFERRED.active and DEFERRED.first_use(W1, W2, Wz, Wg0, Wg2, g1, g2):
            # the step gathers the parameter gradients once, after the whole backward: the weight-gradient chain stays on the side
            # stream, unjoined, and the main stream carries what the NEXT layer's backward waits for (message sums -> dv product)
            with torch.cuda.stream(side):
                wgrad1()
            main.wait_event(vec_done)
            call("svnet_edgeblock_bwd_gather_f32", _p(msg), _p(rev_range), _p(rev_edge), _p(rev_src), _p(ut), _p(ub_tab), _p(ge_tab),
                 _p(coef), _p(bcoef), Os, _p(dvc), _p(dzc), P, N, Cs, Cv, Ov, _p(acat), Rp, _p(ds_acc), _p(dv_acc), _p(dbeta_perm),
                 _p(dbeta1), GATHER_CHUNK, _p(ovf_items), _p(ovf_count), _stream())
            gathered = main.record_event()
            gemm(3 * P, Cv, R, A=acat, a_rs=Rp, a_cs=1, a_scale=scv, B=wv, b_rs=Cv, b_cs=1, b_exact=True, C=dv_acc, ldc=Cv, accumulate=True)
            with torch.cuda.stream(side):
                side.wait_event(gathered)
                gemm(R, Cv, 3 * P, A=acat, a_rs=1, a_cs=Rp, B=v, b_rs=Cv, b_cs=1, C=GXc, ldc=Cv, accumulate=True)
                dW1, dW2, dWz = torch.empty((Os, K1), **f32), torch.empty((Ov, 2 * Cv), **f32), torch.empty((3, 2 * Cv), **f32)
                dsc1, dsc2, dscz = torch.empty((Os,), **f32), torch.empty((Ov,), **f32), torch.empty((3,), **f32)
                call("svnet_edgeblock_bwd_params_f32", _p(GXp), _p(GXc), _p(W1), _p(sc1), _p(W2), _p(sc2), _p(Wz), _p(scz), Os, Ov, Cs, Cv,
                     _p(dW1), _p(dsc1), _p(dW2), _p(dsc2), _p(dWz), _p(dscz), _stream())
            # (NOT the returned gradients: autograd keeps a returned tensor as the parameter's .grad only while nobody else holds it -
            #  with a second reference it clones it, on the main stream, before the side stream has written it; .grad keeps them alive)
            DEFERRED.keep.append((n16, slot_max, slot_min, gy, bcoef, x_sign, x_nz, dn_out, GXp, GXc, acat, v, W1, sc1, W2, sc2, Wz, scz))
            return (ds_acc.view(B, N, Cs), dv_acc.view(B, N, 3, Cv), None, None, None, dWz, dscz.view(shz), dW1, dbeta1,
                    dsc1.view(sh1), dg1, db1, None, None, dW2, dsc2.view(sh2), dg2, db2, None, None, dWg0, dWg2, None, None)
        wgrad1()
        with torch.cuda.stream(side):
            call("svnet_edgeblock_bwd_gather_f32", _p(msg), _p(rev_range), _p(rev_edge), _p(rev_src), _p(ut), _p(ub_tab), _p(ge_tab),
                 _p(coef), _p(bcoef), Os, _p(dvc), _p(dzc), P, N, Cs, Cv, Ov, _p(acat), Rp, _p(ds_acc), _p(dv_acc), _p(dbeta_perm),
                 _p(dbeta1), GATHER_CHUNK, _p(ovf_items), _p(ovf_count), _stream())
            gathered = side.record_event()
            # linear2 and the v2s frame: dv += (acat * scv) . wv  (what the next layer waits for: stays behind the gather)
            gemm(3 * P, Cv, R, A=acat, a_rs=Rp, a_cs=1, a_scale=scv, B=wv, b_rs=Cv, b_cs=1, b_exact=True, C=dv_acc, ldc=Cv, accumulate=True)
        # ... and their weight gradients GXc = acat^T . v, which nothing downstream waits for: behind linear1's on the main stream
        main.wait_event(gathered)
        gemm(R, Cv, 3 * P, A=acat, a_rs=1, a_cs=Rp, B=v, b_rs=Cv, b_cs=1, C=GXc, ldc=Cv, accumulate=True)
        main.wait_stream(side)
        dW1, dW2, dWz = torch.empty((Os, K1), **f32), torch.empty((Ov, 2 * Cv), **f32), torch.empty((3, 2 * Cv), **f32)
        dsc1, dsc2, dscz = torch.empty((Os,), **f32), torch.empty((Ov,), **f32), torch.empty((3,), **f32)
        call("svnet_edgeblock_bwd_params_f32", _p(GXp), _p(GXc), _p(W1), _p(sc1), _p(W2), _p(sc2), _p(Wz), _p(scz), Os, Ov, Cs, Cv,
             _p(dW1), _p(dsc1), _p(dW2), _p(dsc2), _p(dWz), _p(dscz), _stream())

        # forward args: s, v, idx, k, training, Wz, scz, W1, beta1, scale1, g1, b1, rm1, rv1, W2, sc2, g2, b2, rm2, rv2, Wg0, Wg2, nbt1, nbt2
        return (ds_acc.view(B, N, Cs), dv_acc.view(B, N, 3, Cv), None, None, None, dWz, dscz.view(shz), dW1, dbeta1,
                dsc1.view(sh1), dg1, db1, None, None, dW2, dsc2.view(sh2), dg2, db2, None, None, dWg0, dWg2, None, None)


class XyzBlock(torch.autograd.Function):
    """get_graph_feature[_cross] -> init_scalar (Vector2Scalar) -> SVBlock (fp) -> svpool(max, mean) of the FIRST edge layer in one
    pass over the edges (csrc/xyzblock.hip): 2 vector channels [x_j - x_i | x_i] (the DGCNN callers' conv1) or 3 with the cross
    product (the PointNet callers' conv_pos).  The coordinates receive no gradient."""

    @staticmethod
    def forward(ctx, x, idx, k, training, W0, Wz, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, Wg0, Wg2, nbt1=None, nbt2=None):
        _hip(x, idx)
        from ._lib import XyzBlockDesc
        x = _f32c(x.detach())
        B, _, N = x.shape
        Os, Ov = W1.shape[0], W2.shape[0]
        NC = W2.shape[1]                 # vector channels of the edge feature: 2 (get_graph_feature) or 3 (get_graph_feature_cross)
        NG = 3 * NC
        P, E = B * N, B * N * k
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        idx = idx.contiguous()
        y_max, y_min = torch.empty((P, Os), **f32), torch.empty((P, Os), **f32)
        slot_max = torch.empty((P, Os), dtype=torch.uint8, device=dev)
        slot_min = torch.empty((P, Os), dtype=torch.uint8, device=dev)
        mv, mvn = torch.empty((P, 3, Ov), **f32), torch.empty((P, 3, Ov), **f32)
        if training:
            stat_y, stat_v, gate_sum = _zeros_pool(dev, ((RED_SLICES * 2 * Os,), torch.float64), ((RED_SLICES * 2 * Ov,), torch.float64),
                                                     ((B, NG), torch.float64))
        else:
            stat_y = stat_v = None
            gate_sum = _zeros((B, NG), torch.float64, dev)
        W0c, Wzc, W1c, W2c = _f32c(W0), _f32c(Wz), _f32c(W1), _f32c(W2)
        d = XyzBlockDesc()
        d.B, d.N, d.k, d.Os, d.Ov = B, N, k, Os, Ov
        d.x, d.idx, d.w0, d.wz, d.w1, d.w2 = _p(x), _p(idx), _p(W0c), _p(Wzc), _p(W1c), _p(W2c)
        d.y_max, d.y_min, d.slot_max, d.slot_min, d.mv, d.mvn = _p(y_max), _p(y_min), _p(slot_max), _p(slot_min), _p(mv), _p(mvn)
        d.stat_y, d.stat_v, d.gate_sum = _p(stat_y), _p(stat_v), _p(gate_sum)
        d.nc = NC
        call("svnet_xyzblock_fwd_f32", ctypes.byref(d), _stream())

        H = Wg0.shape[0]
        h = torch.empty((B, H), **f32)
        gate = torch.empty((B, Ov), **f32)
        gin = torch.empty((B, NG), **f32)
        Wg0c, Wg2c = _f32c(Wg0), _f32c(Wg2)
        job = _lib.GateFwdJob(None, _p(gate_sum), _p(gin), 1.0 / float(N * k), _p(Wg0c), _p(Wg2c), B, NG, H, Ov, _p(h), _p(gate))   # (beside the coefficients)

        coef = torch.empty((4 * Os + 4 * Ov,), **f32)
        s_out = torch.empty((B, N, Os), **f32)
        v_out = torch.empty((B, N, 3, Ov), **f32)
        slot = _SINK.slot(B, N, Os, Ov, dev) if _SINK is not None else None
        kws = knn_table_ahead.workspace(B, N, Os, Ov, dev)
        if _block_tail("svnet_xyzblock_tail_f32", P, N, E, Os, Ov, stat_y, stat_v, None, g1, b1, rm1, rv1, g2, b2, rm2, rv2, training, coef,
                       nbt1, nbt2, job, y_max, y_min, mv, mvn, s_out, v_out, slot, kws):
            pass
        else:
            call("svnet_xyzblock_coeffs_f32", _p(stat_y), _p(stat_v), E, Os, Ov, _p(g1), _p(b1), _p(rm1), _p(rv1), _p(g2), _p(b2), _p(rm2),
                 _p(rv2), int(training), BN_EPS, BN_MOMENTUM, _p(coef), _p(nbt1), _p(nbt2), ctypes.byref(job), _stream())
            if kws is not None:
                call("svnet_xyzblock_apply_knn_f32", _p(y_max), _p(y_min), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out),
                     _p(v_out), *(slot if slot is not None else (None, 0, None, 0)), _p(kws), kws.numel(), _stream())
            else:
                call("svnet_xyzblock_apply_f32", _p(y_max), _p(y_min), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2, _p(s_out), _p(v_out),
                     *(slot if slot is not None else (None, 0, None, 0)), _stream())
        if kws is not None:
            knn_table_ahead.table = (s_out.data_ptr(), v_out.data_ptr(), kws, B, N, Os + 3 * Ov)
        _tap_act(Wg0, 2, h)
        s_view, v_view = _SINK.wrote(s_out, v_out) if slot is not None else (None, None)
        if TAP is not None:
            TAP["pools"].append(torch.where(coef[:Os].view(1, Os) >= 0, slot_max, slot_min))
        ctx.save_for_backward(x, idx, W0c, Wzc, W1c, W2c, y_max, y_min, slot_max, slot_min, mv, mvn, coef, gate, h, gin, g1, g2, Wg0, Wg2)
        ctx.meta = (B, N, k, Os, Ov, bool(training), NC)
        ctx.set_materialize_grads(False)
        return s_out, v_out, s_view, v_view

    @staticmethod
    def backward(ctx, gs, gv, gsc=None, gvc=None):
        from ._lib import XyzBlockBwdDesc
        (x, idx, W0, Wz, W1, W2, y_max, y_min, slot_max, slot_min, mv, mvn, coef, gate, h, gin, g1, g2, Wg0, Wg2) = ctx.saved_tensors
        B, N, k, Os, Ov, training, NC = ctx.meta
        NG, NF = 3 * NC, 6 * NC
        P, E = B * N, B * N * k
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        gs, gs2, gs2_ld = _two_sources(gs, gsc, P, (Os,))
        gv, gv2, gv2_ld = _two_sources(gv, gvc, P, (3, Ov))
        if gs is None and gs2 is None:
            gs = torch.zeros((P, Os), **f32)
        if gv is None and gv2 is None:
            gv = torch.zeros((P, 3, Ov), **f32)
        gv_sum = torch.empty((P, 3, Ov), **f32) if gv2 is not None else None
        gy = torch.empty((P, Os), **f32)
        H = Wg0.shape[0]
        F = torch.float32
        GW = Os * NF + Ov * NC + NF
        red, redv, dgate, dWg0, dWg2, gw = _zeros_pool(dev, ((RED_SLICES * 2 * Os,), F), ((RED_SLICES * 2 * Ov,), F), ((B, Ov), F), ((H, NG), F), ((Ov, H), F),
                                                       ((_sliced_len(GW),), F))        # gw: sliced accumulator
        call("svnet_xyzblock_bwd_prelude_f32", _p(gs), _p(gv), _p(y_max), _p(y_min), _p(mv), _p(mvn), _p(coef), _p(gate), P, N, Os, Ov, 0.2,
             _p(gy), _p(red), _p(redv), _p(dgate), _p(gs2), gs2_ld, _p(gv2), gv2_ld, _p(gv_sum), _stream())
        if gv_sum is not None:
            gv = gv_sum
        bcoef = torch.empty((3 * Os + 2 * Ov,), **f32)
        dg1, db1 = torch.empty((Os,), **f32), torch.empty((Os,), **f32)
        dg2, db2 = torch.empty((Ov,), **f32), torch.empty((Ov,), **f32)
        # gate MLP backward, in extra workgroups of the coefficient launch
        gconst = torch.empty((B, NG), **f32)
        inv_nk = 1.0 / float(N * k)
        Wg0c, Wg2c = _f32c(Wg0), _f32c(Wg2)
        job = _lib.GateBwdJob(_p(dgate), _p(gate), _p(h), _p(gin), inv_nk, _p(Wg0c), _p(Wg2c), B, NG, H, Ov, inv_nk, _p(gconst), _p(dWg0), _p(dWg2))
        call("svnet_edgeblock_bwd_coeffs_f32", _p(red), _p(redv), _p(coef), _p(g1), _p(g2), E, Os, Ov, int(training), None, _p(bcoef),
             _p(dg1), _p(db1), _p(dg2), _p(db2), ctypes.byref(job), _stream())
        # edge pass: parameter gradients
        d = XyzBlockBwdDesc()
        d.B, d.N, d.k, d.Os, d.Ov = B, N, k, Os, Ov
        d.x, d.idx, d.w0, d.wz, d.w1, d.w2 = _p(x), _p(idx), _p(W0), _p(Wz), _p(W1), _p(W2)
        d.slot_max, d.slot_min, d.coef, d.bcoef, d.gate = _p(slot_max), _p(slot_min), _p(coef), _p(bcoef), _p(gate)
        d.gy, d.gv, d.gconst, d.gw = _p(gy), _p(gv), _p(gconst), _p(gw)
        d.nc = NC
        call("svnet_xyzblock_bwd_f32", ctypes.byref(d), _stream())
        call("svnet_slices_sum_f32", _p(gw), GW, _stream())
        o = Os * NF
        dW1 = gw[:o].view(Os, NF)
        dW2 = gw[o:o + Ov * NC].view(Ov, NC)
        dW0 = gw[o + Ov * NC:o + Ov * NC + NG].view(3, NC)
        dWz = gw[o + Ov * NC + NG:o + Ov * NC + 2 * NG].view(3, NC)
        # forward args: x, idx, k, training, W0, Wz, W1, g1, b1, rm1, rv1, W2, g2, b2, rm2, rv2, Wg0, Wg2
        return (None, None, None, None, dW0, dWz, dW1, dg1, db1, None, None, dW2, dg2, db2, None, None, dWg0, dWg2, None, None)


_SIDE_STREAMS = {}


def _side_stream(dev):
    """One extra HIP stream per device for independent kernels of a fused backward (forked / joined with events, so the
    pattern is also valid inside a hipGraph capture)."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        # (a high-priority side stream was measured: 10.7 ms per step against 6.4 - the tile kernel on the main stream starves)
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key, priority=config.SIDE_PRIORITY)
    return _SIDE_STREAMS[key]


def _aux_stream(dev, cur):
    """A helper stream OF the stream `cur`: the weight-gradient product of a big dense layer's backward beside its input-gradient
    product (the two only share their inputs).  One per origin stream (main, and _side_stream when it carries the other path of an
    SVBlock): a helper forked from two different streams of one hipGraph capture crashed hipStreamEndCapture."""
    key = ("aux", torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), cur.cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key[1])
    return _SIDE_STREAMS[key]


class _Beside:
    """`with _Beside(dev, on): ...` runs the block on the aux stream, forked from the current stream at entry; .join() makes the
    current stream wait for it and hands the given tensors (allocated in the block) over to it.  `on` False: a no-op."""

    def __init__(self, dev, on):
        self.on = bool(on)
        if self.on:
            self.cur = torch.cuda.current_stream(dev)
            if self.cur.cuda_stream == _side_stream(dev).cuda_stream:
                self.on = False        # already on a forked stream (the vector path of an SVBlock): a second-level fork crashed hipStreamEndCapture
                return
            self.aux = _aux_stream(dev, self.cur)
            self.ctx = torch.cuda.stream(self.aux)

    def __enter__(self):
        if self.on:
            self.aux.wait_stream(self.cur)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.on:
            self.cur.wait_stream(self.aux)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.cur)


_PERM_CACHE = {}
DEBUG_BUFFER = None     # optional int64[4] device tensor: first out-of-range neighbour id seen by the fused backward


def _fused_columns(Cs, Cv, dev):
    """column of the fused bit order (5 x 64) that holds reference feature f of [s_j-s_i | s_i | s_v (c2*3+jz)]"""
    key = (Cs, Cv, str(dev))
    if key not in _PERM_CACHE:
        f = torch.arange(2 * Cs + 6 * Cv)
        g = (f - 2 * Cs).clamp(min=0)
        col = torch.where(f < Cs, f, torch.where(f < 2 * Cs, 64 + f - Cs, 128 + 64 * (g % 3) + g // 3))
        _PERM_CACHE[key] = col.to(dev)
    return _PERM_CACHE[key]
