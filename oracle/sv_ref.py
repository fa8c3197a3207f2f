"""ORACLE — functional CPU restatement of the SVNet hot path (test infrastructure only).

State lives in a plain {name: tensor} dict that uses the reference's state_dict names
(SURVEY.md Appendix D), so one set of weights can be loaded into the reference, into this
oracle and into the HIP product.  `Ctx` carries train/eval mode, the k-NN backend and, in
train mode, collects the BatchNorm running-statistic updates the reference applies in place.

Reference files restated (all under /root/reference/models/):
  utils/sv_util.py:19-144   knn, get_graph_feature[_cross|_sv], svpool, svcat
  sv_layers.py:20-244       Linear, Conv1d, VectorBN, Vector2Scalar, VectorReLU, SVBlock, SVFuse, SV_STNkd
  sv_dgcnn_cls.py:22-82, sv_pointnet_cls.py:12-81, sv_dgcnn_partseg.py:40-128, sv_pointnet_partseg.py:12-97   the four callers
  ../utils.py:33-50         cal_loss
"""
import torch
import torch.nn.functional as F

from . import knn as _knn

EPS = 1e-6           # sv_layers.py:18
BN_EPS = 1e-5        # nn.BatchNorm1d default
BN_MOMENTUM = 0.1    # nn.BatchNorm1d default
STE_CLIP = 1.2       # sv_layers.py:41,47


class Ctx:
    """exact_ste: train-mode binarize evaluates to EXACTLY sign() (same STE gradient) instead of the reference's
    fp32 `(sign + x) - x`, whose result is 1 +- 1.2e-7 in 10-20 % of the elements.  That noise is the only thing
    that separates max-pool ties between equal integer popcounts in the reference, so WHICH tied element receives
    the gradient there is an accident of rounding; with exact_ste the first-index rule of torch.max decides.
    pool_record / pool_replay: every max-pool appends its arg-max to `pool_record` (a list) and, when
    `pool_replay` is given (a list recorded by another run), takes its selection from there instead of from its
    own values -- this is how tests/golden/make_golden.py proves that the two modes differ ONLY in tie-breaks."""

    def __init__(self, train=False, knn="exact", collect_bn=False, exact_ste=False, pool_record=None, pool_replay=None):
        self.train = train
        self.knn = knn                      # "exact" (C fmaf chain) | "torch" (reference op chain)
        self.bn_updates = {} if collect_bn else None
        self.taps = None                    # optional {name: tensor} of intermediate results
        self.exact_ste = exact_ste
        self.pool_record = pool_record
        self.pool_replay = list(pool_replay) if pool_replay is not None else None
        self.knn_record = None              # optional list: neighbour ids of every graph, in call order
        self.knn_replay = None              # optional list of ids to use instead of recomputing (feature-space graphs)
        self.decisions = None               # optional Decisions: another implementation's discrete choices, replayed and certified
        self.decision_record = None         # optional Decisions that this run's own choices are appended to (tests of the replay itself)
        self.edge_mag = None                # (scratch of the certificate: magnitudes behind the last get_graph_feature_sv's s_j - s_i)


class Decisions:
    """The DISCRETE decisions another implementation of the path (the HIP product) took in one forward, replayed into the oracle:
      knn    neighbour lists [B,N,k] of every graph, in call order;
      signs  per binarized activation (Linear(ba) / Conv1d(binary), in call order) a pair (sign in {-1,0,+1}, STE mask in {0,1})
             shaped like that layer's flattened input rows [M,K];
      pools  the arg-max (index along the pooled axis) of every max-pool, in call order (optional: empty = the oracle's own);
      acts   {name: mask of z > 0} for the ReLU / LeakyReLU that follows the BatchNorm `name` and for the ReLU of the gate MLP whose first
             linear is `name` (channel-last rows [M,C]; optional, by NAME: a layer the other implementation fuses away simply has
             no entry and keeps the oracle's own decision).
    Everything downstream of a decision is a smooth function, so with the decisions replayed the two implementations must agree
    element-wise to rounding.  Replaying is only legitimate where the oracle ITSELF is undecided; every disagreement is therefore
    certified against the oracle's own arithmetic and logged in `log` (one dict per decision point; `largest_margin` in units of
    the threshold), and `check()` fails on any that is not a knife edge.  Thresholds (+ `noise_factor` x the fp32 oracle's own rms
    distance from a float64 run of itself at that decision point, when `truth` holds one - an ill-conditioned caller's decisions
    are undecided over a wider band, and that band is measured, not assumed):
      sign: the oracle's pre-sign value t = x + beta is within `tau` of zero relative to the magnitude of what it is summed from
            (|t| <= tau * (column rms of t + magnitude hint of the entry + |beta|));
      STE:  ||t| - 1.2| <= tau * (same magnitude);
      pool: the value at the replayed arg-max equals the maximum within tau of the largest pooled magnitude (exact ties: 0);
      act:  the oracle's pre-activation value z is within tau of zero relative to its column's rms;
      knn:  slot by slot, the squared distance (oracle features, float64) to the replayed neighbour equals the distance to the
            oracle's own neighbour within tau_knn * (|x_i|^2 + |x_j|^2), and the replayed list has no duplicates."""

    def __init__(self, knn=(), signs=(), pools=(), tau=2e-5, tau_knn=1e-5, max_fraction=1e-3, noise_factor=20.0, acts=None):
        self.knn, self.signs, self.pools = list(knn), list(signs), list(pools)
        self.acts = dict(acts or {})
        self.replay_pools = bool(self.pools)
        self.noise_factor = noise_factor
        self.value_record = None    # {"knn": [], "signs": [], "pools": []}: this run's values AT the decision points are appended (float64 run)
        self.truth = None           # the value_record of a float64 run on the same decisions: adds its distance to every threshold
        self.tau, self.tau_knn, self.max_fraction = tau, tau_knn, max_fraction
        self.log = []

    def check(self):
        """Raise unless every replayed decision that differs from the oracle's own was a certified knife edge; returns a summary."""
        assert not self.knn and not self.signs and not self.pools, "decisions left over: %d graphs, %d sign layers, %d max-pools (call order differs)" % (
            len(self.knn), len(self.signs), len(self.pools))
        assert not self.acts, "activation decisions left over (no such BatchNorm + activation in the oracle): %r" % (sorted(self.acts),)
        for e in self.log:
            assert e["uncertified"] == 0, "decision replay: %r" % (e,)
            assert e["forced"] <= max(8, self.max_fraction * e["numel"]), "decision replay: too many forced decisions: %r" % (e,)
        return {"forced": sum(e["forced"] for e in self.log), "points": len(self.log),
                "largest_margin": max([e["largest_margin"] for e in self.log] + [0.0]), "layers": [e for e in self.log if e["forced"]]}


# ----------------------------------------------------------------------------- graph utilities

def knn_indices(x, k, ctx=None):
    """sv_util.py:19-25. x: [B,C,N] (any strides). -> [B,N,k] int64, nearest first."""
    if ctx is not None and ctx.decisions is not None:
        idx = _replayed_graph(x.detach(), k, ctx.decisions)
    elif ctx is not None and ctx.knn_replay is not None:
        idx = ctx.knn_replay.pop(0)
    elif ctx is not None and ctx.knn == "torch":
        idx = _knn.knn_torch(x.detach(), k)
    else:
        idx = _knn.knn_exact(x.detach(), k)
    if ctx is not None and ctx.knn_record is not None:
        ctx.knn_record.append(idx)
    if ctx is not None and ctx.decision_record is not None:
        ctx.decision_record.knn.append(idx.clone())
    return idx


def _replayed_graph(x, k, dec):
    """Decisions.knn: take the other implementation's neighbour lists, certified against this oracle's own (see Decisions)."""
    idx = dec.knn.pop(0)
    own = _knn.knn_exact(x.float(), k)
    assert idx.shape == own.shape, (idx.shape, own.shape)
    diff = idx != own
    truth = dec.truth["knn"].pop(0) if dec.truth is not None else None
    if dec.value_record is not None:
        dec.value_record["knn"].append(x.double().clone())
    entry = {"kind": "knn", "numel": idx.numel(), "forced": int(diff.sum()), "uncertified": 0, "largest_margin": 0.0}
    if entry["forced"]:
        B, C, N = x.shape

        def d2(feat, ids):
            pts = feat.transpose(1, 2).double()                                  # [B,N,C]
            xx = (pts * pts).sum(-1)
            nb = torch.gather(pts.unsqueeze(1).expand(B, N, N, C), 2, ids.unsqueeze(-1).expand(B, N, k, C))
            return ((nb - pts.unsqueeze(2)) ** 2).sum(-1), xx.unsqueeze(-1) + torch.gather(xx.unsqueeze(1).expand(B, N, N), 2, ids)
        d_own, _ = d2(x, own)
        d_rep, mag = d2(x, idx)
        thr = dec.tau_knn * (mag + 1e-300)
        if truth is not None:       # how far this (fp32) run's distances are from the float64 run's: the noise level of the comparison
            thr = thr + dec.noise_factor * (d_rep - d2(truth, idx)[0]).pow(2).mean().sqrt()
        rel = (d_own - d_rep).abs() / thr
        srt = idx.sort(dim=-1)[0]
        dup = int((srt[..., 1:] == srt[..., :-1]).sum())
        entry["largest_margin"] = float(rel[diff].max())
        entry["uncertified"] = int((rel[diff] > 1.0).sum()) + dup
    dec.log.append(entry)
    return idx


def _forced_signs(t, dec, name, mag=None):
    """Decisions.signs for the pre-sign values t [M,K] of one binarized activation: returns (sign, ste) to use, after
    certifying every entry where the replayed decision differs from sign(t) / (|t| <= 1.2) as a knife edge (see Decisions)."""
    sgn, ste = dec.signs.pop(0)
    assert tuple(sgn.shape) == tuple(t.shape), "%s: replayed signs %s vs rows %s" % (name, tuple(sgn.shape), tuple(t.shape))
    with torch.no_grad():
        td = t.detach()
        truth = dec.truth["signs"].pop(0) if dec.truth is not None else None
        if dec.value_record is not None:
            dec.value_record["signs"].append(td.double().clone())
        own_s, own_m = torch.sign(td), (td.abs() <= STE_CLIP)
        ds, dm = sgn.to(td.dtype) != own_s, ste.bool() != own_m
        entry = {"kind": "sign", "layer": name, "numel": td.numel(), "forced": int(ds.sum()), "forced_ste": int(dm.sum()),
                 "uncertified": 0, "largest_margin": 0.0}
        if entry["forced"] or entry["forced_ste"]:
            rms = td.double().pow(2).mean(dim=0, keepdim=True).sqrt().to(td.dtype)
            thr = dec.tau * (rms + (td.abs() if mag is None else mag) + 1e-30)
            if truth is not None:   # + the column's fp32 rounding noise: rms distance of this run's values from the float64 run's
                thr = thr + dec.noise_factor * (td.double() - truth).pow(2).mean(dim=0, keepdim=True).sqrt().to(td.dtype)
            both = torch.cat([(td.abs() / thr)[ds], ((td.abs() - STE_CLIP).abs() / (thr + dec.tau * STE_CLIP))[dm]])
            entry["largest_margin"] = float(both.max())
            entry["uncertified"] = int((both > 1.0).sum())
            if entry["uncertified"]:      # what the worst one looked like (diagnostic: value, its threshold, the threshold's noise share)
                w = int(both.argmax())
                tv = torch.cat([td.abs()[ds], (td.abs() - STE_CLIP).abs()[dm]])[w]
                th = torch.cat([thr.expand_as(td)[ds], (thr + dec.tau * STE_CLIP).expand_as(td)[dm]])[w]
                entry["worst"] = {"distance_from_the_edge": float(tv), "threshold": float(th), "tau": dec.tau, "noise_factor": dec.noise_factor,
                                  "rms_of_the_layer": float(rms.mean())}
        dec.log.append(entry)
    return sgn.to(t.dtype), ste.to(t.dtype)


def _record_signs(t, ctx):
    if ctx is not None and ctx.decision_record is not None:
        td = t.detach()
        ctx.decision_record.signs.append((torch.sign(td).float(), (td.abs() <= STE_CLIP).float()))


def bn_act(x2d, P, name, slope, ctx=None):
    """act(BatchNorm1d(x)) over rows [M,C], act = ReLU (slope 0) or LeakyReLU(slope): sv_layers.py:189-190, sv_dgcnn_cls.py:76-78.
    With Decisions.acts[name] the kink decision z > 0 is the replayed one, certified where it differs from the oracle's own."""
    return kink(batch_norm(x2d, P, name, ctx), slope, name, ctx)


def kink(z, slope, name, ctx=None):
    """ReLU (slope 0) / LeakyReLU(slope) of rows z [M,C] - nn.ReLU / nn.LeakyReLU(0.2) at sv_layers.py:158,190, sv_dgcnn_cls.py:76-78,
    sv_pointnet_cls.py:78-79, sv_dgcnn_partseg.py:60-77, sv_pointnet_partseg.py:30-50 - whose decision z > 0 is replayable under
    `name` (Decisions.acts): the BatchNorm + activation layers (name = the BatchNorm's) and the gate MLP's hidden layer (name = its
    first linear's).  Without a replayed entry it is exactly F.relu / F.leaky_relu."""
    dec = ctx.decisions if ctx is not None else None
    if dec is not None and dec.value_record is not None and "acts" in dec.value_record:
        dec.value_record["acts"][name] = z.detach().double().clone()
    if dec is None or name not in dec.acts:
        return F.leaky_relu(z, slope) if slope else torch.relu(z)
    mask = dec.acts.pop(name)
    assert mask.numel() == z.numel(), "%s: replayed activation mask %s vs rows %s" % (name, tuple(mask.shape), tuple(z.shape))
    mask = mask.reshape(z.shape).bool()
    with torch.no_grad():
        zd = z.detach()
        diff = mask != (zd > 0)
        entry = {"kind": "act", "layer": name, "numel": zd.numel(), "forced": int(diff.sum()), "uncertified": 0, "largest_margin": 0.0}
        if entry["forced"]:
            thr = dec.tau * (zd.double().pow(2).mean(dim=0, keepdim=True).sqrt().to(zd.dtype) + 1e-30)
            truth = dec.truth.get("acts", {}).get(name) if dec.truth is not None else None
            if truth is not None:   # + the column's fp32 rounding noise: rms distance of this run's values from the float64 run's
                thr = thr + dec.noise_factor * (zd.double() - truth).pow(2).mean(dim=0, keepdim=True).sqrt().to(zd.dtype)
            rel = (zd.abs() / thr)[diff]
            entry["largest_margin"] = float(rel.max())
            entry["uncertified"] = int((rel > 1.0).sum())
        dec.log.append(entry)
    return torch.where(mask, z, z * slope)


def bn_act_cf(x, P, name, slope, ctx=None):
    """bn_act on channel-first [B,C,N] (part-seg heads, sv_dgcnn_partseg.py:60-77)."""
    B, C, N = x.shape
    return bn_act(x.transpose(1, 2).reshape(-1, C), P, name, slope, ctx).view(B, N, C).transpose(1, 2)


def _neighbour_rows(flat_rows, idx, B, N, k):
    """Gather rows of a [B*N, F] table for cloud-local idx [B,N,k] (sv_util.py:40-51)."""
    glob = idx + (torch.arange(B).view(B, 1, 1) * N)
    return flat_rows[glob.reshape(-1)].view(B, N, k, -1)


def graph_feature(x, k=20, idx=None, x_coord=None, first=False, ctx=None):
    """sv_util.py:28-62 (get_graph_feature). x: [B,1,3m,N] -> [B,N,k,3,2m]."""
    B, N = x.size(0), x.size(3)
    pts = x.reshape(B, -1, N)
    if idx is None:
        src = pts if x_coord is None else x_coord.reshape(B, -1, N)
        idx = knn_indices(src, k, ctx)
    m = pts.size(1) // 3
    rows = pts.transpose(2, 1).contiguous()                       # [B,N,3m], channel = m*3+d
    nbr = _neighbour_rows(rows.view(B * N, -1), idx, B, N, k).view(B, N, k, m, 3)
    ctr = rows.view(B, N, 1, m, 3).expand(B, N, k, m, 3)
    rel = nbr - ctr
    second = rel.mean(dim=2, keepdim=True).expand_as(rel) if first else ctr
    return torch.cat((rel, second), dim=3).transpose(-1, -2).contiguous()   # [B,N,k,3,2m]


def graph_feature_cross(x, k=20, idx=None, ctx=None):
    """sv_util.py:64-88 (get_graph_feature_cross). -> [B,N,k,3,3m]: (x_j-x_i, x_i, x_j x x_i)."""
    B, N = x.size(0), x.size(3)
    pts = x.reshape(B, -1, N)
    if idx is None:
        idx = knn_indices(pts, k, ctx)
    m = pts.size(1) // 3
    rows = pts.transpose(2, 1).contiguous()
    nbr = _neighbour_rows(rows.view(B * N, -1), idx, B, N, k).view(B, N, k, m, 3)
    ctr = rows.view(B, N, 1, m, 3).expand(B, N, k, m, 3)
    a, b = nbr, ctr
    crs = torch.stack((a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                       a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                       a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]), dim=-1)
    return torch.cat((nbr - ctr, ctr, crs), dim=3).transpose(-1, -2).contiguous()


def graph_feature_sv(x, k=20, idx=None, ctx=None):
    """sv_util.py:90-116 (get_graph_feature_sv). s:[B,N,Cs], v:[B,N,3,Cv] ->
    s_e [B,N,k,2Cs] = [s_j-s_i, s_i],  v_e [B,N,k,3,2Cv] = [v_j-v_i, v_i].
    A caller-supplied idx is taken as GLOBAL row ids (B*N*k of them), as the reference does (:99-111)."""
    s, v = x
    B, N, Cs = s.shape
    Cv = v.size(-1)
    if idx is None:
        feat = torch.cat([s, v.reshape(B, N, -1)], dim=-1)       # channel order: s, then v row-major (3,Cv)
        loc = knn_indices(feat.transpose(-1, -2), k, ctx)
        glob = (loc + torch.arange(B).view(B, 1, 1) * N).reshape(-1)
    else:
        glob = idx.reshape(-1)
    v_j = v.reshape(B * N, -1)[glob].view(B, N, k, 3, Cv)
    v_i = v.view(B, N, 1, 3, Cv).expand(B, N, k, 3, Cv)
    s_j = s.reshape(B * N, -1)[glob].view(B, N, k, Cs)
    s_i = s.view(B, N, 1, Cs).expand(B, N, k, Cs)
    if ctx is not None and ctx.decisions is not None:       # magnitude of what s_j - s_i is formed from (knife-edge certificate)
        ctx.edge_mag = torch.cat((s_j.detach().abs() + s_i.detach().abs(), s_i.detach().abs()), dim=-1)
    return torch.cat((s_j - s_i, s_i), dim=-1), torch.cat((v_j - v_i, v_i), dim=-1)


def _replayed_argmax(s, dim, dec):
    """Decisions.pools: the other implementation's arg-max of one max-pool, certified: where it differs from torch.max's own
    (first index), the value it selects equals the maximum within the knife-edge threshold (see Decisions)."""
    own = s.max(dim=dim, keepdim=True)[1]
    arg = dec.pools.pop(0).reshape(own.shape).long()
    diff = arg != own
    sd = s.detach()
    truth = dec.truth["pools"].pop(0) if dec.truth is not None else None
    if dec.value_record is not None:
        dec.value_record["pools"].append(sd.double().clone())
    entry = {"kind": "pool", "numel": own.numel(), "forced": int(diff.sum()), "uncertified": 0, "largest_margin": 0.0}
    if entry["forced"]:
        thr = dec.tau * (sd.abs().amax(dim=dim, keepdim=True) + 1e-30)
        if truth is not None:       # + the fp32 rounding noise of this channel's values (rms over the pooled axis)
            thr = thr + dec.noise_factor * (sd.double() - truth).pow(2).mean(dim=dim, keepdim=True).sqrt().to(sd.dtype)
        gap = (sd.gather(dim, own) - sd.gather(dim, arg)).abs() / thr
        entry["largest_margin"] = float(gap[diff].max())
        entry["uncertified"] = int((gap[diff] > 1.0).sum())
    dec.log.append(entry)
    return arg


def max_over(s, dim, keepdim=False, ctx=None):
    """torch.max over `dim` (values), with the optional arg-max record / replay of Ctx."""
    if ctx is None or (ctx.pool_record is None and ctx.pool_replay is None and not (ctx.decisions is not None and ctx.decisions.replay_pools)
                       and ctx.decision_record is None):
        return s.max(dim=dim, keepdim=keepdim)[0]
    dim = dim % s.dim()
    if ctx.decisions is not None and ctx.decisions.replay_pools:
        arg = _replayed_argmax(s, dim, ctx.decisions)
    elif ctx.pool_replay is not None:
        arg = ctx.pool_replay.pop(0)
    else:
        arg = s.max(dim=dim, keepdim=True)[1]
    if ctx.pool_record is not None:
        ctx.pool_record.append((arg, s.detach()))
    if ctx.decision_record is not None:
        ctx.decision_record.pools.append(arg.clone())
    out = s.gather(dim, arg)
    return out if keepdim else out.squeeze(dim)


def svpool(x, dim=2, keepdim=False, spool="max", ctx=None):
    """sv_util.py:118-132. s: max (or mean) over `dim`; v: mean over `dim`."""
    s, v = x
    if spool == "max":
        s = max_over(s, dim, keepdim, ctx)
    elif spool == "mean":
        s = s.mean(dim=dim, keepdim=keepdim)
    else:
        raise ValueError("not recognized pooling mean {}".format(spool))
    return s, v.mean(dim=dim, keepdim=keepdim)


def svcat(xs):
    """sv_util.py:134-144."""
    return torch.cat([a for a, _ in xs], dim=-1), torch.cat([b for _, b in xs], dim=-1)


# ----------------------------------------------------------------------------- layers

def binarize(t, train, exact=False, forced=None):
    """sv_layers.py:38-42 / :44-48. eval: sign (sign(0)=0 -> ternary). train: clamp + STE, evaluated
    in the reference's fp32 order ((sign + t) - t), identity gradient where |t| <= 1.2.
    exact: the same function with the forward value exactly sign(): sign + (t_c - t_c) (see Ctx).
    forced = (sign, ste): replayed decisions (Decisions) - the forward value is `sign`, the gradient mask is `ste`."""
    if forced is not None:
        sgn, ste = forced
        if not train:
            return sgn + (t - t).detach()
        tm = t * ste
        return sgn + (tm - tm.detach())
    if not train:
        return torch.sign(t)
    tc = torch.clamp(t, -STE_CLIP, STE_CLIP)
    if exact:
        return torch.sign(tc).detach() + (tc - tc.detach())
    return torch.sign(tc).detach() + tc - tc.detach()


def linear(x, P, name, bw=False, ba=False, ctx=None, mag=None):
    """sv_layers.py:20-53 (Linear). Params: name.weight [O,K], name.bias?, name.beta [1,K] (ba), name.scale [1,O] (bw).
    mag: optional magnitude hint per input element (what x is summed from) for the knife-edge certificate of Decisions."""
    W = P[name + ".weight"]
    bias = P.get(name + ".bias")
    if not bw and not ba:
        return F.linear(x, W, bias)
    if ba and not bw:
        raise AttributeError("Linear(ba=True, bw=False) has no scale (sv_layers.py:49)")
    train = bool(ctx and ctx.train)
    exact = bool(ctx and ctx.exact_ste)
    rows = x.reshape(-1, x.shape[-1])
    if ba:
        t = rows + P[name + ".beta"]
        forced = None
        if ctx is not None and ctx.decisions is not None:
            hint = None if mag is None else mag.reshape(rows.shape).abs() + P[name + ".beta"].detach().abs()
            forced = _forced_signs(t, ctx.decisions, name, hint)
        _record_signs(t, ctx)
        rows = binarize(t, train, exact, forced)
    y = (rows @ binarize(W, train, exact).t()) * P[name + ".scale"]
    if bias is not None:
        y = y + bias
    return y.view(x.shape[:-1] + (y.shape[-1],))


def conv1d(x, P, name, binary=False, ctx=None):
    """sv_layers.py:55-78 (Conv1d, kernel 1, no bias). x: [B,C,N]; name.weight [O,C,1], beta [1,C,1], scale [1,O,1]."""
    W = P[name + ".weight"]
    if not binary:
        return torch.einsum("oc,bcn->bon", W[:, :, 0], x)
    train = bool(ctx and ctx.train)
    exact = bool(ctx and ctx.exact_ste)
    t = x + P[name + ".beta"]
    forced = None
    if ctx is not None and ctx.decisions is not None:       # replayed decisions are stored per channel-last row [B*N, C]
        B_, C_, N_ = x.shape
        sgn, ste = _forced_signs(t.transpose(1, 2).reshape(B_ * N_, C_), ctx.decisions, name)
        forced = (sgn.view(B_, N_, C_).transpose(1, 2), ste.view(B_, N_, C_).transpose(1, 2))
    _record_signs(t.transpose(1, 2).reshape(-1, x.shape[1]), ctx)
    xb = binarize(t, train, exact, forced)
    wb = binarize(W, train, exact)
    return torch.einsum("oc,bcn->bon", wb[:, :, 0], xb) * P[name + ".scale"]


def batch_norm(x2d, P, name, ctx=None):
    """nn.BatchNorm1d over rows of [M,C] (sv_layers.py:84,166,189): train = batch statistics
    (biased variance), eval = running statistics."""
    w, b = P[name + ".weight"], P[name + ".bias"]
    if ctx is not None and ctx.train:
        if ctx.bn_updates is not None:
            with torch.no_grad():
                M = x2d.shape[0]
                mean = x2d.mean(dim=0)
                unb = x2d.var(dim=0, unbiased=False) * (M / max(M - 1, 1))
                rm, rv = P[name + ".running_mean"], P[name + ".running_var"]
                ctx.bn_updates[name + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                ctx.bn_updates[name + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unb
        # y = (x - mean_batch) / sqrt(var_batch_biased + eps) * w + b, evaluated by the same ATen
        # primitive the reference's nn.BatchNorm1d calls, so that train-mode parity with the reference
        # is not limited by the summation order of the batch statistics (binary nets amplify 1e-6
        # differences into sign flips one layer later).
        return F.batch_norm(x2d, None, None, w, b, True, 0.0, BN_EPS)
    mean, var = P[name + ".running_mean"], P[name + ".running_var"]
    return F.batch_norm(x2d, mean, var, w, b, False, 0.0, BN_EPS)


def batch_norm_cf(x, P, name, ctx=None):
    """nn.BatchNorm1d on channel-first [B,C,N] (part-seg head, sv_dgcnn_partseg.py:60-77)."""
    B, C, N = x.shape
    y = batch_norm(x.transpose(1, 2).reshape(-1, C), P, name, ctx)
    return y.view(B, N, C).transpose(1, 2)


def vector_bn(v, P, name, ctx=None):
    """sv_layers.py:81-102 (VectorBN). v: [...,3,C]. n = |v|_2 over the 3-axis + EPS; v * BN(n) / n."""
    C = v.shape[-1]
    n = torch.linalg.vector_norm(v, dim=-2) + EPS
    n_bn = batch_norm(n.reshape(-1, C), P, name + ".bn", ctx).view(n.shape)
    return v / n.unsqueeze(-2) * n_bn.unsqueeze(-2)


def vector2scalar(v, P, name, binary=False, trans_back=False, ctx=None):
    """sv_layers.py:104-129 (Vector2Scalar). z = Linear(v) [...,3,multi]; s[d*multi+j] = sum_i v[i,d] z[i,j]."""
    assert v.dim() in (3, 4, 5), "dim of v should be in [4, 5], got {}".format(v.dim())
    z = linear(v, P, name + ".linear", bw=binary, ctx=ctx)
    s = torch.matmul(v.transpose(-1, -2), z)                     # [...,C,multi]
    s = s.reshape(z.shape[:-2] + (-1,))
    return (s, z) if trans_back else s


def vector_relu(x, div=10):
    """sv_layers.py:131-149 (VectorReLU; never instantiated by any model)."""
    shape = x.shape
    B, C = shape[0], shape[-1]
    rows = x.reshape(B, -1, 3, C)
    kth = rows.shape[1] // div
    nrm = torch.sqrt((rows * rows).sum(dim=2, keepdim=True)).detach()
    thr = torch.kthvalue(nrm, kth, dim=1, keepdim=True)[0]
    return torch.where(nrm > thr, rows, torch.zeros_like(rows)).view(shape)


def svblock(x, P, name, binary=False, ctx=None):
    """sv_layers.py:151-196 (SVBlock)."""
    s, v = x
    pooled = s.reshape(s.shape[0], -1, s.shape[-1]).mean(dim=1)                       # :179-180
    gate = torch.sigmoid(F.linear(kink(F.linear(pooled, P[name + ".gate.0.weight"]), 0.0, name + ".gate.0", ctx),
                                  P[name + ".gate.2.weight"]))                         # :156-161,181
    gate = gate.view((gate.shape[0],) + (1,) * (v.dim() - 2) + (gate.shape[1],))       # :182-183
    s_v = vector2scalar(v, P, name + ".v2s", binary=binary, ctx=ctx)                    # :185
    mag = None
    if binary and ctx is not None and ctx.decisions is not None:
        # what each invariant scalar is summed from: |v|^T (|v| |sign(Wz)|^T) |scale_z| - the yard-stick of its rounding error
        with torch.no_grad():
            Wz = (torch.sign(P[name + ".v2s.linear.weight"]) * P[name + ".v2s.linear.scale"].view(-1, 1)).abs()
            va = v.detach().abs()
            s_mag = s.detach().abs()
            if getattr(ctx, "edge_mag", None) is not None and ctx.edge_mag.shape == s.shape:
                s_mag, ctx.edge_mag = ctx.edge_mag, None
            mag = torch.cat([s_mag, torch.matmul(va.transpose(-1, -2), va @ Wz.t()).reshape(s_v.shape)], dim=-1)
    y = linear(torch.cat([s, s_v], dim=-1), P, name + ".linear1", bw=binary, ba=binary, ctx=ctx, mag=mag)  # :186-187
    y = bn_act(y.reshape(-1, y.shape[-1]), P, name + ".bn1", 0.2, ctx).view(y.shape)    # :188-190
    u = linear(v, P, name + ".linear2", bw=binary, ctx=ctx)                             # :192
    u = vector_bn(u, P, name + ".bn2", ctx) * gate                                      # :193-194
    return y, u


def edge_sign_margins(x, idx, k, P, name):
    """Test diagnostics: how close the binarized invariant scalars of a fused edge layer are to a sign change.
    x = (s [B,N,Cs], v [B,N,3,Cv]) point tables, idx [B,N,k] cloud-local graph, block `name` (binary).  For every edge row the
    linear1 inputs that come from Vector2Scalar are t = s_v + beta; returns per POINT the smallest |t| / (|v_e|^T |z| + |beta|)
    over its k edges and 6Cv columns: a relative margin in units of the magnitude of the terms t is summed from.  Any
    implementation that evaluates s_v in another order (a BLAS, the fused kernels' per-point products) moves t by a few ulps of
    that magnitude; where the margin is that small the SIGN — and with it the integer popcount, possibly the arg-max — may differ."""
    s, v = x
    B, N, Cs = s.shape
    glob = (idx + torch.arange(B).view(B, 1, 1) * N).reshape(-1)
    with torch.no_grad():
        _, v_e = graph_feature_sv((s, v), k=k, idx=glob)
        Wz = torch.sign(P[name + ".v2s.linear.weight"]) * P[name + ".v2s.linear.scale"].view(-1, 1)      # [3, 2Cv]
        z = v_e @ Wz.t()                                                                                   # [B,N,k,3,3]
        s_v = torch.matmul(v_e.transpose(-1, -2), z).reshape(B, N, k, -1)
        bound = torch.matmul(v_e.abs().transpose(-1, -2), (v_e.abs() @ Wz.abs().t())).reshape(B, N, k, -1)
        beta = P[name + ".linear1.beta"].view(-1)[2 * Cs:]
        r = (s_v + beta).abs() / (bound + beta.abs() + 1e-30)
    return r.amin(dim=(2, 3))


def svfuse(x, P, name, binary, trans_back=False, ctx=None):
    """sv_layers.py:198-220 (SVFuse)."""
    s, v = x
    if trans_back:
        s_v, z = vector2scalar(v, P, name + ".v2s", binary=binary, trans_back=True, ctx=ctx)
        return torch.cat([s, s_v], dim=-1), z
    return torch.cat([s, vector2scalar(v, P, name + ".v2s", binary=binary, ctx=ctx)], dim=-1)


def sv_stnkd(x, P, name, binary, ctx=None):
    """sv_layers.py:222-244 (SV_STNkd)."""
    for blk in ("conv1", "conv2", "conv3"):
        x = svblock(x, P, name + "." + blk, binary, ctx)
    x = svpool(x, dim=1, ctx=ctx)
    for blk in ("fc1", "fc2", "fc3"):
        x = svblock(x, P, name + "." + blk, binary, ctx)
    return x


# ----------------------------------------------------------------------------- callers (models)

def _tap(ctx, key, val):
    if ctx is not None and ctx.taps is not None:
        ctx.taps[key] = val


def sv_dgcnn_cls(x, P, k=20, binary=True, ctx=None):
    """sv_dgcnn_cls.py:46-82 (SV_DGCNN_CLS.forward). x: [B,3,N] -> logits [B,num_class]."""
    B = x.size(0)
    v = graph_feature(x.unsqueeze(1), k=k, ctx=ctx)
    s = vector2scalar(v, P, "init_scalar", ctx=ctx)
    feats = []
    h = svpool(svblock((s, v), P, "conv1", False, ctx), ctx=ctx)
    feats.append(h)
    _tap(ctx, "x1", h)
    for i, blk in enumerate(("conv2", "conv3", "conv4")):
        h = svpool(svblock(graph_feature_sv(h, k=k, ctx=ctx), P, blk, binary, ctx), ctx=ctx)
        feats.append(h)
        _tap(ctx, "x%d" % (i + 2), h)
    h = svblock(svcat(feats), P, "conv5", binary, ctx)
    _tap(ctx, "x5", h)
    f = svfuse(h, P, "svfuse", binary, ctx=ctx)                   # [B,N,1022]
    g = torch.cat((max_over(f, 1, ctx=ctx), f.mean(dim=1)), dim=1)        # adaptive max / avg pool over points
    _tap(ctx, "pooled", g)
    g = bn_act(linear(g, P, "linear1", binary, binary, ctx), P, "bn1", 0.2, ctx)
    g = bn_act(linear(g, P, "linear2", binary, binary, ctx), P, "bn2", 0.2, ctx)
    return linear(g, P, "linear3")


def sv_pointnet_encoder(x, P, name, k, binary, ctx=None):
    """sv_pointnet_cls.py:31-60 (SVPointNetEncoder.forward)."""
    v = graph_feature_cross(x.unsqueeze(1), k=k, ctx=ctx)
    s = vector2scalar(v, P, name + ".init_scalar", ctx=ctx)
    h = svpool(svblock((s, v), P, name + ".conv_pos", False, ctx), ctx=ctx)
    h = svblock(h, P, name + ".conv1", binary, ctx)
    g = sv_stnkd(h, P, name + ".fstn", binary, ctx)
    g = (g[0].unsqueeze(1).expand_as(h[0]), g[1].unsqueeze(1).expand_as(h[1]))
    h = svcat([h, g])
    h = svblock(h, P, name + ".conv2", binary, ctx)
    h = svblock(h, P, name + ".conv3", binary, ctx)
    m = svpool(h, dim=1, keepdim=True, ctx=ctx)
    h = svcat([h, (m[0].expand_as(h[0]), m[1].expand_as(h[1]))])
    h = svblock(h, P, name + ".conv_fuse", binary, ctx)
    h = svpool(h, dim=1, ctx=ctx)
    return svfuse(h, P, name + ".svfuse", binary, ctx=ctx)


def sv_pointnet_cls(x, P, k=20, binary=True, ctx=None):
    """sv_pointnet_cls.py:75-81 (SV_PointNet_CLS.forward). Dropout is p=0 (binary) or eval-only here."""
    f = sv_pointnet_encoder(x, P, "feat", k, binary, ctx)
    f = bn_act(linear(f, P, "fc1", binary, binary, ctx), P, "bn1", 0.0, ctx)
    f = bn_act(linear(f, P, "fc2", binary, binary, ctx), P, "bn2", 0.0, ctx)
    return linear(f, P, "fc3")


def sv_dgcnn_pseg(x, l, P, k=40, binary=True, ctx=None):
    """sv_dgcnn_partseg.py:80-128 (SV_DGCNN_PSEG.forward). x: [B,3,N], l: [B,16] one-hot -> [B,num_part,N]."""
    B, N = x.size(0), x.size(2)
    v = graph_feature(x.unsqueeze(1), k=k, ctx=ctx)
    s = vector2scalar(v, P, "init_scalar", ctx=ctx)
    feats = []
    h = svpool(svblock((s, v), P, "conv1", False, ctx), ctx=ctx)
    feats.append(h)
    _tap(ctx, "x1", h)
    for i, blk in enumerate(("conv2", "conv3", "conv4")):
        h = svpool(svblock(graph_feature_sv(h, k=k, ctx=ctx), P, blk, binary, ctx), ctx=ctx)
        feats.append(h)
        _tap(ctx, "x%d" % (i + 2), h)
    h = svcat(feats)
    fine = svfuse(h, P, "svfuse1", binary, ctx=ctx)                                   # [B,N,544]
    h = svblock(h, P, "conv5", binary, ctx)
    pooled = svblock(svpool(h, dim=1, keepdim=True, ctx=ctx), P, "conv6", binary, ctx)
    pooled = svfuse(pooled, P, "svfuse2", binary, ctx=ctx)                            # [B,1,520]
    glob = max_over(svfuse(h, P, "svfuse3", binary, ctx=ctx), 1, ctx=ctx).unsqueeze(-1)       # [B,1016,1]
    lab = torch.einsum("oc,bcn->bon", P["conv7.0.weight"][:, :, 0], l.view(B, -1, 1))
    lab = bn_act_cf(lab, P, "conv7.1", 0.2, ctx)                                      # [B,64,1]
    g = torch.cat([glob, pooled.transpose(-1, -2), lab], dim=1).expand(-1, -1, N)
    y = torch.cat([g, fine.transpose(-1, -2)], dim=1)                                 # [B,2144,N]
    for blk in ("conv8", "conv9", "conv10"):
        y = conv1d(y, P, blk + ".0", binary, ctx)
        y = bn_act_cf(y, P, blk + ".1", 0.2, ctx)
    return torch.einsum("oc,bcn->bon", P["conv11.weight"][:, :, 0], y)


def sv_pointnet_pseg(x, l, P, k=40, binary=True, ctx=None):
    """sv_pointnet_partseg.py:53-97 (SV_PointNet_PSEG.forward). x: [B,3,N], l: [B,16] one-hot -> [B,num_part,N]."""
    B, N = x.size(0), x.size(2)
    v = graph_feature_cross(x.unsqueeze(1), k=k, ctx=ctx)
    s = vector2scalar(v, P, "init_scalar", ctx=ctx)
    h = svpool(svblock((s, v), P, "conv_pos", False, ctx), ctx=ctx)
    out1 = svblock(h, P, "conv1", binary, ctx)
    out2 = svblock(out1, P, "conv2", binary, ctx)
    out3 = svblock(out2, P, "conv3", binary, ctx)
    g = sv_stnkd(out3, P, "fstn", binary, ctx)
    g = (g[0].unsqueeze(1).expand_as(out3[0]), g[1].unsqueeze(1).expand_as(out3[1]))
    out4 = svblock(svcat([out3, g]), P, "conv4", binary, ctx)
    out5 = svblock(out4, P, "conv5", binary, ctx)
    m = svpool(out5, dim=1, keepdim=True, spool="mean", ctx=ctx)
    f, trans = svfuse(svcat([out5, (m[0].expand_as(out5[0]), m[1].expand_as(out5[1]))]), P, "svfuse", binary, trans_back=True, ctx=ctx)
    y = f.transpose(-1, -2).contiguous()                                              # [B,channels,N]
    for blk in ("conv_fuse1", "conv_fuse2"):
        y = bn_act_cf(conv1d(y, P, blk + ".0", binary, ctx), P, blk + ".1", 0.0, ctx)
    y = y.mean(dim=-1) if binary else max_over(y, -1, ctx=ctx)                         # :75-78
    x_l = torch.cat([y, l.reshape(B, -1)], dim=1).view(B, -1, 1).repeat(1, 1, N)
    cs, cv = svcat([out1, out2, out3, out4, out5])
    cv = torch.einsum("bimj,bijk->bimk", cv.transpose(-1, -2), trans).reshape(B, N, -1)   # :89
    y = torch.cat([x_l, torch.cat([cs, cv], dim=-1).transpose(-1, -2)], dim=1)
    for blk in ("convs1", "convs2", "convs3"):
        y = bn_act_cf(conv1d(y, P, blk + ".0", binary, ctx), P, blk + ".1", 0.0, ctx)
    return torch.einsum("oc,bcn->bon", P["convs4.weight"][:, :, 0], y) + P["convs4.bias"].view(1, -1, 1)


def cal_loss(pred, target, smoothing=True):
    """utils.py:33-50: label-smoothed cross entropy (eps = 0.2)."""
    target = target.reshape(-1)
    if not smoothing:
        return F.cross_entropy(pred, target)
    eps, C = 0.2, pred.size(1)
    soft = torch.full_like(pred, eps / (C - 1))
    soft.scatter_(1, target.view(-1, 1), 1 - eps)
    return -(soft * F.log_softmax(pred, dim=1)).sum(dim=1).mean()
