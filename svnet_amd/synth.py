"""Deterministic synthetic clouds, labels and weights.

Everything is derived from a counter-based hash (splitmix64) so that this
container and the GPU box produce bit-identical inputs without shipping arrays.
Pure numpy: importable without a GPU and without the HIP library.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """One splitmix64 output per uint64 counter (vectorised)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def stream_id(name):
    """Stable 32-bit id of a named stream (e.g. a parameter name)."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def _counters(seed, stream, n, lane):
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(int(seed) << 32) ^ int(stream)], dtype=np.uint64))[0]
        base = _splitmix64(np.array([base ^ np.uint64(lane)], dtype=np.uint64))[0]
        return (base + np.arange(n, dtype=np.uint64)) & _M64


def uniform01(seed, stream, n, lane=0):
    """n doubles in (0, 1]."""
    bits = _splitmix64(_counters(seed, stream, n, lane))
    return ((bits >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def normal(seed, stream, shape):
    """iid N(0,1) as float32 (Box-Muller evaluated in float64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, stream, n, lane=1)
    u2 = uniform01(seed, stream, n, lane=2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def uniform(seed, stream, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, stream, n, lane=3)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def integers(seed, stream, shape, mod):
    n = int(np.prod(shape)) if len(shape) else 1
    bits = _splitmix64(_counters(seed, stream, n, lane=4))
    return (bits % np.uint64(mod)).astype(np.int64).reshape(shape)


def cloud_batch(seed, step, rank, batch, num_points):
    """[B,3,N] float32 clouds: iid gaussian points, centred, scaled into the unit sphere
    (the normalisation the reference's loader applies, data.py:15-20)."""
    stream = stream_id("cloud/%d/%d" % (step, rank))
    pts = normal(seed, stream, (batch, num_points, 3)).astype(np.float64)
    pts -= pts.mean(axis=1, keepdims=True)
    pts /= np.sqrt((pts ** 2).sum(-1)).max(axis=1)[:, None, None]
    return np.ascontiguousarray(pts.transpose(0, 2, 1)).astype(np.float32)


def class_labels(seed, step, rank, batch, num_class=40):
    return integers(seed, stream_id("label/%d/%d" % (step, rank)), (batch,), num_class)


def category_onehot(seed, step, rank, batch, num_cat=16):
    cat = integers(seed, stream_id("cat/%d/%d" % (step, rank)), (batch,), num_cat)
    out = np.zeros((batch, num_cat), dtype=np.float32)
    out[np.arange(batch), cat] = 1.0
    return out


def seg_labels(seed, step, rank, batch, num_points, num_part=50):
    return integers(seed, stream_id("seg/%d/%d" % (step, rank)), (batch, num_points), num_part)


def random_rotation(seed, stream):
    """A proper rotation matrix (float64 3x3) from a hashed quaternion."""
    q = normal(seed, stream, (4,)).astype(np.float64)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def synthetic_state(spec, seed, trained_like=True):
    """Fill a {name: shape} spec with deterministic 'trained-like' values.

    Names follow the reference's state_dict layout:
      *.weight of linear layers  ~ U(-1/sqrt(in), 1/sqrt(in))
      *.beta  (activation shift) ~ N(0, 0.3)   (0 when trained_like=False)
      *.scale                    ~ (1/sqrt(in)) * (1 + 0.1 N)
      BN weight ~ 1 + 0.6 N (a few negative channels), BN bias ~ 0.1 N,
      running_mean ~ 0.1 N, running_var ~ 1 + 0.1 |N|.
    """
    out = {}
    for name, shape in spec.items():
        sid = stream_id(name)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = 0.1 * normal(seed, sid, shape)
        elif leaf == "running_var":
            out[name] = 1.0 + 0.1 * np.abs(normal(seed, sid, shape))
        elif leaf == "beta":
            out[name] = (0.3 * normal(seed, sid, shape)) if trained_like else np.zeros(shape, np.float32)
        elif leaf == "scale":
            fan_in = spec[name[: -len("scale")] + "weight"][1]
            out[name] = ((1.0 + 0.1 * normal(seed, sid, shape)) / np.sqrt(fan_in)).astype(np.float32)
        elif leaf == "bias" and len(shape) == 1 and (name[: -len("bias")] + "running_mean") in spec:
            out[name] = 0.1 * normal(seed, sid, shape)
        elif leaf == "weight" and len(shape) == 1:
            out[name] = (1.0 + 0.6 * normal(seed, sid, shape)).astype(np.float32)
        elif leaf == "weight":
            bound = 1.0 / np.sqrt(shape[1])
            out[name] = uniform(seed, sid, shape, -bound, bound)
        elif leaf == "bias":
            out[name] = uniform(seed, sid, shape, -0.1, 0.1)
        else:
            raise KeyError("no synthetic rule for %s" % name)
        out[name] = np.asarray(out[name], dtype=np.int64 if leaf == "num_batches_tracked" else np.float32)
    return out
