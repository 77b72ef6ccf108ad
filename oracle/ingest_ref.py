"""CPU oracle (numpy, float64) of the real-image ingest — TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg); the product path is sota_imagenet_amd/csrc/ingest.hip and never imports this.

Restates the GPU half of the reference's DALI pipelines (read as text; nothing is imported from /root/reference):
  train  sota_imagenet/dali_dataloader.py:69-78   random crop at decode -> fn.resize(size=S, INTERP_TRIANGULAR)
         sota_imagenet/dali_dataloader.py:111-120 crop_mirror_normalize(mirror=coin, mean 127.5, std 51 (:27-29), FLOAT, NCHW)
  val    sota_imagenet/dali_dataloader.py:144-157 resize_shorter = ceil((S*1.14 + 8) // 16 * 16), centre crop, normalise
The arithmetic of DALI's resampling kernel is in the un-vendored `nvidia.dali` (no version pinned: docker/Dockerfile builds on
nvcr.io/nvidia/pytorch:20.09-py3), so the filter is restated from its published definition — a separable triangle whose
support grows with the down-scale factor — with Pillow's border law (taps outside the image are dropped and the rest
renormalised).  PIN: tests/test_image_loader_host.py checks `resize()` against Pillow's own Image.resize(BILINEAR) on u8 images
(<= 1 LSB + Pillow's 8-bit intermediate), an implementation this file shares no code with.  Parity with DALI itself: unpinned."""
import math

import numpy as np

DATA_MEAN, DATA_STD = 127.5, 51.0  # sota_imagenet/dali_dataloader.py:27-29 (0.5 * 255, 0.2 * 255, all three channels)


def tri_taps(o, n_in, n_out):
    """(lo, weights) of output index o: taps lo .. lo+len(weights)-1 of the source axis, weights normalised (float64).
    The tap RANGE is computed in float32 exactly as the kernel does (a tap at the boundary has weight ~0 either way)."""
    f = np.float32
    scale = f(n_in) / f(n_out)
    support = max(scale, f(1.0))
    centre = (f(o) + f(0.5)) * scale
    lo = max(0, int(centre - support + f(0.5)))
    hi = min(n_in, int(centre + support + f(0.5)))
    x = np.arange(lo, hi, dtype=np.float64)
    w = np.clip(1.0 - np.abs(x + 0.5 - float(centre)) / float(support), 0.0, None)
    return lo, w / w.sum()


def axis_matrix(n_in, n_out, first, count, mirror=False):
    """[count, n_in] resampling matrix of output indices first..first+count-1 (reversed when mirror)."""
    A = np.zeros((count, n_in))
    for k in range(count):
        o = first + (count - 1 - k if mirror else k)
        lo, w = tri_taps(o, n_in, n_out)
        A[k, lo:lo + len(w)] = w
    return A


def _apply(Ay, Ax, img):
    """rows then columns: [i,h] x [h,w,c] x [j,w] -> [i,j,c] (float64)"""
    h, w, c = img.shape
    t = (Ay @ img.astype(np.float64).reshape(h, w * c)).reshape(Ay.shape[0], w, c)
    return np.einsum("iwc,jw->ijc", t, Ax, optimize=True)


def resize(img, rh, rw):
    """whole image [h,w,3] u8 -> float64 [rh,rw,3], triangular filter"""
    h, w = img.shape[:2]
    Ay, Ax = axis_matrix(h, rh, 0, rh), axis_matrix(w, rw, 0, rw)
    return _apply(Ay, Ax, img)


def ingest_one(img, rh, rw, oy, ox, S, mirror, mean=DATA_MEAN, std=DATA_STD):
    """one sample of mi355_ingest_u8: -> float32 [3,S,S]"""
    h, w = img.shape[:2]
    Ay = axis_matrix(h, rh, oy, S)
    Ax = axis_matrix(w, rw, ox, S, mirror=bool(mirror))
    v = _apply(Ay, Ax, img).transpose(2, 0, 1)
    return ((v - mean) / std).astype(np.float32)


def val_geometry(h, w, S, full_crop=False):
    """val pipeline (:144-157): shorter side -> crop_size keeping the aspect ratio, then the centred S x S window.
    Returns (rh, rw, oy, ox).  (DALI rounds the longer side to nearest and centres with crop_pos 0.5; the window offset is
    taken as floor((r - S) / 2) here — the half-pixel case is unpinned.)"""
    crop = S if full_crop else math.ceil((S * 1.14 + 8) // 16 * 16)
    if h <= w:
        rh, rw = crop, max(crop, int(round(w * crop / h)))
    else:
        rh, rw = max(crop, int(round(h * crop / w))), crop
    return rh, rw, (rh - S) // 2, (rw - S) // 2
