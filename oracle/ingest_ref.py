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


def _kernel(t, filt):
    t = np.abs(t)
    if filt == 0:  # triangle
        return np.clip(1.0 - t, 0.0, None)
    a = -0.5  # cubic convolution (Keys), the kernel Pillow calls BICUBIC
    return np.where(t < 1.0, ((a + 2.0) * t - (a + 3.0)) * t * t + 1.0, np.where(t < 2.0, ((a * t - 5.0 * a) * t + 8.0 * a) * t - 4.0 * a, 0.0))


def taps(o, n_in, n_out, filt=0):
    """(lo, weights) of output index o: taps lo .. lo+len(weights)-1 of the source axis, weights normalised (float64).
    The tap RANGE is computed in float32 exactly as the kernel does (a tap at the boundary has weight ~0 either way)."""
    f = np.float32
    scale = f(n_in) / f(n_out)
    fs = max(scale, f(1.0))
    support = f(2.0 if filt else 1.0) * fs
    centre = (f(o) + f(0.5)) * scale
    lo = max(0, int(centre - support + f(0.5)))
    hi = min(n_in, int(centre + support + f(0.5)))
    x = np.arange(lo, hi, dtype=np.float64)
    w = _kernel((x + 0.5 - float(centre)) / float(fs), filt)
    return lo, w / w.sum()


def tri_taps(o, n_in, n_out):
    return taps(o, n_in, n_out, 0)


def axis_matrix(n_in, n_out, first, count, mirror=False, filt=0):
    """[count, n_in] resampling matrix of output indices first..first+count-1 (reversed when mirror)."""
    A = np.zeros((count, n_in))
    for k in range(count):
        o = first + (count - 1 - k if mirror else k)
        lo, w = taps(o, n_in, n_out, filt)
        A[k, lo:lo + len(w)] = w
    return A


def _apply(Ay, Ax, img):
    """rows then columns: [i,h] x [h,w,c] x [j,w] -> [i,j,c] (float64)"""
    h, w, c = img.shape
    t = (Ay @ img.astype(np.float64).reshape(h, w * c)).reshape(Ay.shape[0], w, c)
    return np.einsum("iwc,jw->ijc", t, Ax, optimize=True)


def resize(img, rh, rw, filt=0):
    """whole image [h,w,3] u8 -> float64 [rh,rw,3] (cubic: clipped to the 8-bit range like an 8-bit resampler's output)"""
    h, w = img.shape[:2]
    out = _apply(axis_matrix(h, rh, 0, rh, filt=filt), axis_matrix(w, rw, 0, rw, filt=filt), img)
    return np.clip(out, 0.0, 255.0) if filt else out


# ---- augmentations of the train pipeline (dali_dataloader.py:85-114), applied to the resized window in this order -------------
LUMA = np.array([0.299, 0.587, 0.114])
RGB2YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]])
YIQ2RGB = np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.107, 1.705]])


def twist_matrix(brightness=1.0, contrast=1.0, hue_deg=0.0, saturation=1.0):
    """3 x 4 colour matrix of fn.color_twist on 0..255 values: hue rotation / saturation scaling of the chroma plane in YIQ, then
    contrast about 128 (the uint8 contrast centre), then brightness:  v' = b * (128 + c * (HS v - 128))."""
    h = math.radians(hue_deg)
    rot = np.array([[1.0, 0.0, 0.0], [0.0, saturation * math.cos(h), -saturation * math.sin(h)], [0.0, saturation * math.sin(h), saturation * math.cos(h)]])
    hs = YIQ2RGB @ rot @ RGB2YIQ
    m = np.zeros((3, 4))
    m[:, :3] = brightness * contrast * hs
    m[:, 3] = brightness * (1.0 - contrast) * 128.0
    return m


def gaussian_blur(win, sigma):
    """[S,S,3] float64, window 11, reflect-101 border (fn.gaussian_blur window_size=11 :86)"""
    d = np.arange(-5, 6)
    w = np.exp(-0.5 * (d / sigma) ** 2)
    w /= w.sum()
    S = win.shape[0]

    def idx(k):
        k = np.abs(k)
        k = np.where(k >= S, 2 * S - 2 - k, k)
        return np.clip(k, 0, S - 1)

    rows = sum(w[t] * win[idx(np.arange(S) + d[t])] for t in range(11))
    return sum(w[t] * rows[:, idx(np.arange(S) + d[t])] for t in range(11))


def augment(win, aug, mean=DATA_MEAN):
    """win: [S,S,3] float64 resized window (unmirrored, 0..255).  aug: dict(color 3x4, blur_sigma, gray, boxes [(y0,x0,y1,x1)])"""
    if aug.get("blur_sigma", 0.0) > 0.0:
        win = gaussian_blur(win, aug["blur_sigma"])
    m = np.asarray(aug.get("color", np.eye(3, 4)), dtype=np.float64).reshape(3, 4)
    win = np.clip(win @ m[:, :3].T + m[:, 3], 0.0, 255.0)
    if aug.get("gray", 0):
        win = np.repeat((win @ LUMA)[..., None], 3, axis=-1)
    win = win.copy()
    for (y0, x0, y1, x1) in aug.get("boxes", []):
        win[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)] = mean
    return win


def ingest_one(img, rh, rw, oy, ox, S, mirror, mean=DATA_MEAN, std=DATA_STD, filt=0, aug=None):
    """one sample of mi355_ingest_u8 / mi355_ingest_u8_aug: -> float32 [3,S,S]"""
    h, w = img.shape[:2]
    win = _apply(axis_matrix(h, rh, oy, S, filt=filt), axis_matrix(w, rw, ox, S, filt=filt), img)
    if filt:
        win = np.clip(win, 0.0, 255.0)
    if aug is not None:
        win = augment(win, aug, mean)
    if mirror:
        win = win[:, ::-1]
    return ((win.transpose(2, 0, 1) - mean) / std).astype(np.float32)


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
