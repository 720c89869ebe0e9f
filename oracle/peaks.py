"""NumPy restatement of the reference's belief-map peak extraction.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  * peaks_from_belief_maps            /root/reference/dream/image_proc.py:914-1018
  * the keypoint selection rule       /root/reference/dream/network.py:529-581
  * create_belief_map                 /root/reference/dream/image_proc.py:866-910
The Gaussian smoothing inside peaks_from_belief_maps is ``scipy.ndimage.gaussian_filter(m, 3)``
(image_proc.py:935).  scipy is a third-party dependency (requirements.txt:13, unpinned; 1.15.3
in this image) whose source is not under /root/reference; its published algorithm is restated in
``gaussian_filter_sigma3`` below and pinned bit-for-bit against the installed scipy by
tests/test_oracle_peaks.py, and against outputs of the real reference function through
tests/golden/peaks_golden.npz.
"""
import numpy as np

SIGMA = 3.0
RADIUS = int(4.0 * SIGMA + 0.5)          # scipy truncate=4.0 -> 12, 25 taps
THRESH = 0.01                            # image_proc.py:925
NO_DETECTION = -999.999                  # network.py:572,577
NEXT_BEST_SCORE = 0.25                   # network.py:191


def gaussian_weights():
    """scipy.ndimage._filters._gaussian_kernel1d(sigma=3, order=0, radius=12), float64."""
    x = np.arange(-RADIUS, RADIUS + 1)
    phi = np.exp(-0.5 / (SIGMA * SIGMA) * x ** 2)
    return phi / phi.sum()


def _correlate_symmetric(x32, axis, w):
    """One scipy NI_Correlate1D pass, symmetric-kernel branch: the line is converted to double,
    extended by half-sample reflection ('reflect' == numpy 'symmetric'), accumulated as
    centre*w[c] then (x[l+i]+x[l-i])*w[c+i] for i=-R..-1, and stored back as float32."""
    x = np.moveaxis(x32.astype(np.float64), axis, 0)
    n = x.shape[0]
    xp = np.pad(x, [(RADIUS, RADIUS)] + [(0, 0)] * (x.ndim - 1), mode="symmetric")
    acc = xp[RADIUS:RADIUS + n] * w[RADIUS]
    for i in range(-RADIUS, 0):
        acc = acc + (xp[RADIUS + i:RADIUS + i + n] + xp[RADIUS - i:RADIUS - i + n]) * w[RADIUS + i]
    return np.moveaxis(acc.astype(np.float32), 0, axis)


def gaussian_filter_sigma3(m32):
    """== scipy.ndimage.gaussian_filter(m32, sigma=3) for a 2-D float32 array (axis 0, then 1)."""
    assert m32.dtype == np.float32 and m32.ndim == 2
    w = gaussian_weights()
    return _correlate_symmetric(_correlate_symmetric(m32, 0, w), 1, w)


def peak_mask(smooth):
    """image_proc.py:936-954: >= each 4-neighbour (zero outside the map) and > 0.01."""
    z = np.zeros_like(smooth)
    up, down, left, right = z.copy(), z.copy(), z.copy(), z.copy()
    up[1:, :] = smooth[:-1, :]
    down[:-1, :] = smooth[1:, :]
    left[:, 1:] = smooth[:, :-1]
    right[:, :-1] = smooth[:, 1:]
    return ((smooth >= up) & (smooth >= down) & (smooth >= left) & (smooth >= right)
            & (smooth > np.float32(THRESH)))


def centroid_5x5(map_ori, x, y, offset):
    """image_proc.py:961-998: weighted mean of column / row index over the 5x5 window of the
    UNFILTERED map; out-of-map cells have weight 0 and index value 0; arrays are laid out
    [col offset][row offset] (matters only for the float64 summation order)."""
    h, w = map_ori.shape
    wts = np.zeros((5, 5))
    iv = np.zeros((5, 5))
    jv = np.zeros((5, 5))
    for i in range(-2, 3):          # row offset
        for j in range(-2, 3):      # col offset
            if 0 <= y + i < h and 0 <= x + j < w:
                iv[j + 2, i + 2] = y + i
                jv[j + 2, i + 2] = x + j
                wts[j + 2, i + 2] = map_ori[y + i, x + j]
    try:
        return (np.average(jv, weights=wts) + offset, np.average(iv, weights=wts) + offset)
    except ZeroDivisionError:
        return (x + offset, y + offset)


def peaks_from_belief_maps(maps, offset_due_to_upsampling):
    """maps: float32 [K,H,W] ndarray.  Returns list (len K) of lists of (x, y, score, id)."""
    maps = np.asarray(maps)
    assert maps.ndim == 3
    all_peaks, counter = [], 0
    for k in range(maps.shape[0]):
        ori = np.ascontiguousarray(maps[k], dtype=np.float32)
        ys, xs = np.nonzero(peak_mask(gaussian_filter_sigma3(ori)))     # row-major order
        found = []
        for n, (x, y) in enumerate(zip(xs, ys)):
            cx, cy = centroid_5x5(ori, int(x), int(y), offset_due_to_upsampling)
            found.append((cx, cy, ori[y, x], counter + n))
        counter += len(found)
        all_peaks.append(found)
    return all_peaks


def select_keypoints(peaks_per_map, use_belief_peak_scores=True, belief_peak_next_best_score=NEXT_BEST_SCORE):
    """network.py:546-577 for one frame: exactly one peak -> it; several -> (when use_belief_peak_scores, :553) the
    best-scoring one iff it beats the runner-up by >= belief_peak_next_best_score (0.25, :189-191; float32 difference);
    otherwise (-999.999, -999.999)."""
    out = []
    for peaks in peaks_per_map:
        if len(peaks) == 1:
            out.append([peaks[0][0], peaks[0][1]])
        elif len(peaks) > 1 and use_belief_peak_scores:
            ranked = sorted(peaks, key=lambda p: p[2], reverse=True)
            if float(ranked[0][2] - ranked[1][2]) >= belief_peak_next_best_score:
                out.append([ranked[0][0], ranked[0][1]])
            else:
                out.append([NO_DETECTION, NO_DETECTION])
        else:
            out.append([NO_DETECTION, NO_DETECTION])
    return out


def upsampling_offset(trained_out_w, trained_out_h):
    """network.py:534-538."""
    return 0.0 if (trained_out_w >= 400 and trained_out_h >= 400) else 0.4395


def keypoints_from_belief_maps(maps_bkhw, offset, use_belief_peak_scores=True, belief_peak_next_best_score=NEXT_BEST_SCORE):
    """The whole post-CNN part of DreamNetwork.inference: float32 [B,K,H,W] -> float32 [B,K,2]."""
    maps_bkhw = np.asarray(maps_bkhw)
    res = [select_keypoints(peaks_from_belief_maps(frame, offset), use_belief_peak_scores, belief_peak_next_best_score)
           for frame in maps_bkhw]
    return np.asarray(res, dtype=np.float64).astype(np.float32).reshape(maps_bkhw.shape[0], -1, 2)


def create_belief_map(image_resolution, points, sigma=2):
    """image_proc.py:866-910: 9x9 (for sigma 2) exp(-(dx^2+dy^2)/(2 sigma^2)) blob at the
    int()-truncated point, drawn only when the whole window (plus one) is inside the frame."""
    width, height = image_resolution
    out = np.zeros((len(points), height, width))
    w = int(sigma * 2)
    dy, dx = np.mgrid[-w:w + 1, -w:w + 1]
    blob = np.exp(-((dx ** 2 + dy ** 2) / (2 * (sigma ** 2))))
    for n, pt in enumerate(points):
        u, v = int(pt[0]), int(pt[1])
        if u - w >= 0 and u + w + 1 < width and v - w >= 0 and v + w + 1 < height:
            out[n, v - w:v + w + 1, u - w:u + w + 1] = blob
    return out


def numpy_pairwise_sum25(a):
    """NumPy's float64 pairwise summation specialised to 25 contiguous elements (n < 128 block:
    8 strided partial sums over the first 24, combined as a balanced tree, then the tail added
    sequentially).  Restated so the C oracle and the HIP kernel can reproduce np.average
    bit-for-bit; checked against ndarray.sum() in tests/test_oracle_peaks.py."""
    a = [float(v) for v in a]
    r = a[0:8]
    for i in (8, 16):
        r = [r[j] + a[i + j] for j in range(8)]
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    return res + a[24]


def c_keypoints_from_belief_maps(maps_bkhw, offset):
    """Same result as keypoints_from_belief_maps through the plain-C restatement (oracle/peaks_c.c,
    built by __graft_entry__.build_oracle()); ~100x faster, used as the single-core CPU baseline."""
    import ctypes
    import os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpeaks_c.so"))
    m = np.ascontiguousarray(maps_bkhw, dtype=np.float32)
    b, k, h, w = m.shape
    kps = np.empty((b * k, 2), np.float32)
    counts = np.empty((b * k,), np.int32)
    taps = np.ascontiguousarray(gaussian_weights())
    P = ctypes.c_void_p
    rc = lib.dream_oracle_keypoints(P(m.ctypes.data), b * k, h, w, ctypes.c_double(offset), P(taps.ctypes.data),
                                    P(kps.ctypes.data), P(counts.ctypes.data))
    assert rc == 0
    return kps.reshape(b, k, 2), counts.reshape(b, k)
