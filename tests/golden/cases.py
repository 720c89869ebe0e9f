"""Seeded synthetic inputs shared by make_golden.py (dev container, runs the real reference) and
by the test-suite (dev container + GPU box, compares oracle / HIP path with the stored outputs).
Pure NumPy: the same seeds give the same bytes everywhere."""
import numpy as np


def blob(res_wh, points, sigma=2):
    """create_belief_map semantics (image_proc.py:866-910), vectorised; float64 [N,H,W]."""
    width, height = res_wh
    out = np.zeros((len(points), height, width))
    w = int(sigma * 2)
    dy, dx = np.mgrid[-w:w + 1, -w:w + 1]
    g = np.exp(-((dx ** 2 + dy ** 2) / (2 * (sigma ** 2))))
    for n, pt in enumerate(points):
        u, v = int(pt[0]), int(pt[1])
        if u - w >= 0 and u + w + 1 < width and v - w >= 0 and v + w + 1 < height:
            out[n, v - w:v + w + 1, u - w:u + w + 1] = g
    return out


def peak_cases():
    """name -> (maps float32 [K,H,W], offset).  Covers: the reference KAT, noisy blobs at every
    output resolution of the four archs, border peaks, plateaus (ties), negative maps, two-blob
    maps with score gaps on both sides of 0.25, all-zero maps, a map smaller than the filter."""
    cases = {}
    cases["ref_kat_80x60"] = (blob((80, 60), [(65.0, 20.0), (100.0, 80.0)]).astype(np.float32), 0.0)
    for name, (h, w), off, seed in [("q100", (100, 100), 0.4395, 11), ("h208", (208, 208), 0.4395, 12),
                                    ("f400", (400, 400), 0.0, 13), ("r416", (416, 416), 0.0, 14),
                                    ("odd133x100", (100, 133), 0.4395, 15)]:
        rs = np.random.RandomState(seed)
        k = 7 if max(h, w) < 300 else 3
        pts = np.stack([rs.uniform(0, w, k), rs.uniform(0, h, k)], 1)
        m = blob((w, h), pts) * rs.uniform(0.3, 1.0, (k, 1, 1)) + rs.normal(0, 0.01, (k, h, w))
        cases[name] = (m.astype(np.float32), off)
    rs = np.random.RandomState(21)
    # two blobs per map, score gap swept across the 0.25 rule
    gaps = [0.0, 0.1, 0.2499, 0.25, 0.2501, 0.4, 0.9]
    two = np.zeros((len(gaps), 100, 100))
    for i, g in enumerate(gaps):
        two[i] = blob((100, 100), [(30, 30)])[0] * 1.0 + blob((100, 100), [(70, 60)])[0] * (1.0 - g)
    cases["two_blobs_gap"] = (two.astype(np.float32), 0.4395)
    edge = np.zeros((6, 60, 80), np.float32)
    edge[0, 0, 0] = 1.0                      # corner spike
    edge[1, 59, 79] = 2.0                    # other corner
    edge[2, 0, 40] = 1.0                     # top edge
    edge[3, 30, 0] = 1.0                     # left edge
    edge[4, 20:24, 30:34] = 0.5              # plateau -> ties in the smoothed map
    edge[5, 10, 10] = 1.0
    edge[5, 10, 69] = 1.0                    # symmetric twin peaks -> equal scores
    cases["edges_plateau"] = (edge, 0.4395)
    neg = (rs.normal(-0.2, 0.05, (3, 50, 50))).astype(np.float32)
    neg[1, 25, 25] = 3.0                     # one strong positive in a negative map
    neg[2] = 0.0                             # all-zero map
    cases["negative_zero"] = (neg, 0.4395)
    cases["tiny_5x7"] = ((rs.uniform(0, 1, (2, 5, 7))).astype(np.float32), 0.0)
    cases["noise_100"] = ((rs.normal(0.05, 0.05, (4, 100, 100))).astype(np.float32), 0.4395)
    # centroid window with exactly cancelling weights -> ZeroDivisionError branch
    ring = np.zeros((40, 40), np.float32)
    yy, xx = np.mgrid[0:40, 0:40]
    r2 = (yy - 20) ** 2 + (xx - 20) ** 2
    ring[(r2 >= 16) & (r2 <= 36)] = 1.0      # ring: smoothed max at the centre, 5x5 window ~empty
    cases["ring_zero_window"] = (ring[None], 0.0)
    return cases


# (use_belief_peak_scores, belief_peak_next_best_score) settings of DreamNetwork a caller may make (network.py:189-191)
PEAK_RULE_SETTINGS = {"noscores": (False, 0.25), "thr0p1": (True, 0.1), "thr0p5": (True, 0.5), "thr0p3": (True, 0.3)}
PEAK_RULE_CASES = ("two_blobs_gap", "q100", "edges_plateau", "noise_100")


def belief_map_cases():
    """name -> ((W,H), keypoints float64 [K,2], sigma) for create_belief_map (image_proc.py:866-910): in-frame blobs, windows
    that touch each border by one pixel either way (drawn / rejected), coordinates a hair below an integer in float64
    (int() truncation: 57.9999999 is pixel 57 although its fp32 rounding is 58.0), negative and far-outside
    coordinates, a non-default sigma (window half-width int(2*sigma))."""
    rs = np.random.RandomState(41)
    c = {}
    c["inframe_80x60"] = ((80, 60), np.stack([rs.uniform(6, 73, 7), rs.uniform(6, 53, 7)], 1), 2)
    # window [u-4, u+4] must satisfy u-4 >= 0 and u+5 < W: u in [4, W-6]
    c["borders_80x60"] = ((80, 60), np.array([[4.0, 30.0], [3.0, 30.0], [74.0, 30.0], [75.0, 30.0], [40.0, 4.0],
                                              [40.0, 3.0], [40.0, 54.0], [40.0, 55.0], [4.0, 4.0], [74.0, 54.0]]), 2)
    c["truncation_100"] = ((100, 100), np.array([[57.9999999, 20.0], [58.0, 20.0], [20.0, 41.99999999999], [3.9999999, 50.0],
                                                 [4.0000001, 50.0], [94.0000001, 50.0], [93.9999999, 94.9999999], [-0.5, 50.0],
                                                 [-0.9999, 4.5], [50.5, 50.5], [1e6, 50.0], [50.0, -1e6]]), 2)
    c["sigma3_208"] = ((208, 208), np.stack([rs.uniform(0, 208, 9), rs.uniform(0, 208, 9)], 1), 3)
    c["sigma1p5_64x48"] = ((64, 48), np.stack([rs.uniform(0, 64, 8), rs.uniform(0, 48, 8)], 1), 1.5)
    c["ragged_133x100"] = ((133, 100), np.stack([rs.uniform(-5, 138, 17), rs.uniform(-5, 105, 17)], 1), 2)
    return c


def softargmax_cases():
    """name -> (maps float32 [B,K,H,W], beta)."""
    kat = np.zeros((1, 3, 20, 30), np.float32)
    kat[0, 0, 5, 7] = 10.0
    kat[0, 1, 10, 15] = 10.0
    kat[0, 2, 19, 29] = 10.0
    rs = np.random.RandomState(31)
    return {
        "kat_b25": (kat, 25.0),
        "rand_b1": (rs.normal(0, 1, (2, 7, 100, 100)).astype(np.float32), 1.0),
        "rand_b25": (rs.normal(0, 0.2, (2, 7, 100, 100)).astype(np.float32), 25.0),
        "odd_b5": (rs.normal(0, 0.5, (1, 17, 33, 21)).astype(np.float32), 5.0),
    }


def image_batch(b, h, w, seed=0):
    """SURVEY.md 8d synthetic frames: uint8 RGB -> ToTensor -> Normalize(0.5, 0.5); NCHW float32."""
    rs = np.random.RandomState(seed)
    u8 = rs.randint(0, 256, (b, h, w, 3)).astype(np.uint8)
    x = (u8.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def target_batch(b, k, out_wh, in_wh=(400, 400), seed=0):
    """panda_synth_train_dr-shaped targets: K keypoints uniform in the input frame, scaled to the
    output resolution, sigma-2 blobs; float32 [B,K,Ho,Wo]."""
    rs = np.random.RandomState(seed + 1000)
    wo, ho = out_wh
    out = np.zeros((b, k, ho, wo), np.float32)
    for i in range(b):
        pts = np.stack([rs.uniform(0, in_wh[0], k) * wo / in_wh[0],
                        rs.uniform(0, in_wh[1], k) * ho / in_wh[1]], 1)
        out[i] = blob((wo, ho), pts).astype(np.float32)
    return out


# Structured end-to-end fixture (G11): frames with coloured Gaussian blobs, and a last layer calibrated (by make_golden.py, on
# the reference's own activations) so that the belief maps are O(1) with clear peaks.  arch -> (last-layer key prefix,
# (B, H, W), weight recipe, exact-zero background?).  vgg_q: the general recipe weights on uint8-quantised noisy frames;
# resnet_h: oracle.models.structured_weights (no additive terms, bilinear transposed convs) on frames whose background is
# exactly zero, so the maps are exactly zero away from the blobs.  The calibrated last layer is stored in the fixture.
# Since round 3 keyed by CASE name: name -> (arch, manipulator, K, last layer, (B, H, W), recipe, zero background): the headline
# shape (vgg_q at 2 x 400 x 400), the deconv decoder (vgg_f) and the full-resolution ResNet decoder (resnet_f, 17 keypoints).
STRUCTURED_CASES = {
    "vgg_q": ("vgg_q", "panda", 7, "heads_0.4", (2, 200, 200), "recipe", False),
    "resnet_h": ("resnet_h", "panda", 7, "upsample.12", (2, 400, 400), "structured", True),
    "vgg_q_400": ("vgg_q", "panda", 7, "heads_0.4", (2, 400, 400), "smooth", False),
    "vgg_f": ("vgg_f", "panda", 7, "heads_0.4", (1, 160, 160), "smooth", False),
    "resnet_f": ("resnet_f", "baxter", 17, "upsample2.3", (1, 200, 200), "structured", True),
    # Round 6 (round-5 advice): the smooth recipe's symmetric binomial kernels cannot see a flipped kernel, a wrong tap order or a
    # transposed-conv phase / offset error -- they cancel out.  The deconv decoder (vgg_f) keeps a second structured case on the
    # asymmetric recipe weights (noise-like maps: fewer detections, but every tap and every phase matters).
    "vgg_f_recipe": ("vgg_f", "panda", 7, "heads_0.4", (1, 160, 160), "recipe", False),
}

# ResNet training golden (G12): name -> (arch, manipulator, K, (B, H, W), last-layer keys scaled by TRAIN_FINAL_SCALE)
STRUCTURED_BLOBS = {"vgg_q_400": 4, "vgg_f": 4, "resnet_h": 4}     # blobs per frame where not the default 3
# Round 5: per-frame contrast of the blobs (multiplies the blob's colour).  Frame 0 has ONE dominant blob -- most channel mixes then give
# one clear peak (a detection) --, frame 1 two comparable ones, which the 0.25 rule (dream/network.py:553-568) rejects for the mixes
# that weigh them alike.  make_golden.py accepts the first channel mix with >= STRUCTURED_MIN_DETECTIONS detections AND >=
# STRUCTURED_MIN_REJECTIONS maps with several peaks of which none wins by 0.25.
STRUCTURED_BLOB_GAINS = {
    "vgg_q_400": [[1.0, 0.4, 0.35, 0.3], [1.0, 0.9, 0.35, 0.3]],
    "vgg_f": [[1.0, 0.8, 0.6, 0.5]],
    "resnet_h": [[1.0, 0.4, 0.35, 0.3], [1.0, 0.62, 0.35, 0.3]],
}
# background grey level where not 96: at 128 the normalised background is ~0, i.e. what the convs' zero padding supplies -- the smooth
# networks' maps then have no dark frame along the borders
STRUCTURED_BG_LEVEL = {"vgg_q_400": 128.0, "vgg_f": 128.0}
STRUCTURED_MIN_REJECTIONS = {"vgg_q_400": 2, "vgg_f": 2, "resnet_h": 2}
# least fraction of (frame, keypoint) maps with a detection the generator accepts (default 0.25); at 400 x 400 the random
# network gives most maps several comparable peaks, which the 0.25 rule rejects
# (vgg_f_recipe: 0 -- the recipe weights give every full-resolution map several comparable peaks, all rejected by the 0.25 rule; the case
# is there for the MAP values, which every tap of every layer and every transposed-conv phase feeds asymmetrically)
STRUCTURED_MIN_DETECTIONS = {"vgg_q_400": 0.7, "vgg_f": 0.55, "resnet_h": 0.64, "vgg_f_recipe": 0.0}


def structured_input(case):
    """Frames of a structured case -> (NCHW float32, blob centres)."""
    _, _, _, _, (b, h, w), _, zero_bg = STRUCTURED_CASES[case]
    return blob_image_batch(b, h, w, seed=91, n_blobs=STRUCTURED_BLOBS.get(case, 3), zero_background=zero_bg,
                            gains=STRUCTURED_BLOB_GAINS.get(case), bg_level=STRUCTURED_BG_LEVEL.get(case, 96.0))


RESNET_TRAIN_CASES = {
    "resnet_h": ("resnet_h", "panda", 7, (2, 128, 128), ("upsample.12.weight", "upsample.12.bias")),
    "resnet_f": ("resnet_f", "baxter", 17, (2, 64, 64), ("upsample2.3.weight", "upsample2.3.bias")),
}
RESNET_TRAIN_LR = 1e-6                     # SGD
RESNET_DECODER_PREFIXES = ("upsample.", "upsample2.")


def blob_image_batch(b, h, w, seed=0, n_blobs=3, zero_background=False, gains=None, bg_level=96.0):
    """RGB frames with n_blobs coloured Gaussian blobs (sigma 8-14 px).  Default: background level + noise, quantised to uint8,
    then ToTensor + Normalize(0.5, 0.5) as image_batch().  zero_background: float frames, exactly 0 away from the blobs.
    Returns (NCHW float32, blob centres [b, n_blobs, 2] (x, y))."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    imgs = np.zeros((b, h, w, 3))
    centres = np.zeros((b, n_blobs, 2))
    for i in range(b):
        img = np.zeros((h, w, 3)) if zero_background else np.full((h, w, 3), bg_level) + rs.normal(0, 2.0, (h, w, 3))
        for j in range(n_blobs):
            cx, cy = rs.uniform(0.15 * w, 0.85 * w), rs.uniform(0.15 * h, 0.85 * h)
            sg = rs.uniform(8, 14)
            colour = rs.uniform(-1, 1, 3) if zero_background else rs.uniform(-90, 150, 3)
            if gains is not None:
                colour = colour * gains[i][j]
            g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
            if zero_background:
                g[g < 1e-3] = 0.0
            img += g[..., None] * colour
            centres[i, j] = (cx, cy)
        imgs[i] = img
    if zero_background:
        x = imgs.astype(np.float32)
    else:
        u8 = np.clip(np.rint(imgs), 0, 255).astype(np.uint8)
        x = (u8.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)), centres


# CNN parity cases: arch -> (n_keypoints, manipulator, [(B, H, W), ...])
CNN_CASES = {
    "vgg_q": (7, "panda", [(2, 64, 80), (1, 400, 400), (1, 50, 75)]),
    "vgg_f": (7, "panda", [(2, 64, 80)]),
    "resnet_h": (7, "panda", [(2, 64, 96)]),
    "resnet_f": (17, "baxter", [(1, 64, 64)]),
}

# Constructor branches the shipped YAMLs do not reach (oracle.models.VARIANTS gives the architecture overrides):
# variant -> ([(B, H, W)] inference cases, training golden? (2 Adam steps at B=2, 64x96))
VARIANT_CASES = {
    "vgg_q_skip": ([(1, 64, 80)], True),
    "vgg_f_skip": ([(1, 48, 64)], True),
    "vgg_full": ([(1, 48, 64)], False),
    "vgg_q_softmax": ([(2, 64, 80)], False),
    "vgg_ms2": ([(1, 64, 80)], True),
    "vgg_f_ms2_skip": ([(1, 32, 48)], True),
    "vgg_ms3_full": ([(1, 32, 48)], False),
}
VARIANT_TRAIN_SHAPE = (2, 64, 96)

# one-training-step golden (G5): learning rates and the recipe tweak that keep the loss finite
TRAIN_LR = {"adam": 1e-5, "sgd": 1e-6}
TRAIN_FINAL_KEYS = ("heads_0.4.weight", "heads_0.4.bias")
TRAIN_FINAL_SCALE = 0.1


def keypoint_conversion_cases():
    """name -> (keypoints [N,2] float32 in the net-output frame, net_output_res, net_input_res, raw_res), all (W,H)."""
    rs = np.random.RandomState(11)
    out = {}
    for name, out_res, in_res, raw_res in (("vgg_q_vga", (100, 100), (400, 400), (640, 480)),
                                           ("resnet_h_hd", (208, 208), (400, 400), (1280, 720)),
                                           ("full_portrait", (400, 400), (400, 400), (480, 640)),
                                           ("ragged", (93, 70), (375, 281), (641, 479))):
        kps = (rs.uniform(-5, 5, (24, 2)) + rs.uniform(0, 1, (24, 2)) * np.array(out_res)).astype(np.float32)
        kps[3] = kps[17] = np.float32(-999.999)
        kps[5] = (0.0, 0.0)
        out[name] = (kps, out_res, in_res, raw_res)
    return out
