"""Host-side mirror of the parts of /root/reference/dream/image_proc.py that sit on the hot path or
directly beside it: ``peaks_from_belief_maps`` (image_proc.py:914-1018, runs on the GPU here), the
resolution arithmetic DreamNetwork needs (image_proc.py:94-133, 291-351) and the keypoint frame
conversions used by ``keypoints_from_image`` (image_proc.py:135-260)."""
import numpy as np
import torch
from PIL import Image as PILImage

from . import _hip, ops

KNOWN_IMAGE_PREPROC_TYPES = ["none", "resize", "shrink", "shrink-and-crop"]

# scipy.ndimage._filters._gaussian_kernel1d(sigma=3, order=0, radius=12)[0:13] as float64 hex: the 13
# distinct taps hard-coded in csrc/peaks.hip (index 12 = centre).  tests/test_oracle_peaks.py checks
# that the installed scipy/numpy would compute exactly these.
GAUSS_SIGMA3_HALF_TAPS_HEX = (
    "0x1.763a210dfb306p-15", "0x1.4fbe39149e277p-13", "0x1.0d8a5ad43c165p-11", "0x1.8345966f69518p-10",
    "0x1.f1e9915139406p-9", "0x1.1e6bccad344bap-7", "0x1.26defcaeb0202p-6", "0x1.0fa58939b528fp-5",
    "0x1.bfde9c12bec92p-5", "0x1.4a614d1afd337p-4", "0x1.b42a57d56c0bep-4", "0x1.01a25f86eb137p-3",
    "0x1.105a329f98197p-3")


def peaks_from_belief_maps(belief_map_tensor, offset_due_to_upsampling):
    """[N,H,W] tensor -> list (len N) of lists of (x, y, score, id), exactly the reference's return
    value (float64 centroids, float32 scores, running ids), computed by the HIP peak kernels."""
    assert (
        len(belief_map_tensor.shape) == 3
    ), "Expected belief_map_tensor to have shape [N x height x width], but it is {}.".format(belief_map_tensor.shape)
    maps = _hip.device_tensor(belief_map_tensor.detach())
    xy, score, counts = ops.peaks_list(maps.float(), offset_due_to_upsampling)
    xy, score, counts = xy.cpu().numpy(), score.cpu().numpy(), counts.cpu().numpy()
    all_peaks, counter = [], 0
    for j in range(maps.shape[0]):
        n = int(counts[j])
        all_peaks.append([(xy[j, i, 0], xy[j, i, 1], score[j, i], counter + i) for i in range(n)])
        counter += n
    return all_peaks


# ---- resolution arithmetic (image_proc.py:94-133, 300-351) ---------------------------------------------
def shrink_resolution(image_input_resolution, image_ref_resolution):
    factor = float(image_ref_resolution[1]) / float(image_input_resolution[1])
    return (int(image_input_resolution[0] * factor), image_ref_resolution[1])


def shrink_and_crop_resolution(image_input_resolution, image_ref_resolution):
    """-> ((cropped width, cropped height), (crop x0, crop y0)) in input pixels: the largest centred
    window of the input with the reference aspect ratio (same int() truncations as
    image_proc.py:317-351, so (640,480) vs (400,400) gives ((480,480),(80,0)))."""
    in_w, in_h = image_input_resolution
    ref_w, ref_h = image_ref_resolution
    h_from_w = int(float(in_w) / float(ref_w) * ref_h)
    w_from_h = int(float(in_h) / float(ref_h) * ref_w)
    if in_w >= w_from_h:
        cropped = (w_from_h, in_h)
    else:
        assert in_h >= h_from_w
        cropped = (in_w, h_from_w)
    return cropped, ((in_w - cropped[0]) // 2, (in_h - cropped[1]) // 2)


def resolution_after_preprocessing(image_input_resolution, image_ref_resolution, image_preprocessing):
    assert (
        image_preprocessing in KNOWN_IMAGE_PREPROC_TYPES
    ), 'Image preprocessing type "{}" is not recognized.'.format(image_preprocessing)
    if image_preprocessing == "none":
        return image_input_resolution
    if image_preprocessing == "shrink":
        return shrink_resolution(image_input_resolution, image_ref_resolution)
    return image_ref_resolution              # resize, shrink-and-crop


def preprocess_image(input_image, image_ref_resolution, image_preprocessing):
    assert isinstance(input_image, PILImage.Image), 'Expected "input_image" to be a PIL Image, but it is "{}".'.format(
        type(input_image))
    assert image_preprocessing in KNOWN_IMAGE_PREPROC_TYPES, 'Image preprocessing type "{}" is not recognized.'.format(
        image_preprocessing)
    if image_preprocessing == "none":
        return input_image
    if image_preprocessing == "resize":
        return input_image.resize(image_ref_resolution, resample=PILImage.BILINEAR)
    if image_preprocessing == "shrink":
        return input_image.resize(shrink_resolution(input_image.size, image_ref_resolution), resample=PILImage.BILINEAR)
    (cw, ch), (x0, y0) = shrink_and_crop_resolution(input_image.size, image_ref_resolution)
    return input_image.crop((x0, y0, x0 + cw, y0 + ch)).resize(image_ref_resolution, resample=PILImage.BILINEAR)


# ---- keypoint frame conversions (image_proc.py:135-147, 215-260) ----------------------------------------
def convert_keypoints_to_netin_from_netout(keypoints_netout, net_output_resolution, net_input_resolution):
    k = np.asarray(keypoints_netout, dtype=float).reshape(-1, 2)
    return np.stack([k[:, 0] / net_output_resolution[0] * net_input_resolution[0],
                     k[:, 1] / net_output_resolution[1] * net_input_resolution[1]], axis=1)


def convert_keypoints_to_raw_from_netin(keypoints_netin, net_input_resolution, image_raw_resolution,
                                        image_preprocessing):
    assert image_preprocessing in KNOWN_IMAGE_PREPROC_TYPES, 'Image preprocessing type "{}" is not recognized.'.format(
        image_preprocessing)
    k = np.asarray(keypoints_netin, dtype=float).reshape(-1, 2)
    if image_preprocessing == "none":
        return k
    if image_preprocessing in ("resize", "shrink"):
        return np.stack([k[:, 0] / net_input_resolution[0] * image_raw_resolution[0],
                         k[:, 1] / net_input_resolution[1] * image_raw_resolution[1]], axis=1)
    (cw, ch), (x0, y0) = shrink_and_crop_resolution(image_raw_resolution, net_input_resolution)
    return np.stack([k[:, 0] / net_input_resolution[0] * cw + x0,
                     k[:, 1] / net_input_resolution[1] * ch + y0], axis=1)


def convert_keypoints_batch(keypoints_netout, net_output_resolution, net_input_resolution, image_raw_resolution,
                            image_preprocessing):
    """The two conversions above for a whole batch on the device: keypoints [..., 2] fp32 in the
    net-output frame (as DreamNetwork.inference computes them) -> (netin, raw) float64 tensors of the same shape, bit
    for bit what the per-keypoint Python loops of the reference produce (dream/analysis.py:219-232)."""
    assert image_preprocessing in KNOWN_IMAGE_PREPROC_TYPES, 'Image preprocessing type "{}" is not recognized.'.format(
        image_preprocessing)
    k = _hip.device_tensor(torch.as_tensor(keypoints_netout, dtype=torch.float32)).contiguous()
    n = k.numel() // 2
    netin = torch.empty(tuple(k.shape), dtype=torch.float64, device=k.device)
    raw = torch.empty_like(netin)
    (ow, oh), (iw, ih) = net_output_resolution, net_input_resolution
    if image_preprocessing == "none":
        mode, span, origin = 0, (1.0, 1.0), (0.0, 0.0)
    elif image_preprocessing in ("resize", "shrink"):
        mode, span, origin = 1, image_raw_resolution, (0.0, 0.0)
    else:
        span, origin = shrink_and_crop_resolution(image_raw_resolution, net_input_resolution)
        mode = 1
    _hip.call("dream_convert_keypoints_f64", ops.ptr(k), ops.ptr(netin), ops.ptr(raw), n, float(ow), float(oh), float(iw),
              float(ih), float(span[0]), float(span[1]), float(origin[0]), float(origin[1]), mode, ops.stream())
    return netin, raw


# ---- the steps right before the hot path, on the device ---------------------------------------
def normalize_images_u8(images_u8_bhwc, mean, stdev):
    """uint8 RGB frames [B,H,W,3] (device) -> normalised fp32 [B,3,H,W]: ToTensor + Normalize(mean, stdev) as the
    dataset does (dream/datasets.py:87-94), bit-identical to the torchvision transforms, without the 4x larger fp32
    upload."""
    import ctypes
    x = _hip.device_tensor(images_u8_bhwc)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[3] == 3, "expected uint8 [B,H,W,3]"
    x = x.contiguous()
    b, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in stdev])
    _hip.call("dream_normalize_u8_hwc_to_chw_f32", ops.ptr(x), ops.ptr(out), b, h, w, m, s, ops.stream())
    return out


def create_belief_map(image_resolution, pointsBelief, sigma=2):
    """Drop-in for dream/image_proc.py:866-910 (training targets, called per frame by dream/datasets.py:165-171):
    -> numpy float64 [n_points, H, W].  Rendered on the device by the batched kernel; every value is the fp32 rounding
    of the reference's float64 value, which is exactly what its only caller keeps (``torch.tensor(maps).float()``)."""
    assert len(image_resolution) == 2, \
        'Expected "image_resolution" to have length 2, but it has length {}.'.format(len(image_resolution))
    pts = np.asarray([[float(p[0]), float(p[1])] for p in pointsBelief], dtype=np.float64).reshape(-1, 2)
    if pts.shape[0] == 0:
        return np.zeros((0, int(image_resolution[1]), int(image_resolution[0])))
    maps = create_belief_map_batch(image_resolution, torch.from_numpy(pts)[None], sigma)
    return maps[0].cpu().numpy().astype(np.float64)


def create_belief_map_batch(image_resolution, keypoints_bk2, sigma=2):
    """Batched, on-device create_belief_map (dream/image_proc.py:866-910): keypoints [B,K,2] (x, y) in the belief-map
    frame -> fp32 [B,K,H,W], bit-identical to torch.tensor(create_belief_map(res, kps)).float() per frame."""
    assert len(image_resolution) == 2, \
        'Expected "image_resolution" to have length 2, but it has length {}.'.format(len(image_resolution))
    width, height = int(image_resolution[0]), int(image_resolution[1])
    # float64 all the way: the reference truncates the float64 coordinate with int() (image_proc.py:889-890)
    kps = _hip.device_tensor(torch.as_tensor(keypoints_bk2).to(torch.float64)).contiguous()
    b, k = int(kps.shape[0]), int(kps.shape[1])
    w = int(sigma * 2)
    dy, dx = np.mgrid[-w:w + 1, -w:w + 1]
    blob64 = np.exp(-((dx ** 2 + dy ** 2) / (2 * (sigma ** 2))))           # float64, as the reference computes it
    blob = torch.from_numpy(blob64.astype(np.float32)).to(kps.device)        # the .float() cast of the reference
    out = torch.empty((b, k, height, width), dtype=torch.float32, device=kps.device)
    _hip.call("dream_create_belief_maps_f64kps_f32", ops.ptr(kps), ops.ptr(blob), ops.ptr(out), b * k, height, width, w, ops.stream())
    return out
