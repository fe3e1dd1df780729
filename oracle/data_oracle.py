"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of the per-item preparation of the reference's
paired dataset, compressai/datasets/utils.py:

  :207-212  paired crop  img[startH:startH+ph, startW:startW+pw]  of the RGB uint8 pictures
  :215-220  homo_img = cv2.resize(crop, (256, 256)) -> ToTensor -> Normalize(MEAN, STD) (:26-27, :125-131: the channel MEANS
            of the ImageNet constants, one scalar each) -> mean over the 3 channels
  :260-261  patch  homo_img[:, y:y+128, x:x+128]
  :276-285  transform(img) = ToTensor: uint8 HWC -> float32 CHW / 255

PARITY UNPINNED: cv2 (opencv-python, unpinned in the reference's requirements) is neither in the reference tree nor in this
image and the reference holds no fixture of this step.  cv2.resize for uint8 / INTER_LINEAR is restated from OpenCV's
published algorithm (modules/imgproc/src/resize.cpp): pixel-centre mapping fx = (dx + .5) * scale - .5, 11-bit fixed-point
coefficients cvRound(w * 2048), horizontal pass in int32, vertical pass ((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16),
+ 2, >> 2; and its special case: an exact 2x2 decimation is done as INTER_AREA (rounded 2 x 2 box mean)."""
import numpy as np

MEAN = np.float32(np.float32([0.485, 0.456, 0.406]).mean())
STD = np.float32(np.float32([0.229, 0.224, 0.225]).mean())


def _coeffs(dst, src):
    scale = src / dst
    ofs = np.empty(dst, dtype=np.int64)
    co = np.empty((dst, 2), dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - s)
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        ofs[d] = s
        co[d, 0] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(2048)))      # cvRound: half to even
        co[d, 1] = int(np.rint(f * np.float32(2048)))
    return ofs, co


def cv2_resize_linear_u8(img, dsize):
    """img [H, W, C] uint8, dsize = (width, height) as in cv2.resize"""
    H, W, C = img.shape
    dw, dh = dsize
    src = img.astype(np.int64)
    if H == 2 * dh and W == 2 * dw:                       # INTER_LINEAR of an exact 2x decimation runs as INTER_AREA
        s = src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2]
        return ((s + 2) >> 2).astype(np.uint8)
    xo, xa = _coeffs(dw, W)
    yo, ya = _coeffs(dh, H)
    x1 = np.minimum(xo + 1, W - 1)
    rows = src[:, xo] * xa[:, 0][None, :, None] + src[:, x1] * xa[:, 1][None, :, None]          # [H, dw, C]
    y1 = np.minimum(yo + 1, H - 1)
    t0, t1 = rows[yo], rows[y1]
    out = (((ya[:, 0][:, None, None] * (t0 >> 4)) >> 16) + ((ya[:, 1][:, None, None] * (t1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare_view(img, start_h, start_w, ph, pw, homopic=256, x=0, y=0, homopatch=128):
    """-> (float32 [3, ph, pw] picture crop in [0, 1], float32 [1, homopatch, homopatch] grey patch for the homography net)"""
    crop = img[start_h:start_h + ph, start_w:start_w + pw]
    pic = (crop.astype(np.float32) / np.float32(255)).transpose(2, 0, 1).copy()
    small = cv2_resize_linear_u8(crop, (homopic, homopic)).astype(np.float32) / np.float32(255)
    norm = (small - MEAN) / STD
    grey = (norm[..., 0] + norm[..., 1] + norm[..., 2]) / np.float32(3)
    return pic, grey[None, y:y + homopatch, x:x + homopatch].copy()
