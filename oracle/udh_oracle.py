"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of the step between the UDH network and
HSIC.forward's `h_matrix` argument.

  udh/udh/model.py:100-111   corners_hat = corners + delta; h = kornia.get_perspective_transform(corners, corners_hat);
                             h_inv = torch.inverse(h)
  newtrain_codec_real.py:49-59, :129   h_adjust: four in-place row / column scalings from the patch frame to the picture

PARITY UNPINNED: kornia is a pip dependency of the reference (requirements: kornia==0.5.0), absent from its tree and from
this image, and the reference holds no fixture for this step.  get_perspective_transform is restated from kornia's
published algorithm: the 8 x 8 direct-linear-transform system of the 4 correspondences with h33 = 1,
    [x y 1 0 0 0 -x u -y u] h = u,   [0 0 0 x y 1 -x v -y v] h = v     for (x, y) -> (u, v),
solved exactly; float64 here."""
import numpy as np


def get_perspective_transform(src, dst):
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    out = np.empty((src.shape[0], 3, 3))
    for n in range(src.shape[0]):
        A, b = np.zeros((8, 8)), np.zeros(8)
        for k in range(4):
            (x, y), (u, v) = src[n, k], dst[n, k]
            A[2 * k] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
            A[2 * k + 1] = [0, 0, 0, x, y, 1, -x * v, -y * v]
            b[2 * k], b[2 * k + 1] = u, v
        out[n] = np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)
    return out


def h_adjust(ori_h, ori_w, patch_h, patch_w, h):
    """newtrain_codec_real.py:49-59, the in-place order kept"""
    h = np.array(h, dtype=np.float64)
    a, b = ori_h / patch_h, ori_w / patch_w
    h[:, 0, :] = a * h[:, 0, :]
    h[:, :, 0] = (1.0 / a) * h[:, :, 0]
    h[:, 1, :] = b * h[:, 1, :]
    h[:, :, 1] = (1.0 / b) * h[:, :, 1]
    return h


def h_matrix_from_corners(corners, delta, ori_hw, patch_hw):
    corners = np.asarray(corners, dtype=np.float32)
    hat = (corners + np.asarray(delta, dtype=np.float32)).astype(np.float64)       # formed in float32 by the reference
    h = get_perspective_transform(corners.astype(np.float64), hat)
    return h_adjust(ori_hw[0], ori_hw[1], patch_hw[0], patch_hw[1], np.linalg.inv(h))
