// warp.hip -- perspective warp of the MASIC hot path (gfx950).
//
// Reference call sites: coremasic/mywork/MASIC.py:638,644 (mask), 781 (x1), 821/833 (x1_hat):
// kornia==0.5.0 `warp_perspective(src, M, dsize)` with defaults (bilinear, zeros, align_corners
// -> True).  kornia is a pip dependency that is not vendored in the reference; its algorithm
// (normalize_homography -> inverse -> create_meshgrid(normalized) -> transform_points ->
// convert_points_from_homogeneous(eps=1e-8) -> F.grid_sample) is restated here and in
// oracle/hsic_oracle.py:warp_perspective with the same float32 operation order for the sampling
// coordinates (explicit __fmul_rn/__fadd_rn: no FMA contraction), because sub-pixel coordinate
// rounding dominates the error budget of this op.
//
// HBM-bound gather: one thread per destination pixel computes the source location once and
// samples every channel (4 taps each); destination stores are coalesced along W.
#include "common.h"
#include <atomic>

namespace {

// M (pixel space, dst <- src) -> inverse(N_dst . M . N_src^-1), float64 internally.
__global__ void warp_matrix_kernel(const float* __restrict__ M, float* __restrict__ out, int B,
                                   int Hs, int Ws, int Hd, int Wd, int invert_first) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double m[9];
    for (int i = 0; i < 9; ++i) m[i] = (double)M[b * 9 + i];
    auto inv3 = [](const double* a, double* r) {
        const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
        const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
        const double id = 1.0 / det;
        r[0] = c00 * id; r[1] = (a[2] * a[7] - a[1] * a[8]) * id; r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
        r[3] = c01 * id; r[4] = (a[0] * a[8] - a[2] * a[6]) * id; r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
        r[6] = c02 * id; r[7] = (a[1] * a[6] - a[0] * a[7]) * id; r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    };
    if (invert_first) { double t[9]; inv3(m, t); for (int i = 0; i < 9; ++i) m[i] = t[i]; }
    // N(h,w) = [[2/(w-1),0,-1],[0,2/(h-1),-1],[0,0,1]];  N^-1 = [[(w-1)/2,0,(w-1)/2],[0,(h-1)/2,(h-1)/2],[0,0,1]]
    const double sxs = Ws == 1 ? 1e-14 : (double)(Ws - 1), sys = Hs == 1 ? 1e-14 : (double)(Hs - 1);
    const double sxd = Wd == 1 ? 1e-14 : (double)(Wd - 1), syd = Hd == 1 ? 1e-14 : (double)(Hd - 1);
    double t[9];   // M . N_src^-1
    for (int r = 0; r < 3; ++r) {
        t[3 * r + 0] = m[3 * r + 0] * (sxs / 2.0);
        t[3 * r + 1] = m[3 * r + 1] * (sys / 2.0);
        t[3 * r + 2] = m[3 * r + 0] * (sxs / 2.0) + m[3 * r + 1] * (sys / 2.0) + m[3 * r + 2];
    }
    double u[9];   // N_dst . t
    for (int c = 0; c < 3; ++c) {
        u[0 + c] = (2.0 / sxd) * t[0 + c] - t[6 + c];
        u[3 + c] = (2.0 / syd) * t[3 + c] - t[6 + c];
        u[6 + c] = t[6 + c];
    }
    double r9[9];
    inv3(u, r9);
    for (int i = 0; i < 9; ++i) out[b * 9 + i] = (float)r9[i];
}

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, const float* __restrict__ minv,
                                                   float* __restrict__ dst, int C, int Hs, int Ws, int Hd, int Wd,
                                                   int out_ctot, int out_coff, int align_corners) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hd * Wd) return;
    const int oy = pix / Wd, ox = pix - oy * Wd;
    const float* m = minv + b * 9;
    // create_meshgrid(normalized_coordinates=True): (i / (n-1) - 0.5) * 2
    const float gx = __fmul_rn(__fsub_rn(__fdiv_rn((float)ox, (float)(Wd - 1)), 0.5f), 2.0f);
    const float gy = __fmul_rn(__fsub_rn(__fdiv_rn((float)oy, (float)(Hd - 1)), 0.5f), 2.0f);
    const float X = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[0]), __fmul_rn(gy, m[1])), m[2]);
    const float Y = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[3]), __fmul_rn(gy, m[4])), m[5]);
    const float Z = __fadd_rn(__fadd_rn(__fmul_rn(gx, m[6]), __fmul_rn(gy, m[7])), m[8]);
    const float scale = fabsf(Z) > 1e-8f ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
    const float nx = __fmul_rn(X, scale), ny = __fmul_rn(Y, scale);
    const float fx = masic_grid_unnormalize(nx, Ws, align_corners);
    const float fy = masic_grid_unnormalize(ny, Hs, align_corners);
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = __fsub_rn(fx, x0f), wy = __fsub_rn(fy, y0f);
    const float ex = __fsub_rn(1.0f, wx), ey = __fsub_rn(1.0f, wy);
    const float w_nw = __fmul_rn(ex, ey), w_ne = __fmul_rn(wx, ey), w_sw = __fmul_rn(ex, wy), w_se = __fmul_rn(wx, wy);
    // float -> int with saturation so that far-away samples stay out of range instead of wrapping
    const float lim = 1.0e9f;
    const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < Ws, vx1 = x1 >= 0 && x1 < Ws;
    const bool vy0 = y0 >= 0 && y0 < Hs, vy1 = y1 >= 0 && y1 < Hs;
    const bool finite = (fx == fx) && (fy == fy);
    const size_t splane = (size_t)Hs * Ws, dplane = (size_t)Hd * Wd;
    float* d = dst + ((size_t)b * out_ctot + out_coff) * dplane + pix;
    if (src == nullptr) {   // warp of an all-ones image (MASIC.py:636-638): the sum of the in-range weights
        float v = 0.0f;
        if (finite) {
            v = (vx0 && vy0) ? w_nw : 0.0f;
            v = __fadd_rn(v, (vx1 && vy0) ? w_ne : 0.0f);
            v = __fadd_rn(v, (vx0 && vy1) ? w_sw : 0.0f);
            v = __fadd_rn(v, (vx1 && vy1) ? w_se : 0.0f);
        }
        d[0] = v;
        return;
    }
    const float* s = src + (size_t)b * C * splane;
    for (int c = 0; c < C; ++c) {
        const float* sc = s + (size_t)c * splane;
        float v = 0.0f;
        if (finite) {
            const float nw = (vx0 && vy0) ? sc[(size_t)y0 * Ws + x0] : 0.0f;
            const float ne = (vx1 && vy0) ? sc[(size_t)y0 * Ws + x1] : 0.0f;
            const float sw = (vx0 && vy1) ? sc[(size_t)y1 * Ws + x0] : 0.0f;
            const float se = (vx1 && vy1) ? sc[(size_t)y1 * Ws + x1] : 0.0f;
            v = __fmul_rn(nw, w_nw);
            v = __fadd_rn(v, __fmul_rn(ne, w_ne));
            v = __fadd_rn(v, __fmul_rn(sw, w_sw));
            v = __fadd_rn(v, __fmul_rn(se, w_se));
        }
        d[(size_t)c * dplane] = v;
    }
}

// ---- upstream of the path: the homography of a stereo pair from the 4 corner offsets the UDH network
// predicts -- udh/udh/model.py:100-111 (kornia.get_perspective_transform(corners, corners + delta), torch.inverse) followed by
// h_adjust (newtrain_codec_real.py:49-59: rescale from the 128 x 128 patch frame to the picture).  kornia builds the 8 x 8
// direct-linear-transform system of the 4 correspondences and solves it with an LU solve (h33 = 1); here one lane per pair does
// the same elimination with partial pivoting in float64, inverts the 3 x 3 by cofactors and applies h_adjust's four in-place
// scalings in the reference's order and float32.  kornia is not part of the reference tree: parity is pinned by a float64
// restatement and by the defining property H [corner, 1] ~ [corner + delta, 1].
__global__ __launch_bounds__(64) void homography_from_corners_kernel(const float* __restrict__ corners, const float* __restrict__ delta,
                                                                     float* __restrict__ h_out, int B, float a, float b) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B) return;
    double A[8][9];
    for (int k = 0; k < 4; ++k) {
        const double px = corners[(i * 4 + k) * 2], py = corners[(i * 4 + k) * 2 + 1];
        const double qx = (double)(corners[(i * 4 + k) * 2] + delta[(i * 4 + k) * 2]);            // corners_hat is formed in float32
        const double qy = (double)(corners[(i * 4 + k) * 2 + 1] + delta[(i * 4 + k) * 2 + 1]);
        double* rx = A[2 * k];
        double* ry = A[2 * k + 1];
        rx[0] = px; rx[1] = py; rx[2] = 1; rx[3] = 0; rx[4] = 0; rx[5] = 0; rx[6] = -px * qx; rx[7] = -py * qx; rx[8] = qx;
        ry[0] = 0; ry[1] = 0; ry[2] = 0; ry[3] = px; ry[4] = py; ry[5] = 1; ry[6] = -px * qy; ry[7] = -py * qy; ry[8] = qy;
    }
    bool singular = false;
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        for (int r = c + 1; r < 8; ++r)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0) { singular = true; break; }
        if (piv != c)
            for (int k = 0; k < 9; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        for (int r = c + 1; r < 8; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
        }
    }
    double x[9];
    x[8] = 1.0;
    for (int c = 7; c >= 0 && !singular; --c) {
        double s = A[c][8];
        for (int k = c + 1; k < 8; ++k) s -= A[c][k] * x[k];
        x[c] = s / A[c][c];
    }
    // inverse of H = [x0 x1 x2; x3 x4 x5; x6 x7 1]
    const double c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
    const double det = x[0] * c00 + x[1] * c01 + x[2] * c02;
    float h[9];
    if (singular || det == 0.0) {
        for (int k = 0; k < 9; ++k) h[k] = __builtin_nanf("");
    } else {
        const double r = 1.0 / det;
        h[0] = (float)(c00 * r); h[1] = (float)((x[2] * x[7] - x[1] * x[8]) * r); h[2] = (float)((x[1] * x[5] - x[2] * x[4]) * r);
        h[3] = (float)(c01 * r); h[4] = (float)((x[0] * x[8] - x[2] * x[6]) * r); h[5] = (float)((x[2] * x[3] - x[0] * x[5]) * r);
        h[6] = (float)(c02 * r); h[7] = (float)((x[1] * x[6] - x[0] * x[7]) * r); h[8] = (float)((x[0] * x[4] - x[1] * x[3]) * r);
    }
    const float ia = 1.0f / a, ib = 1.0f / b;
    for (int k = 0; k < 3; ++k) h[k] = a * h[k];                   // h[:, 0, :] *= a
    for (int k = 0; k < 3; ++k) h[3 * k] = ia * h[3 * k];          // h[:, :, 0] *= 1/a
    for (int k = 0; k < 3; ++k) h[3 + k] = b * h[3 + k];           // h[:, 1, :] *= b
    for (int k = 0; k < 3; ++k) h[3 * k + 1] = ib * h[3 * k + 1];  // h[:, :, 1] *= 1/b
    for (int k = 0; k < 9; ++k) h_out[(size_t)i * 9 + k] = h[k];
}

}  // namespace

extern "C" int masic_homography_from_corners(const float* corners, const float* delta, float* h_out, int B, float scale_a, float scale_b,
                                             void* stream) {
    MASIC_REQUIRE(corners && delta && h_out, MASIC_ERR_ARG, "homography_from_corners: null pointer");
    MASIC_REQUIRE(B > 0 && scale_a > 0.0f && scale_b > 0.0f, MASIC_ERR_SHAPE, "homography_from_corners: B=%d, scales %g %g", B, (double)scale_a, (double)scale_b);
    hipLaunchKernelGGL(homography_from_corners_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, corners, delta, h_out, B,
                       scale_a, scale_b);
    return masic_launch_status("homography_from_corners");
}

extern "C" int masic_warp_matrix(const float* M, float* minv_norm, int B, int Hs, int Ws, int Hd, int Wd,
                                 int invert_first, void* stream) {
    MASIC_REQUIRE(M && minv_norm, MASIC_ERR_ARG, "warp_matrix: null pointer");
    MASIC_REQUIRE(B > 0, MASIC_ERR_SHAPE, "warp_matrix: B=%d", B);
    hipLaunchKernelGGL(warp_matrix_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, M, minv_norm, B,
                       Hs, Ws, Hd, Wd, invert_first);
    return masic_launch_status("warp_matrix");
}

extern "C" int masic_warp_perspective_fwd(const float* src, const float* minv_norm, float* dst,
                                          int B, int C, int Hs, int Ws, int Hd, int Wd,
                                          int out_ctot, int out_coff, void* stream) {
    MASIC_REQUIRE(minv_norm && dst, MASIC_ERR_ARG, "warp_perspective_fwd: null pointer");
    MASIC_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, MASIC_ERR_SHAPE, "warp_perspective_fwd: bad shape");
    MASIC_REQUIRE(src != nullptr || C == 1, MASIC_ERR_ARG, "warp_perspective_fwd: ones-source needs C == 1");
    MASIC_REQUIRE(out_coff >= 0 && out_coff + C <= out_ctot, MASIC_ERR_SHAPE, "warp_perspective_fwd: output view out of range");
    hipLaunchKernelGGL(warp_kernel, dim3(ceil_div(Hd * Wd, 256), B), dim3(256), 0, (hipStream_t)stream, src, minv_norm,
                       dst, C, Hs, Ws, Hd, Wd, out_ctot, out_coff, masic_warp_align_corners_value());
    return masic_launch_status("warp_perspective_fwd");
}

static std::atomic<int> g_warp_align_corners{1};
int masic_warp_align_corners_value() { return g_warp_align_corners.load(std::memory_order_relaxed); }
// grid_sample convention of every warp of the library (forward, backward, the F16K gated warp): 1 = align_corners=True (kornia 0.5.0,
// the default), 0 = align_corners=False (kornia <= 0.4.1).  Process-wide; set it once before the first forward.
extern "C" void masic_set_warp_align_corners(int align_corners) { g_warp_align_corners.store(align_corners ? 1 : 0, std::memory_order_relaxed); }
extern "C" int masic_get_warp_align_corners(void) { return masic_warp_align_corners_value(); }
