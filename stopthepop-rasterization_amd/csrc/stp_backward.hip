// stp_backward.hip -- backward of the per-Gaussian stages.
//
// Replaces (reference cuda_rasterizer/backward.cu):
//   computeCov2DCUDA              :146-312  (dL/dconic -> dL/dcov3D, covariance path of dL/dmean3D, optional
//                                            Mip-Splatting opacity-scaling gradient)
//   preprocessCUDA<3> (backward)  :384-434  (projection path of dL/dmean3D)
//   computeColorFromSH (backward) :22-141
//   computeCov3D (backward)       :316-379
// The reference runs two kernels back to back; both are per-Gaussian with no cross-thread dependency,
// so here they are one HBM-streaming kernel: each Gaussian's inputs are read once and its nine gradient
// rows written once.
#include "stp_internal.h"
#include "stp_device.h"

namespace stp {

namespace {

struct BwdPreArgs {
    int P, D, M, proper_ewa_scaling;
    float h_x, h_y, tan_fovx, tan_fovy, scale_modifier;
    const float* means3D;
    const int* radii;
    const float* shs;
    const uint8_t* clamped;
    const float* opacities;
    const float* scales;
    const float* rotations;
    const float* cov3Ds;
    const float* view;
    const float* proj;
    const float* cam;
    const float* grad_rec; // P x grad_stride: sums of the render half (layout: stp_raster.h, stp_backward)
    int grad_stride;       // 16 (one line per Gaussian, two 16-byte loads) or 9 (compact records of a tile-row shard)
    int clear_rec;         // leave every record read zero-filled again (phases bit 3): the caller's buffer is ready for the next backward
    int block0;            // first 256-Gaussian block of this launch (the grid covers a range of blocks: stp_backward_phases, chunked per-Gaussian half)
    float* dL_dmean2D;     // P x 3  (out)
    float* dL_dopacity;    // P      (out)
    float* dL_dcolor;      // P x 3  (out)
    float* dL_dmean3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscale;
    float* dL_drot;
};

// Everything for one visible Gaussian.  sh_row / dsh_row: this Gaussian's 3M SH coefficients and their gradient,
// both in the SAME LDS row (the kernel stages them, see below): each channel reads what it needs before it writes.
__device__ __forceinline__ void gaussian_backward(const BwdPreArgs& a, int idx, float* sh_row)
{
    const float* __restrict__ view = a.view;
    const float* __restrict__ proj = a.proj;
    const float3 mean = make_float3(a.means3D[3 * (size_t)idx], a.means3D[3 * (size_t)idx + 1], a.means3D[3 * (size_t)idx + 2]);
    float3 dmean;
    // the render half's sums: one 64-byte record, two 16-byte loads and a scalar
    float4 rec0, rec1; // colour r g b, mean2D x | mean2D y, conic xx xy yy
    float rec_op;
    if (a.grad_stride == 16) {
        rec0 = *reinterpret_cast<const float4*>(a.grad_rec + 16 * (size_t)idx);
        rec1 = *reinterpret_cast<const float4*>(a.grad_rec + 16 * (size_t)idx + 4);
        rec_op = a.grad_rec[16 * (size_t)idx + 8];
    } else { // compact records: 36-byte stride, scalar loads
        const float* __restrict__ r = a.grad_rec + (size_t)a.grad_stride * idx;
        rec0 = make_float4(r[0], r[1], r[2], r[3]);
        rec1 = make_float4(r[4], r[5], r[6], r[7]);
        rec_op = r[8];
    }
    if (a.clear_rec) { // only records of visible Gaussians are ever written by the render half, and all of those pass through here
        float* const r = const_cast<float*>(a.grad_rec) + (size_t)a.grad_stride * idx;
        if (a.grad_stride == 16) {
            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            reinterpret_cast<float4*>(r)[0] = z; reinterpret_cast<float4*>(r)[1] = z; reinterpret_cast<float4*>(r)[2] = z;
        } else {
#pragma unroll
            for (int k = 0; k < 9; k++) r[k] = 0.0f;
        }
    }
    a.dL_dcolor[3 * (size_t)idx] = rec0.x; a.dL_dcolor[3 * (size_t)idx + 1] = rec0.y; a.dL_dcolor[3 * (size_t)idx + 2] = rec0.z;
    a.dL_dmean2D[3 * (size_t)idx] = rec0.w; a.dL_dmean2D[3 * (size_t)idx + 1] = rec1.x;

    // ---- dL/dconic -> dL/dcov2D -> dL/dcov3D and dL/dmean (covariance path) ----
    {
        const float* cov3D = a.cov3Ds + 6 * (size_t)idx;
        const float dcx = rec1.y, dcy = rec1.z, dcz = rec1.w;
        float3 t;
        t.x = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12];
        t.y = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13];
        t.z = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float txtz = t.x / t.z, tytz = t.y / t.z;
        t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
        t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
        Mat3 J;
        J.m[0][0] = a.h_x / t.z; J.m[0][1] = 0.0f;        J.m[0][2] = -(a.h_x * t.x) / (t.z * t.z);
        J.m[1][0] = 0.0f;        J.m[1][1] = a.h_y / t.z; J.m[1][2] = -(a.h_y * t.y) / (t.z * t.z);
        J.m[2][0] = 0.0f;        J.m[2][1] = 0.0f;        J.m[2][2] = 0.0f;
        Mat3 Wm;
        Wm.m[0][0] = view[0]; Wm.m[0][1] = view[4]; Wm.m[0][2] = view[8];
        Wm.m[1][0] = view[1]; Wm.m[1][1] = view[5]; Wm.m[1][2] = view[9];
        Wm.m[2][0] = view[2]; Wm.m[2][1] = view[6]; Wm.m[2][2] = view[10];
        Mat3 Vrk;
        Vrk.m[0][0] = cov3D[0]; Vrk.m[0][1] = cov3D[1]; Vrk.m[0][2] = cov3D[2];
        Vrk.m[1][0] = cov3D[1]; Vrk.m[1][1] = cov3D[3]; Vrk.m[1][2] = cov3D[4];
        Vrk.m[2][0] = cov3D[2]; Vrk.m[2][1] = cov3D[4]; Vrk.m[2][2] = cov3D[5];
        const Mat3 T = mat_mul(Wm, J);
        const Mat3 cov2D = mat_mul(mat_mul(mat_transpose(T), mat_transpose(Vrk)), T);
        float c_xx = cov2D.m[0][0], c_xy = cov2D.m[0][1], c_yy = cov2D.m[1][1];
        const float det_cov_orig = c_xx * c_yy - c_xy * c_xy;
        const float h_var = 0.3f;
        c_xx += h_var; c_yy += h_var;
        float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
        if (a.proper_ewa_scaling) {
            // Mip-Splatting opacity scaling (reference backward.cu:214-238).  As in the reference the closed
            // form below is evaluated with the dilated c_xx / c_yy.
            const float det_plus = c_xx * c_yy - c_xy * c_xy;
            const float h_scal = sqrtf(fmaxf(0.000025f, det_cov_orig / det_plus));
            const float dL_dop_v = rec_op;
            const float d_h_scal = dL_dop_v * a.opacities[idx];
            a.dL_dopacity[idx] = dL_dop_v * h_scal;
            const float d_inside_root = (det_cov_orig / det_plus) <= 0.000025f ? 0.f : d_h_scal / (2 * h_scal);
            const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
            const float qd = w * w + w * (x + y) + x * y - z * z;
            const float denom_f = d_inside_root / (qd * qd);
            dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
            dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
            dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
        } else {
            a.dL_dopacity[idx] = rec_op;
        }
        const float denom = c_xx * c_yy - c_xy * c_xy;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dcov[6];
        if (denom2inv != 0) {
            dL_dc_xx += denom2inv * (-c_yy * c_yy * dcx + 2 * c_xy * c_yy * dcy + (denom - c_xx * c_yy) * dcz);
            dL_dc_yy += denom2inv * (-c_xx * c_xx * dcz + 2 * c_xx * c_xy * dcy + (denom - c_xx * c_yy) * dcx);
            dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dcx - (denom + 2 * c_xy * c_xy) * dcy + c_xx * c_xy * dcz);
            dcov[0] = (T.m[0][0] * T.m[0][0] * dL_dc_xx + T.m[0][0] * T.m[1][0] * dL_dc_xy + T.m[1][0] * T.m[1][0] * dL_dc_yy);
            dcov[3] = (T.m[0][1] * T.m[0][1] * dL_dc_xx + T.m[0][1] * T.m[1][1] * dL_dc_xy + T.m[1][1] * T.m[1][1] * dL_dc_yy);
            dcov[5] = (T.m[0][2] * T.m[0][2] * dL_dc_xx + T.m[0][2] * T.m[1][2] * dL_dc_xy + T.m[1][2] * T.m[1][2] * dL_dc_yy);
            dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_dc_xx + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][1] * dL_dc_yy;
            dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_dc_xx + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_dc_xy + 2 * T.m[1][0] * T.m[1][2] * dL_dc_yy;
            dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_dc_xx + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_dc_xy + 2 * T.m[1][1] * T.m[1][2] * dL_dc_yy;
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];

        const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_xy;
        const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_xy;
        const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xx +
                              (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_xy;
        const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_dc_xy;
        const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_dc_xy;
        const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc_yy +
                              (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_dc_xy;
        const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
        const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
        const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
        const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
        const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dL_dtx = x_grad_mul * -a.h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -a.h_y * tz2 * dL_dJ12;
        const float dL_dtz = -a.h_x * tz2 * dL_dJ00 - a.h_y * tz2 * dL_dJ11 + (2 * a.h_x * t.x) * tz3 * dL_dJ02 + (2 * a.h_y * t.y) * tz3 * dL_dJ12;
        dmean.x = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
        dmean.y = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
        dmean.z = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
    }

    // ---- projection path of dL/dmean (reference backward.cu:408-425) ----
    {
        const float mhw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
        const float m_w = 1.0f / (mhw + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float gx = rec0.w, gy = rec1.x;
        dmean.x += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        dmean.y += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        dmean.z += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    }

    // ---- SH colour backward, including the view-direction dependence on the mean ----
    if (a.shs != nullptr) {
        const float3 cam = make_float3(a.cam[0], a.cam[1], a.cam[2]);
        const float3 v = make_float3(mean.x - cam.x, mean.y - cam.y, mean.z - cam.z);
        const float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        const float x = v.x / len, y = v.y / len, z = v.z / len;
        float* const dsh = sh_row; // in place: gradient row over coefficient row
        float dRGB[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = (ch == 0 ? rec0.x : ch == 1 ? rec0.y : rec0.z) * (a.clamped[3 * (size_t)idx + ch] ? 0.0f : 1.0f);
        float ddir[3] = {0, 0, 0};
        const int D = a.D;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float dx = 0, dy = 0, dz = 0;
            const float g = dRGB[ch];
            // the coefficients this channel's direction derivative needs, read before their slots are overwritten
            float sh[16];
#pragma unroll
            for (int k = 1; k < 16; k++) sh[k] = (k < a.M && k < (D + 1) * (D + 1)) ? sh_row[3 * k + ch] : 0.0f;
            for (int k = (D + 1) * (D + 1); k < a.M; k++) dsh[3 * k + ch] = 0.0f; // coefficients above the active degree
            dsh[ch] = kSH_C0 * g;
            if (D > 0) {
                dsh[3 + ch] = (-kSH_C1 * y) * g; dsh[6 + ch] = (kSH_C1 * z) * g; dsh[9 + ch] = (-kSH_C1 * x) * g;
                dx = -kSH_C1 * sh[3]; dy = -kSH_C1 * sh[1]; dz = kSH_C1 * sh[2];
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    dsh[12 + ch] = (kSH_C2[0] * xy) * g; dsh[15 + ch] = (kSH_C2[1] * yz) * g; dsh[18 + ch] = (kSH_C2[2] * (2.f * zz - xx - yy)) * g;
                    dsh[21 + ch] = (kSH_C2[3] * xz) * g; dsh[24 + ch] = (kSH_C2[4] * (xx - yy)) * g;
                    dx += kSH_C2[0] * y * sh[4] + kSH_C2[2] * 2.f * -x * sh[6] + kSH_C2[3] * z * sh[7] + kSH_C2[4] * 2.f * x * sh[8];
                    dy += kSH_C2[0] * x * sh[4] + kSH_C2[1] * z * sh[5] + kSH_C2[2] * 2.f * -y * sh[6] + kSH_C2[4] * 2.f * -y * sh[8];
                    dz += kSH_C2[1] * y * sh[5] + kSH_C2[2] * 2.f * 2.f * z * sh[6] + kSH_C2[3] * x * sh[7];
                    if (D > 2) {
                        dsh[27 + ch] = (kSH_C3[0] * y * (3.f * xx - yy)) * g; dsh[30 + ch] = (kSH_C3[1] * xy * z) * g;
                        dsh[33 + ch] = (kSH_C3[2] * y * (4.f * zz - xx - yy)) * g; dsh[36 + ch] = (kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                        dsh[39 + ch] = (kSH_C3[4] * x * (4.f * zz - xx - yy)) * g; dsh[42 + ch] = (kSH_C3[5] * z * (xx - yy)) * g;
                        dsh[45 + ch] = (kSH_C3[6] * x * (xx - 3.f * yy)) * g;
                        dx += (kSH_C3[0] * sh[9] * 3.f * 2.f * xy + kSH_C3[1] * sh[10] * yz + kSH_C3[2] * sh[11] * -2.f * xy +
                               kSH_C3[3] * sh[12] * -3.f * 2.f * xz + kSH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                               kSH_C3[5] * sh[14] * 2.f * xz + kSH_C3[6] * sh[15] * 3.f * (xx - yy));
                        dy += (kSH_C3[0] * sh[9] * 3.f * (xx - yy) + kSH_C3[1] * sh[10] * xz + kSH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) +
                               kSH_C3[3] * sh[12] * -3.f * 2.f * yz + kSH_C3[4] * sh[13] * -2.f * xy + kSH_C3[5] * sh[14] * -2.f * yz +
                               kSH_C3[6] * sh[15] * -3.f * 2.f * xy);
                        dz += (kSH_C3[1] * sh[10] * xy + kSH_C3[2] * sh[11] * 4.f * 2.f * yz + kSH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) +
                               kSH_C3[4] * sh[13] * 4.f * 2.f * xz + kSH_C3[5] * sh[14] * (xx - yy));
                    }
                }
            }
            ddir[0] += dx * g; ddir[1] += dy * g; ddir[2] += dz * g;
        }
        // derivative of the normalisation (reference auxiliary.h:179-189)
        const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean.x += ((+sum2 - v.x * v.x) * ddir[0] - v.y * v.x * ddir[1] - v.z * v.x * ddir[2]) * invsum32;
        dmean.y += (-v.x * v.y * ddir[0] + (sum2 - v.y * v.y) * ddir[1] - v.z * v.y * ddir[2]) * invsum32;
        dmean.z += (-v.x * v.z * ddir[0] - v.y * v.z * ddir[1] + (sum2 - v.z * v.z) * ddir[2]) * invsum32;
    }
    a.dL_dmean3D[3 * (size_t)idx + 0] = dmean.x;
    a.dL_dmean3D[3 * (size_t)idx + 1] = dmean.y;
    a.dL_dmean3D[3 * (size_t)idx + 2] = dmean.z;

    // ---- covariance -> scale / rotation ----
    if (a.scales != nullptr) {
        const float4 qv = reinterpret_cast<const float4*>(a.rotations)[idx];
        const float r = qv.x, x = qv.y, y = qv.z, z = qv.w;
        const Mat3 R = quat_to_mat(qv);
        const float sx = a.scale_modifier * a.scales[3 * (size_t)idx], sy = a.scale_modifier * a.scales[3 * (size_t)idx + 1],
                    sz = a.scale_modifier * a.scales[3 * (size_t)idx + 2];
        const Mat3 Mm = mat_mul(mat_diag(sx, sy, sz), R);
        const float* d = a.dL_dcov3D + 6 * (size_t)idx;
        Mat3 dSig;
        dSig.m[0][0] = d[0];        dSig.m[0][1] = 0.5f * d[1]; dSig.m[0][2] = 0.5f * d[2];
        dSig.m[1][0] = 0.5f * d[1]; dSig.m[1][1] = d[3];        dSig.m[1][2] = 0.5f * d[4];
        dSig.m[2][0] = 0.5f * d[2]; dSig.m[2][1] = 0.5f * d[4]; dSig.m[2][2] = d[5];
        Mat3 M2;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) M2.m[i][j] = 2.0f * Mm.m[i][j];
        const Mat3 dL_dM = mat_mul(M2, dSig);
        const Mat3 Rt = mat_transpose(R);
        Mat3 dMt = mat_transpose(dL_dM);
        a.dL_dscale[3 * (size_t)idx + 0] = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
        a.dL_dscale[3 * (size_t)idx + 1] = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
        a.dL_dscale[3 * (size_t)idx + 2] = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
#pragma unroll
        for (int j = 0; j < 3; j++) { dMt.m[0][j] *= sx; dMt.m[1][j] *= sy; dMt.m[2][j] *= sz; }
        float4 dq;
        dq.x = 2 * z * (dMt.m[0][1] - dMt.m[1][0]) + 2 * y * (dMt.m[2][0] - dMt.m[0][2]) + 2 * x * (dMt.m[1][2] - dMt.m[2][1]);
        dq.y = 2 * y * (dMt.m[1][0] + dMt.m[0][1]) + 2 * z * (dMt.m[2][0] + dMt.m[0][2]) + 2 * r * (dMt.m[1][2] - dMt.m[2][1]) - 4 * x * (dMt.m[2][2] + dMt.m[1][1]);
        dq.z = 2 * x * (dMt.m[1][0] + dMt.m[0][1]) + 2 * r * (dMt.m[2][0] - dMt.m[0][2]) + 2 * z * (dMt.m[1][2] + dMt.m[2][1]) - 4 * y * (dMt.m[2][2] + dMt.m[0][0]);
        dq.w = 2 * r * (dMt.m[0][1] - dMt.m[1][0]) + 2 * x * (dMt.m[2][0] + dMt.m[0][2]) + 2 * y * (dMt.m[1][2] + dMt.m[2][1]) - 4 * z * (dMt.m[1][1] + dMt.m[0][0]);
        reinterpret_cast<float4*>(a.dL_drot)[idx] = dq; // gradient w.r.t. the quaternion as given (not re-normalised)
    }
}

// One workgroup = 256 consecutive Gaussians.  Their SH coefficients (3M floats each, 192 B at degree 3) and the SH
// gradients are the bulk of this kernel's traffic, and a thread-per-Gaussian access to them is a 192-byte-stride
// gather/scatter (measured: 4x the compulsory HBM traffic).  The block therefore moves both through LDS: coalesced
// 16-byte loads of the whole 256 x 3M block, rows padded to 3M+1 words (conflict-free row access), the gradient
// written over the coefficients in place, coalesced stores back.  Every output row of every Gaussian is written
// (zeros for the invisible ones), so none of the dL_d* outputs needs a zero-fill by the caller.
__global__ void __launch_bounds__(256) preprocess_backward_kernel(const BwdPreArgs a)
{
    extern __shared__ float s_rows[]; // [256][3M + 1]
    const int tid = (int)threadIdx.x;
    const int base = ((int)blockIdx.x + a.block0) * 256;
    const int idx = base + tid;
    const int rows = min(256, a.P - base);
    const int row_len = 3 * a.M, row_stride = row_len + 1;
    const bool have_sh = a.shs != nullptr && a.M > 0;
    if (have_sh) {
        const float* __restrict__ src = a.shs + (size_t)base * row_len;
        const int total = rows * row_len;
        if (row_len == 48 && rows == 256) { // SH degree 3 (M = 16), full block: every load in flight at once (stp_device.h)
            stage_rows_in<256, 48>(src, s_rows, tid);
        } else if ((row_len & 3) == 0) {
            for (int f = 4 * tid; f < total; f += 4 * 256) {
                const float4 v = *reinterpret_cast<const float4*>(src + f);
                const int r = f / row_len, j = f - r * row_len;
                float* d = s_rows + r * row_stride + j;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int f = tid; f < total; f += 256) {
                const int r = f / row_len, j = f - r * row_len;
                s_rows[r * row_stride + j] = src[f];
            }
        }
        __syncthreads();
    }
    if (idx < a.P) {
        if (a.radii[idx] > 0) {
            gaussian_backward(a, idx, s_rows + tid * row_stride);
        } else { // invisible: all gradients are zero
            a.dL_dcolor[3 * (size_t)idx] = 0.0f; a.dL_dcolor[3 * (size_t)idx + 1] = 0.0f; a.dL_dcolor[3 * (size_t)idx + 2] = 0.0f;
            a.dL_dmean2D[3 * (size_t)idx] = 0.0f; a.dL_dmean2D[3 * (size_t)idx + 1] = 0.0f;
            a.dL_dopacity[idx] = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; i++) a.dL_dmean3D[3 * (size_t)idx + i] = 0.0f;
            if (a.scales != nullptr) {
#pragma unroll
                for (int i = 0; i < 3; i++) a.dL_dscale[3 * (size_t)idx + i] = 0.0f;
                reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (have_sh) for (int j = 0; j < row_len; j++) s_rows[tid * row_stride + j] = 0.0f;
        }
        a.dL_dmean2D[3 * (size_t)idx + 2] = 0.0f; // z component: never used by the reference either
    }
    if (have_sh) {
        __syncthreads();
        float* __restrict__ dst = a.dL_dsh + (size_t)base * row_len;
        const int total = rows * row_len;
        if (row_len == 48 && rows == 256) {
            stage_rows_out<256, 48>(dst, s_rows, tid);
        } else if ((row_len & 3) == 0) {
            for (int f = 4 * tid; f < total; f += 4 * 256) {
                const int r = f / row_len, j = f - r * row_len;
                const float* d = s_rows + r * row_stride + j;
                *reinterpret_cast<float4*>(dst + f) = make_float4(d[0], d[1], d[2], d[3]);
            }
        } else {
            for (int f = tid; f < total; f += 256) {
                const int r = f / row_len, j = f - r * row_len;
                dst[f] = s_rows[r * row_stride + j];
            }
        }
    }
}

} // namespace

hipError_t launch_preprocess_backward(const FrameParams& f, const GeometryState& g, const int* radii, const BackwardParams& bw, hipStream_t st)
{
    BwdPreArgs a;
    a.P = f.P; a.D = f.D; a.M = f.M; a.proper_ewa_scaling = f.s.proper_ewa_scaling;
    a.h_x = f.focal_x; a.h_y = f.focal_y; a.tan_fovx = f.tan_fovx; a.tan_fovy = f.tan_fovy; a.scale_modifier = f.scale_modifier;
    a.means3D = f.means3D; a.radii = radii; a.shs = f.shs; a.clamped = g.clamped; a.opacities = f.opacities; a.scales = f.scales;
    a.rotations = f.rotations; a.cov3Ds = f.cov3D_precomp ? f.cov3D_precomp : g.cov3D; // reference rasterizer_impl.cu:500
    a.view = f.viewmatrix; a.proj = f.projmatrix; a.cam = f.cam_pos;
    a.dL_dmean2D = bw.dL_dmean2D; a.grad_rec = bw.grad_rec; a.grad_stride = bw.grad_stride; a.clear_rec = bw.clear_rec; a.dL_dopacity = bw.dL_dopacity; a.dL_dcolor = bw.dL_dcolor;
    a.dL_dmean3D = bw.dL_dmean3D; a.dL_dcov3D = bw.dL_dcov3D; a.dL_dsh = bw.dL_dsh; a.dL_dscale = bw.dL_dscale; a.dL_drot = bw.dL_drot;
    const size_t lds = (a.shs != nullptr && a.M > 0) ? (size_t)256 * (3 * a.M + 1) * sizeof(float) : 0;
    if (lds > 64 * 1024) { // above the default dynamic-LDS limit (M > 21: no SH degree the reference knows)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int n_blocks = (f.P + 255) / 256;
    int b0 = 0, b1 = n_blocks;
    if (bw.chunks > 1) { // Gaussians are independent in this half: a tile-row shard runs it on the id range whose records have been all-reduced already
        b0 = (int)((long long)n_blocks * bw.chunk / bw.chunks);
        b1 = (int)((long long)n_blocks * (bw.chunk + 1) / bw.chunks);
    }
    a.block0 = b0;
    if (b1 <= b0) return hipSuccess;
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3(b1 - b0), dim3(256), lds, st, a);
    return hipGetLastError();
}

} // namespace stp
