// Per-Gaussian backward for gfx950: gathers each Gaussian's per-instance gradient rows (written
// by blend_bwd.hip) in fixed order, then propagates through conic -> cov2D -> cov3D / mean,
// projection, depth, SH colour and scale / rotation.
//
// Behaviour matched (reference, RAST = submodules/depth-diff-gaussian-rasterization):
//   RAST/cuda_rasterizer/backward.cu:144-274  computeCov2DCUDA
//   RAST/cuda_rasterizer/backward.cu:346-412  preprocessCUDA (mean from mean2D + depth)
//   RAST/cuda_rasterizer/backward.cu:20-139   SH backward
//   RAST/cuda_rasterizer/backward.cu:278-341  cov3D -> scale / rotation (no normalisation Jacobian)
// The two reference kernels are fused into one pass; Gaussians covering more than
// GS4D_BIG_ROWS tiles are deferred to a second launch where a whole wave sums their rows
// (fixed lane assignment + fixed-order shuffle tree, so still deterministic) -- otherwise one lane
// would walk thousands of dependent gathers.
// Compiled with -ffp-contract=off like the forward preprocess.
#include "gs4d_internal.h"
#include "gs4d_math.h"

using namespace gs4d;

#define GS4D_BIG_ROWS 64
#define NQ 10

namespace {

struct GeomBwdArgs {
    int P, D, M, W, H;
    const float* means3D; const int* radii; const float* shs; const uint8_t* clamped;
    const float* scales; const float* rotations; float scale_modifier; const float* cov3D;
    const float* view; const float* proj; float h_x, h_y, tan_fovx, tan_fovy; const float* campos;
    const uint32_t* tiles_touched; const uint32_t* point_offsets; const float* inst_grad;
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor; float* dL_ddepth;
    float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot;
    uint32_t* big_list; uint32_t* big_count;
    Gs4dActBwd act; int has_act;             // the composite call's activation backward, applied per Gaussian (gs4d_internal.h)
};

// the activation backward of Gaussian idx from the dL_dscale / dL_drot / dL_dopacity this thread has just written
__device__ __forceinline__ void gaussian_activation_backward(const GeomBwdArgs& A, int idx)
{
    activation_backward_point(idx, A.act.scales_a, reinterpret_cast<const float4*>(A.act.rot_raw), A.act.opacity_a, A.dL_dscale,
                              reinterpret_cast<const float4*>(A.dL_drot), A.dL_dopacity, A.act.d_log_scales,
                              reinterpret_cast<float4*>(A.act.d_rot), A.act.d_opacity_logit);
}

__device__ __forceinline__ float3 dnormvdv3(const float3 v, const float3 dv)
{
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 o;
    o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return o;
}

// SH backward: writes dL_dsh for the active degree, returns the mean gradient through the view direction.
// Every read of `sh` comes before the first write of `dL_dsh`: the staged form of geom_backward_kernel hands over ONE
// LDS row for both (the coefficients in, their gradients out).  The expressions are the reference's, term for term.
__device__ __forceinline__ float3 sh_backward(int deg, int max_coeffs, const float3 pos, const float3 campos,
                                              const float* __restrict__ sh, const uint8_t* __restrict__ clamped3,
                                              const float dL_dcolor[3], float* dL_dsh)
{
    const float3 dir_orig = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    float g[3] = {dL_dcolor[0], dL_dcolor[1], dL_dcolor[2]};
    g[0] *= clamped3[0] ? 0 : 1;
    g[1] *= clamped3[1] ? 0 : 1;
    g[2] *= clamped3[2] ? 0 : 1;
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
    // ---- reads
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            ddx[c] = -kSH_C1 * sh[3 * 3 + c];
            ddy[c] = -kSH_C1 * sh[1 * 3 + c];
            ddz[c] = kSH_C1 * sh[2 * 3 + c];
        }
        if (deg > 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float s4 = sh[4 * 3 + c], s5 = sh[5 * 3 + c], s6 = sh[6 * 3 + c], s7 = sh[7 * 3 + c],
                            s8 = sh[8 * 3 + c];
                ddx[c] += kSH_C2[0] * y * s4 + kSH_C2[2] * 2.f * -x * s6 + kSH_C2[3] * z * s7 + kSH_C2[4] * 2.f * x * s8;
                ddy[c] += kSH_C2[0] * x * s4 + kSH_C2[1] * z * s5 + kSH_C2[2] * 2.f * -y * s6 + kSH_C2[4] * 2.f * -y * s8;
                ddz[c] += kSH_C2[1] * y * s5 + kSH_C2[2] * 2.f * 2.f * z * s6 + kSH_C2[3] * x * s7;
            }
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float s9 = sh[9 * 3 + c], s10 = sh[10 * 3 + c], s11 = sh[11 * 3 + c], s12 = sh[12 * 3 + c];
                    const float s13 = sh[13 * 3 + c], s14 = sh[14 * 3 + c], s15 = sh[15 * 3 + c];
                    ddx[c] += (kSH_C3[0] * s9 * 3.f * 2.f * xy + kSH_C3[1] * s10 * yz + kSH_C3[2] * s11 * -2.f * xy +
                               kSH_C3[3] * s12 * -3.f * 2.f * xz + kSH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +
                               kSH_C3[5] * s14 * 2.f * xz + kSH_C3[6] * s15 * 3.f * (xx - yy));
                    ddy[c] += (kSH_C3[0] * s9 * 3.f * (xx - yy) + kSH_C3[1] * s10 * xz +
                               kSH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) + kSH_C3[3] * s12 * -3.f * 2.f * yz +
                               kSH_C3[4] * s13 * -2.f * xy + kSH_C3[5] * s14 * -2.f * yz +
                               kSH_C3[6] * s15 * -3.f * 2.f * xy);
                    ddz[c] += (kSH_C3[1] * s10 * xy + kSH_C3[2] * s11 * 4.f * 2.f * yz +
                               kSH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) + kSH_C3[4] * s13 * 4.f * 2.f * xz +
                               kSH_C3[5] * s14 * (xx - yy));
                }
            }
        }
    }
    // ---- writes
#define GS4D_SETSH(k, wgt)                                        \
    do {                                                          \
        const float w_ = (wgt);                                   \
        dL_dsh[(k) * 3 + 0] = w_ * g[0];                          \
        dL_dsh[(k) * 3 + 1] = w_ * g[1];                          \
        dL_dsh[(k) * 3 + 2] = w_ * g[2];                          \
    } while (0)
    GS4D_SETSH(0, kSH_C0);
    if (deg > 0) {
        GS4D_SETSH(1, -kSH_C1 * y);
        GS4D_SETSH(2, kSH_C1 * z);
        GS4D_SETSH(3, -kSH_C1 * x);
        if (deg > 1) {
            GS4D_SETSH(4, kSH_C2[0] * xy);
            GS4D_SETSH(5, kSH_C2[1] * yz);
            GS4D_SETSH(6, kSH_C2[2] * (2.f * zz - xx - yy));
            GS4D_SETSH(7, kSH_C2[3] * xz);
            GS4D_SETSH(8, kSH_C2[4] * (xx - yy));
            if (deg > 2) {
                GS4D_SETSH(9, kSH_C3[0] * y * (3.f * xx - yy));
                GS4D_SETSH(10, kSH_C3[1] * xy * z);
                GS4D_SETSH(11, kSH_C3[2] * y * (4.f * zz - xx - yy));
                GS4D_SETSH(12, kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                GS4D_SETSH(13, kSH_C3[4] * x * (4.f * zz - xx - yy));
                GS4D_SETSH(14, kSH_C3[5] * z * (xx - yy));
                GS4D_SETSH(15, kSH_C3[6] * x * (xx - 3.f * yy));
            }
        }
    }
#undef GS4D_SETSH
    const float3 dL_ddir = make_float3(ddx[0] * g[0] + ddx[1] * g[1] + ddx[2] * g[2],
                                       ddy[0] * g[0] + ddy[1] * g[1] + ddy[2] * g[2],
                                       ddz[0] * g[0] + ddz[1] * g[1] + ddz[2] * g[2]);
    return dnormvdv3(dir_orig, dL_ddir);
}

// Everything after the row sums are known.  r[0..2] colour, r[3] depth, r[4..5] mean2D, r[6..8] conic, r[9] opacity.
// sh_row / dsh_row: this Gaussian's coefficients and where their gradients go (global rows, or one LDS row for both)
__device__ __forceinline__ void gaussian_backward(const GeomBwdArgs& A, int idx, const float r[NQ], const float* sh_row, float* dsh_row)
{
    const float* view = A.view;
    const float* proj = A.proj;
    // the blend sums are also part of the API (and dL_dmean2D / dL_dcolor / dL_dopacity are returned to autograd)
    A.dL_dcolor[3 * idx + 0] = r[0];
    A.dL_dcolor[3 * idx + 1] = r[1];
    A.dL_dcolor[3 * idx + 2] = r[2];
    A.dL_ddepth[idx] = r[3];
    A.dL_dmean2D[3 * idx + 0] = r[4];
    A.dL_dmean2D[3 * idx + 1] = r[5];
    A.dL_dmean2D[3 * idx + 2] = 0.f;
    A.dL_dconic[4 * idx + 0] = r[6];
    A.dL_dconic[4 * idx + 1] = r[7];
    A.dL_dconic[4 * idx + 3] = r[8];
    A.dL_dopacity[idx] = r[9];

    const float3 mean = make_float3(A.means3D[3 * idx], A.means3D[3 * idx + 1], A.means3D[3 * idx + 2]);
    float c3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c3[k] = A.cov3D[6 * (size_t)idx + k];
    const float dcx = r[6], dcy = r[7], dcz = r[8];
    const Ewa e = ewa_project(mean, A.h_x, A.h_y, A.tan_fovx, A.tan_fovy, c3, view);
    const float limx = 1.3f * A.tan_fovx, limy = 1.3f * A.tan_fovy;
    const float x_grad_mul = (e.txtz < -limx || e.txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (e.tytz < -limy || e.tytz > limy) ? 0.f : 1.f;
    const float a = e.cov.e[0][0] + 0.3f;
    const float b = e.cov.e[0][1];
    const float c = e.cov.e[1][1] + 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(c_, r_) e.T.e[c_][r_]
#define VV(c_, r_) e.V.e[c_][r_]
#define WW(c_, r_) e.Wm.e[c_][r_]
    float dcov[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
        dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
        dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
        dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
        dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
        dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
        dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) dcov[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) A.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];

    const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                          (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
    const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                          (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
    const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                          (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
    const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                          (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
    const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                          (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
    const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                          (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
    const float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
    const float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
    const float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
    const float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
    const float tz = 1.f / e.t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -A.h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -A.h_y * tz2 * dL_dJ12;
    const float dL_dtz = -A.h_x * tz2 * dL_dJ00 - A.h_y * tz2 * dL_dJ11 + (2 * A.h_x * e.t.x) * tz3 * dL_dJ02 +
                         (2 * A.h_y * e.t.y) * tz3 * dL_dJ12;
    float3 dmean;
    dmean.x = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
    dmean.y = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    dmean.z = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

    // mean gradient from the 2D mean (projection Jacobian) and from the depth
    const float3 m = mean;
    const float4 m_hom = projective_point(m, proj);
    const float m_w = 1.0f / (m_hom.w + 0.0000001f);
    const float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    const float g2x = r[4], g2y = r[5];
    float3 d1;
    d1.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    d1.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    d1.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    dmean.x += d1.x; dmean.y += d1.y; dmean.z += d1.z;
    const float mul3 = view[2] * m.x + view[6] * m.y + view[10] * m.z + view[14];
    float3 d2;
    d2.x = (view[2] - view[3] * mul3) * r[3];
    d2.y = (view[6] - view[7] * mul3) * r[3];
    d2.z = (view[10] - view[11] * mul3) * r[3];
    dmean.x += d2.x; dmean.y += d2.y; dmean.z += d2.z;

    if (A.shs != nullptr) {
        const float3 cp = make_float3(A.campos[0], A.campos[1], A.campos[2]);
        const float3 d3 = sh_backward(A.D, A.M, m, cp, sh_row, A.clamped + 3 * (size_t)idx, r, dsh_row);
        dmean.x += d3.x; dmean.y += d3.y; dmean.z += d3.z;
    }
    A.dL_dmean3D[3 * idx + 0] = dmean.x;
    A.dL_dmean3D[3 * idx + 1] = dmean.y;
    A.dL_dmean3D[3 * idx + 2] = dmean.z;

    if (A.scales != nullptr) {
        const float4 q = reinterpret_cast<const float4*>(A.rotations)[idx];
        const float rr = q.x, x = q.y, y = q.z, z = q.w;
        const Mat3 R = quat_columns(q);
        const float3 s = make_float3(A.scale_modifier * A.scales[3 * idx], A.scale_modifier * A.scales[3 * idx + 1],
                                     A.scale_modifier * A.scales[3 * idx + 2]);
        const Mat3 Mm = mat3_product(scale_columns(s), R);
        const Mat3 dSigma = mat3_from_columns(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2],
                                              0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                                              0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
        Mat3 M2;
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int w = 0; w < 3; w++) M2.e[cc][w] = Mm.e[cc][w] * 2.0f;
        const Mat3 dL_dM = mat3_product(M2, dSigma);
        const Mat3 Rt = mat3_transposed(R);
        Mat3 Mt = mat3_transposed(dL_dM);
        A.dL_dscale[3 * idx + 0] = Rt.e[0][0] * Mt.e[0][0] + Rt.e[0][1] * Mt.e[0][1] + Rt.e[0][2] * Mt.e[0][2];
        A.dL_dscale[3 * idx + 1] = Rt.e[1][0] * Mt.e[1][0] + Rt.e[1][1] * Mt.e[1][1] + Rt.e[1][2] * Mt.e[1][2];
        A.dL_dscale[3 * idx + 2] = Rt.e[2][0] * Mt.e[2][0] + Rt.e[2][1] * Mt.e[2][1] + Rt.e[2][2] * Mt.e[2][2];
#pragma unroll
        for (int w = 0; w < 3; w++) {
            Mt.e[0][w] *= s.x;
            Mt.e[1][w] *= s.y;
            Mt.e[2][w] *= s.z;
        }
#define MT(c_, w_) Mt.e[c_][w_]
        float4 dq;
        dq.x = 2 * z * (MT(0, 1) - MT(1, 0)) + 2 * y * (MT(2, 0) - MT(0, 2)) + 2 * x * (MT(1, 2) - MT(2, 1));
        dq.y = 2 * y * (MT(1, 0) + MT(0, 1)) + 2 * z * (MT(2, 0) + MT(0, 2)) + 2 * rr * (MT(1, 2) - MT(2, 1)) - 4 * x * (MT(2, 2) + MT(1, 1));
        dq.z = 2 * x * (MT(1, 0) + MT(0, 1)) + 2 * rr * (MT(2, 0) - MT(0, 2)) + 2 * z * (MT(1, 2) + MT(2, 1)) - 4 * y * (MT(2, 2) + MT(0, 0));
        dq.w = 2 * rr * (MT(0, 1) - MT(1, 0)) + 2 * x * (MT(2, 0) + MT(0, 2)) + 2 * y * (MT(1, 2) + MT(2, 1)) - 4 * z * (MT(1, 1) + MT(0, 0));
#undef MT
        reinterpret_cast<float4*>(A.dL_drot)[idx] = dq;
    }
}

// STAGED (16 SH coefficients per Gaussian, 16-byte aligned arrays): a lane's 48 coefficient reads and 48 gradient writes
// are 192 bytes apart from its neighbour's, i.e. every one of those 96 instructions touched 64 different cache lines and
// the kernel sat on the texture-address path (17 of its 30 us at C2 by that count).  The workgroup now copies its
// Gaussians' coefficient rows into LDS with coalesced float4 loads (row pitch 49 words: conflict-free for equal
// columns), every lane works on its own LDS row -- gradients overwrite the coefficients, sh_backward reads before it
// writes -- and the rows leave as coalesced float4 stores.  Rows of culled and deferred Gaussians leave as zeros (the
// deferred ones are filled in by geom_backward_big_kernel afterwards), so dL_dsh needs no clearing by the caller.
// what gaussian_backward() writes (dL_dmean2D and the SH row apart), as zeros
__device__ __forceinline__ void gaussian_zero_outputs(const GeomBwdArgs& A, int idx)
{
    A.dL_dcolor[3 * idx + 0] = 0.f; A.dL_dcolor[3 * idx + 1] = 0.f; A.dL_dcolor[3 * idx + 2] = 0.f;
    A.dL_ddepth[idx] = 0.f;
    A.dL_dconic[4 * idx + 0] = 0.f; A.dL_dconic[4 * idx + 1] = 0.f; A.dL_dconic[4 * idx + 2] = 0.f; A.dL_dconic[4 * idx + 3] = 0.f;
    A.dL_dopacity[idx] = 0.f;
    A.dL_dmean3D[3 * idx + 0] = 0.f; A.dL_dmean3D[3 * idx + 1] = 0.f; A.dL_dmean3D[3 * idx + 2] = 0.f;
    if (A.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) A.dL_dcov3D[6 * (size_t)idx + k] = 0.f;
    }
    if (A.scales != nullptr) {
        A.dL_dscale[3 * idx + 0] = 0.f; A.dL_dscale[3 * idx + 1] = 0.f; A.dL_dscale[3 * idx + 2] = 0.f;
        reinterpret_cast<float4*>(A.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

#define GEOM_BLOCK 128
#define GEOM_SH_PITCH 49
template <bool STAGED>
__global__ __launch_bounds__(GEOM_BLOCK) void geom_backward_kernel(GeomBwdArgs A)
{
    extern __shared__ float s_sh[];                       // STAGED: [GEOM_BLOCK][GEOM_SH_PITCH]
    const int idx = blockIdx.x * GEOM_BLOCK + threadIdx.x;
    const int base = blockIdx.x * GEOM_BLOCK;
    const int nrows = min(GEOM_BLOCK, A.P - base);
    if (STAGED) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(A.shs + (size_t)base * 48);
        for (int i = threadIdx.x; i < nrows * 12; i += GEOM_BLOCK) {
            const float4 v = src[i];
            float* d = s_sh + (i / 12) * GEOM_SH_PITCH + (i % 12) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
    }
    float* my_row = s_sh + threadIdx.x * GEOM_SH_PITCH;
    bool wrote_row = false;
    if (idx < A.P) {
        // requested together (under the tests below they were three memory latencies in a row)
        const int radius = A.radii[idx];
        const uint32_t cnt = A.tiles_touched[idx];
        const uint32_t end = A.point_offsets[idx];
        if (!(radius > 0)) {
            // dL_dmean2D is written in full (the caller may hand over uninitialised memory for it, see include/gs4d.h), and so
            // is every other per-Gaussian output: a Gaussian that touched no tile gets its zeros here, so the composite call
            // needs no 12 MB memset in front of its backward (csrc/render.hip; the reference's caller passes zero tensors)
            A.dL_dmean2D[3 * idx + 0] = 0.f; A.dL_dmean2D[3 * idx + 1] = 0.f; A.dL_dmean2D[3 * idx + 2] = 0.f;
            gaussian_zero_outputs(A, idx);
            if (A.has_act) gaussian_activation_backward(A, idx);
        } else if (cnt > GS4D_BIG_ROWS) {
            const uint32_t slot = atomicAdd(A.big_count, 1u);
            A.big_list[slot] = (uint32_t)idx;
        } else {
            const uint32_t first = end - cnt;
            // The rows are summed in DOUBLE: the kernel waits on memory most of the time, so the wider adds are free, and the
            // sums then carry only the rounding of the per-tile partials (the reference's float atomics add in arbitrary order;
            // the oracle sums in double) -- dL_dscales / dL_drotations, derived through a cancelling chain, meet 1e-4 this way.
            double r[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) r[q] = 0.0;
            const float4* rows = reinterpret_cast<const float4*>(A.inst_grad);
            // The Gaussian's rows are contiguous (blend_backward_kernel writes a row at its emission slot): four rows are
            // requested together, summed in the same order as one by one.
            uint32_t s = first;
            for (; s + 4 <= end; s += 4) {
                float4 a[4], b[4], c[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { a[u] = rows[(size_t)(s + u) * 3 + 0]; b[u] = rows[(size_t)(s + u) * 3 + 1]; c[u] = rows[(size_t)(s + u) * 3 + 2]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    r[0] += a[u].x; r[1] += a[u].y; r[2] += a[u].z; r[3] += a[u].w;
                    r[4] += b[u].x; r[5] += b[u].y; r[6] += b[u].z; r[7] += b[u].w;
                    r[8] += c[u].x; r[9] += c[u].y;
                }
            }
            for (; s < end; s++) {
                const float4 a = rows[(size_t)s * 3 + 0], b = rows[(size_t)s * 3 + 1], c = rows[(size_t)s * 3 + 2];
                r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w;
                r[4] += b.x; r[5] += b.y; r[6] += b.z; r[7] += b.w;
                r[8] += c.x; r[9] += c.y;
            }
            float rf[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) rf[q] = (float)r[q];
            if (STAGED) {
                gaussian_backward(A, idx, rf, my_row, my_row);
                for (int k = (A.D + 1) * (A.D + 1) * 3; k < 48; k++) my_row[k] = 0.f;     // coefficients above the active degree
                wrote_row = true;
            } else {
                gaussian_backward(A, idx, rf, A.shs ? A.shs + (size_t)idx * A.M * 3 : nullptr,
                                  A.dL_dsh ? A.dL_dsh + (size_t)idx * A.M * 3 : nullptr);
            }
            if (A.has_act) gaussian_activation_backward(A, idx);
        }
    }
    if (STAGED) {
        if (!wrote_row)
            for (int k = 0; k < 48; k++) my_row[k] = 0.f;
        __syncthreads();
        float4* __restrict__ dst = reinterpret_cast<float4*>(A.dL_dsh + (size_t)base * 48);
        for (int i = threadIdx.x; i < nrows * 12; i += GEOM_BLOCK) {
            const float* d = s_sh + (i / 12) * GEOM_SH_PITCH + (i % 12) * 4;
            dst[i] = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
}

// One WAVE per deferred Gaussian (four per workgroup, no workgroup barriers): lane t sums rows first+t, first+t+64, ...,
// a fixed-order shuffle tree combines the 64 partials and lane 0 finishes the Gaussian.
__global__ __launch_bounds__(256) void geom_backward_big_kernel(GeomBwdArgs A)
{
    if (A.has_act && A.act.l1_loss && blockIdx.x == 0) {     // uniform: the forward's L1 loss, finished here (GS4D_LOSS_AFTER_BACKWARD)
        __shared__ float s_l1[256];
        l1_finish(A.act.l1_partial, A.act.l1_n, A.act.l1_loss, s_l1);
    }
    const int lane = threadIdx.x & 63;
    const uint32_t n_big = *A.big_count;
    const uint32_t nwaves = gridDim.x * 4u;
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < n_big; w += nwaves) {      // wave-uniform
        const int idx = (int)A.big_list[w];
        const uint32_t cnt = A.tiles_touched[idx];
        const uint32_t end = A.point_offsets[idx];
        const uint32_t first = end - cnt;
        double r[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) r[q] = 0.0;
        const float4* rows = reinterpret_cast<const float4*>(A.inst_grad);
        for (uint32_t s = first + lane; s < end; s += 64) {
            const float4 a = rows[(size_t)s * 3 + 0], b = rows[(size_t)s * 3 + 1], c = rows[(size_t)s * 3 + 2];
            r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w;
            r[4] += b.x; r[5] += b.y; r[6] += b.z; r[7] += b.w;
            r[8] += c.x; r[9] += c.y;
        }
#pragma unroll
        for (int q = 0; q < NQ; q++)
            for (int off = 32; off >= 1; off >>= 1) r[q] += __shfl_xor(r[q], off, 64);
        float rf[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) rf[q] = (float)r[q];
        if (lane == 0) {
            gaussian_backward(A, idx, rf, A.shs ? A.shs + (size_t)idx * A.M * 3 : nullptr, A.dL_dsh ? A.dL_dsh + (size_t)idx * A.M * 3 : nullptr);
            if (A.has_act) gaussian_activation_backward(A, idx);
        }
    }
}

}  // namespace

// true: the launch below writes every element of dL_dsh (zeros for culled Gaussians and inactive degrees), the caller need
// not clear it -- except when nothing was rendered, when no kernel runs at all
bool gs4d_geom_backward_fills_dsh(int M, const float* shs, const float* dL_dsh)
{
    return M == 16 && shs && dL_dsh && !((uintptr_t)shs & 15) && !((uintptr_t)dL_dsh & 15);
}

int gs4d_launch_geom_backward(int P, int D, int M, int R, int W, int H, const float* means3D, const int* radii,
                              const float* shs, const float* scales, const float* rotations, float scale_modifier,
                              const float* cov3D, const float* viewmatrix, const float* projmatrix,
                              float tan_fovx, float tan_fovy, const float* campos, const GeomState& g,
                              const BinningState& b, const float* inst_grad, float* dL_dmean2D, float* dL_dconic,
                              float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                              float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                              hipStream_t s, const Gs4dActBwd* act)
{
    GS4D_PROF("geom_backward", s);
    GeomBwdArgs A;
    A.has_act = act && scales ? 1 : 0;                       // (the folded form needs the scale / rotation gradients)
    A.act = act ? *act : Gs4dActBwd{};
    A.P = P; A.D = D; A.M = M; A.W = W; A.H = H;
    A.means3D = means3D; A.radii = radii; A.shs = shs; A.clamped = g.clamped;
    A.scales = scales; A.rotations = rotations; A.scale_modifier = scale_modifier; A.cov3D = cov3D;
    A.view = viewmatrix; A.proj = projmatrix;
    A.h_y = H / (2.0f * tan_fovy);
    A.h_x = W / (2.0f * tan_fovx);
    A.tan_fovx = tan_fovx; A.tan_fovy = tan_fovy; A.campos = campos;
    A.tiles_touched = g.tiles_touched; A.point_offsets = g.point_offsets; A.inst_grad = inst_grad;
    A.dL_dmean2D = dL_dmean2D; A.dL_dconic = dL_dconic; A.dL_dopacity = dL_dopacity; A.dL_dcolor = dL_dcolor;
    A.dL_ddepth = dL_ddepth; A.dL_dmean3D = dL_dmean3D; A.dL_dcov3D = dL_dcov3D; A.dL_dsh = dL_dsh;
    A.dL_dscale = dL_dscale; A.dL_drot = dL_drot;
    // deferred-Gaussian worklist lives behind the per-instance rows (see gs4d_backward_scratch_bytes)
    // the counter was cleared by blend_backward_kernel (api.hip)
    A.big_count = gs4d_backward_big_count(inst_grad, R);
    A.big_list = (uint32_t*)((char*)A.big_count + 256);
    if (gs4d_geom_backward_fills_dsh(M, shs, dL_dsh))
        hipLaunchKernelGGL(geom_backward_kernel<true>, dim3((P + GEOM_BLOCK - 1) / GEOM_BLOCK), dim3(GEOM_BLOCK),
                           sizeof(float) * GEOM_BLOCK * GEOM_SH_PITCH, s, A);
    else
        hipLaunchKernelGGL(geom_backward_kernel<false>, dim3((P + GEOM_BLOCK - 1) / GEOM_BLOCK), dim3(GEOM_BLOCK), 0, s, A);
    GS4D_LAUNCH_CHECK("geom_backward", debug, s);
    hipLaunchKernelGGL(geom_backward_big_kernel, dim3(1024), dim3(256), 0, s, A);
    GS4D_LAUNCH_CHECK("geom_backward_big", debug, s);
    return GS4D_OK;
}
