// Internal interface between deform.hip (the deformation field's entry points) and hexplane.hip (the plane gathers and
// the plane-gradient pipeline).  The HexPlane kernels are a translation unit of their own because they are built with
// -fno-slp-vectorize: the SLP vectorizer pairs unrelated scalars of their corner arithmetic into v_pk_* through v_mov
// shuffles (a third of the point kernel's VALU instructions), which costs 9 % of the plane-gradient pipeline at C2 and
// 16 % at C5, while the MFMA kernels of deform.hip are a few per cent FASTER with it.
#pragma once
#include "gs4d_internal.h"

#define HEX_SORTS 3
#define HEX_MAX_BINS 65536
struct HexBins {
    int plane[HEX_SORTS];      // plane whose cells order sort s of a level: xy, yz, xz (one sort per level and plane:
                               // the levels' cells do not nest -- 63 vs 127 intervals -- so a finest-level order makes a coarse
                               // level flip between two cells inside every fine cell: measured 116k atomic rows instead of 10k)
    int bmajor[HEX_SORTS];     // 1: key = y0 * W + x0 (B-major), 0: key = x0 * H + y0
    int shift[GS4D_MAX_LEVELS][HEX_SORTS];   // per level: cells are coarsened by this many bits if the plane has more than HEX_MAX_BINS cells
    int nbins;                 // bins per sort (max over the sorts)
};

// counting-sort geometry of the binned plane backward
HexBins gs4d_hex_bins(const gs4d_deform_desc* d);
// points per batch of the plane-gradient pipeline (GS4D_HEX_BATCH_BYTES)
size_t gs4d_hex_batch_points(const gs4d_deform_desc* d, size_t P);
// feat[p][l * C + c] = product over the six planes of level l of their bilinear samples; tprime[p] = t'
int gs4d_hex_launch_forward(const gs4d_deform_desc* d, int P, const float* points, const float* time, const float* force,
                            float* tprime, float* feat, int ldf, hipStream_t s);
// plane gradients (accumulated into g->planes), dL_dpoints += the path through the planes, per-workgroup partial sums of
// the t' gradient in dtprime (3 floats per 256 points; summed by the caller).  hist arrives zeroed.
int gs4d_hex_launch_backward(const gs4d_deform_desc* d, const gs4d_deform_grads* g, int P, const float* points,
                             const float* tprime, const float* dfeat, int ldf, float* dxyz_hex, float* gsix, int* hist,
                             int* rank, int* key, float4* rec, const float* time, const float* force, float* dL_dpoints,
                             float* dtprime, int* ft_done, hipStream_t s);   // *ft_done: d(ft_w), d(ft_b) are complete already
