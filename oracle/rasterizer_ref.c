/*
 * oracle/rasterizer_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar, single-threaded CPU restatement (plain C99, float32 arithmetic) of the
 * reference's tile-based differentiable Gaussian rasterizer.  It is the checker
 * the HIP kernels in 4dgaussians_amd/csrc are compared against; nothing in the
 * product path may include, link or call it (only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do).
 *
 * Reference being restated (paths relative to /root/reference, RAST =
 * submodules/depth-diff-gaussian-rasterization):
 *   RAST/cuda_rasterizer/auxiliary.h:22-164        SH constants, ndc2Pix, getRect, transforms, in_frustum
 *   RAST/cuda_rasterizer/forward.cu:20-71          SH -> RGB with clamp mask
 *   RAST/cuda_rasterizer/forward.cu:74-113         EWA 2D covariance
 *   RAST/cuda_rasterizer/forward.cu:118-152        3D covariance from scale / (un-normalised) quaternion
 *   RAST/cuda_rasterizer/forward.cu:155-256        per-Gaussian preprocess
 *   RAST/cuda_rasterizer/forward.cu:261-379        per-tile alpha blend
 *   RAST/cuda_rasterizer/rasterizer_impl.cu:35-50  getHigherMsb
 *   RAST/cuda_rasterizer/rasterizer_impl.cu:54-138 checkFrustum, duplicateWithKeys, identifyTileRanges
 *   RAST/cuda_rasterizer/rasterizer_impl.cu:198-339 forward orchestration (scan, sort, ranges)
 *   RAST/cuda_rasterizer/backward.cu:20-139        SH backward
 *   RAST/cuda_rasterizer/backward.cu:144-274       conic -> cov2D -> cov3D / mean backward
 *   RAST/cuda_rasterizer/backward.cu:278-341       cov3D -> scale / rotation backward
 *   RAST/cuda_rasterizer/backward.cu:346-412       per-Gaussian preprocess backward
 *   RAST/cuda_rasterizer/backward.cu:415-590       per-tile blend backward
 *   RAST/third_party/glm (column-major mat3; product evaluation order of
 *     glm/detail/type_mat3x3.inl:486-519) -- restated by mat3_mul below.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or fixtures for the
 * rasterizer and its CUDA sources cannot be built here (no nvcc / CUDA headers), so
 * this restatement is "parity unpinned" against the reference binary.  It is pinned
 * instead by (i) an independent dense float64 PyTorch-autograd formulation
 * (tests/ref_torch64.py), (ii) hand-constructed known-answer tests
 * (tests/test_oracle_rasterizer.py) and (iii) a bit-for-bit check of the mat3 helpers
 * against the vendored glm headers (oracle/glm_check.cpp, built into oracle/_ref/).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off is part of the contract: the integer outputs (radii, tile
 * rectangles, sort keys) are decided by float32 arithmetic, and the HIP kernels pin
 * the same no-contraction policy for that code.
 *
 * Deviations from the reference, all deliberate and documented:
 *   - gradient scatter sums of the blend backward are accumulated in double and
 *     rounded once (the reference uses order-nondeterministic float atomics);
 *   - float->int conversions saturate like the GPU's conversion instruction instead
 *     of being undefined for out-of-range values;
 *   - the blend forward additionally reports, per pixel, how close any discrete
 *     decision (alpha < 1/255, T < 1e-4, power > 0) came to flipping ("margin"),
 *     so tests can exclude pixels whose outcome hinges on the last ulp of exp().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3

/* ---- auxiliary.h:22-39 ---- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* ---- column-major 3x3 helpers with glm's evaluation order (note G in SURVEY.md) ---- */
typedef struct { float m[3][3]; } mat3; /* m[col][row] */

static mat3 mat3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
	mat3 r;
	r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
	r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
	r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
	return r;
}

static mat3 mat3_mul(const mat3* A, const mat3* B)
{
	mat3 r;
	for (int c = 0; c < 3; c++)
		for (int w = 0; w < 3; w++)
			r.m[c][w] = A->m[0][w] * B->m[c][0] + A->m[1][w] * B->m[c][1] + A->m[2][w] * B->m[c][2];
	return r;
}

static mat3 mat3_transpose(const mat3* A)
{
	mat3 r;
	for (int c = 0; c < 3; c++)
		for (int w = 0; w < 3; w++)
			r.m[c][w] = A->m[w][c];
	return r;
}

/* exported so oracle/glm_check.cpp can compare these helpers with glm bit for bit */
void gs4d_ref_mat3_mul(const float* A, const float* B, float* out)
{
	mat3 a, b; memcpy(&a, A, sizeof a); memcpy(&b, B, sizeof b);
	mat3 r = mat3_mul(&a, &b); memcpy(out, &r, sizeof r);
}
void gs4d_ref_mat3_transpose(const float* A, float* out)
{
	mat3 a; memcpy(&a, A, sizeof a);
	mat3 r = mat3_transpose(&a); memcpy(out, &r, sizeof r);
}

/* saturating float -> int, as the GPU's conversion does (NaN -> 0) */
static int f2i(float v)
{
	if (v != v) return 0;
	if (v >= 2147483648.0f) return INT32_MAX;
	if (v <= -2147483648.0f) return INT32_MIN;
	return (int)v;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ---- auxiliary.h:41-97 ---- */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin, uint32_t* rmax)
{
	rmin[0] = (uint32_t)imin(gx, imax(0, f2i((px - max_radius) / BLOCK_X)));
	rmin[1] = (uint32_t)imin(gy, imax(0, f2i((py - max_radius) / BLOCK_Y)));
	rmax[0] = (uint32_t)imin(gx, imax(0, f2i((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
	rmax[1] = (uint32_t)imin(gy, imax(0, f2i((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

static void transformPoint4x3(const float* p, const float* m, float* o)
{
	o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
	o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
	o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}

static void transformPoint4x4(const float* p, const float* m, float* o)
{
	o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
	o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
	o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
	o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

static void transformVec4x3Transpose(const float* p, const float* m, float* o)
{
	o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
	o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
	o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}

/* auxiliary.h:107-117 */
static void dnormvdv3(const float* v, const float* dv, float* o)
{
	float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
	o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
	o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* rasterizer_impl.cu:35-50 */
uint32_t gs4d_ref_higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step; else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

/* forward.cu:118-152 (quaternion NOT normalised: forward.cu:127) */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
	mat3 S = mat3_cols(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
	S.m[0][0] = mod * scale[0];
	S.m[1][1] = mod * scale[1];
	S.m[2][2] = mod * scale[2];
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	mat3 R = mat3_cols(
		1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
		2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
		2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
	mat3 Mm = mat3_mul(&S, &R);
	mat3 Mt = mat3_transpose(&Mm);
	mat3 Sigma = mat3_mul(&Mt, &Mm);
	cov3D[0] = Sigma.m[0][0];
	cov3D[1] = Sigma.m[0][1];
	cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1];
	cov3D[4] = Sigma.m[1][2];
	cov3D[5] = Sigma.m[2][2];
}

/* shared by forward.cu:74-113 and backward.cu:166-199: builds T = W*J and cov2D (before the +0.3) */
static void cov2d_parts(const float* mean, float fx, float fy, float tan_fovx, float tan_fovy,
                        const float* cov3D, const float* view, float* t_out, float* txtz_out, float* tytz_out,
                        mat3* T_out, mat3* Vrk_out, mat3* W_out, mat3* cov_out)
{
	float t[3];
	transformPoint4x3(mean, view, t);
	const float limx = 1.3f * tan_fovx;
	const float limy = 1.3f * tan_fovy;
	const float txtz = t[0] / t[2];
	const float tytz = t[1] / t[2];
	t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
	t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
	mat3 J = mat3_cols(
		fx / t[2], 0.0f, -(fx * t[0]) / (t[2] * t[2]),
		0.0f, fy / t[2], -(fy * t[1]) / (t[2] * t[2]),
		0, 0, 0);
	mat3 Wm = mat3_cols(
		view[0], view[4], view[8],
		view[1], view[5], view[9],
		view[2], view[6], view[10]);
	mat3 T = mat3_mul(&Wm, &J);
	mat3 Vrk = mat3_cols(
		cov3D[0], cov3D[1], cov3D[2],
		cov3D[1], cov3D[3], cov3D[4],
		cov3D[2], cov3D[4], cov3D[5]);
	mat3 Tt = mat3_transpose(&T);
	mat3 Vt = mat3_transpose(&Vrk);
	mat3 tmp = mat3_mul(&Tt, &Vt);
	mat3 cov = mat3_mul(&tmp, &T);
	t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
	*txtz_out = txtz; *tytz_out = tytz;
	*T_out = T; *Vrk_out = Vrk; *W_out = Wm; *cov_out = cov;
}

/* forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float* out)
{
	float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
	float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
	dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float res[3];
	for (int c = 0; c < 3; c++) {
		float result = SH_C0 * sh[0 * 3 + c];
		if (deg > 0) {
			float x = dir[0], y = dir[1], z = dir[2];
			result = result - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
			if (deg > 1) {
				float xx = x * x, yy = y * y, zz = z * z;
				float xy = x * y, yz = y * z, xz = x * z;
				result = result +
					SH_C2[0] * xy * sh[4 * 3 + c] +
					SH_C2[1] * yz * sh[5 * 3 + c] +
					SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] +
					SH_C2[3] * xz * sh[7 * 3 + c] +
					SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
				if (deg > 2) {
					result = result +
						SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] +
						SH_C3[1] * xy * z * sh[10 * 3 + c] +
						SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
						SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
						SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
						SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
						SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
				}
			}
		}
		result += 0.5f;
		res[c] = result;
	}
	clamped[3 * idx + 0] = (res[0] < 0);
	clamped[3 * idx + 1] = (res[1] < 0);
	clamped[3 * idx + 2] = (res[2] < 0);
	out[0] = fmaxf(res[0], 0.0f);
	out[1] = fmaxf(res[1], 0.0f);
	out[2] = fmaxf(res[2], 0.0f);
}

/* rasterizer_impl.cu:54-66 + auxiliary.h:139-164 */
void gs4d_ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                           uint8_t* present)
{
	(void)projmatrix;
	for (int idx = 0; idx < P; idx++) {
		float pv[3];
		transformPoint4x3(means3D + 3 * idx, viewmatrix, pv);
		present[idx] = pv[2] > 0.2f;
	}
}

/*
 * forward.cu:155-256 + the tile-count prefix sum of rasterizer_impl.cu:278.
 * All output arrays are caller-allocated with P entries (cov3D 6P, rgb 3P, clamped 3P,
 * means2D 2P, conic_opacity 4P).  Arrays are NOT cleared for culled Gaussians except
 * radii / tiles_touched (matching the reference, which leaves scratch uninitialised);
 * callers should pre-zero what they want to compare.  Returns num_rendered, or -1 if
 * `prefiltered` is set and a Gaussian is culled (the reference traps the device).
 */
int64_t gs4d_ref_preprocess(
	int P, int D, int M,
	const float* means3D, const float* scales, float scale_modifier, const float* rotations,
	const float* opacities, const float* shs, const float* cov3D_precomp, const float* colors_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos,
	int W, int H, float tan_fovx, float tan_fovy, int prefiltered,
	int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb, float* conic_opacity,
	uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets)
{
	const float focal_y = H / (2.0f * tan_fovy);
	const float focal_x = W / (2.0f * tan_fovx);
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	for (int idx = 0; idx < P; idx++) {
		radii[idx] = 0;
		tiles_touched[idx] = 0;
		const float* p_orig = means3D + 3 * idx;
		float p_view[3];
		transformPoint4x3(p_orig, viewmatrix, p_view);
		if (p_view[2] <= 0.2f) {
			if (prefiltered) return -1;
			continue;
		}
		float p_hom[4];
		transformPoint4x4(p_orig, projmatrix, p_hom);
		float p_w = 1.0f / (p_hom[3] + 0.0000001f);
		float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};

		const float* cov3D;
		if (cov3D_precomp) {
			cov3D = cov3D_precomp + 6 * (size_t)idx;
		} else {
			computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * (size_t)idx);
			cov3D = cov3Ds + 6 * (size_t)idx;
		}
		float t[3], txtz, tytz; mat3 T, Vrk, Wm, cov2;
		cov2d_parts(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, t, &txtz, &tytz, &T, &Vrk, &Wm, &cov2);
		cov2.m[0][0] += 0.3f;
		cov2.m[1][1] += 0.3f;
		float cx = cov2.m[0][0], cy = cov2.m[0][1], cz = cov2.m[1][1];

		float det = (cx * cz - cy * cy);
		if (det == 0.0f) continue;
		float det_inv = 1.f / det;
		float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};

		float mid = 0.5f * (cx + cz);
		float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
		float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
		float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
		float pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
		uint32_t rmin[2], rmax[2];
		getRect(pix[0], pix[1], f2i(my_radius), gx, gy, rmin, rmax);
		if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

		if (!colors_precomp)
			computeColorFromSH(idx, D, M, means3D, cam_pos, shs, clamped, rgb + 3 * (size_t)idx);

		depths[idx] = p_view[2];
		radii[idx] = f2i(my_radius);
		means2D[2 * idx] = pix[0];
		means2D[2 * idx + 1] = pix[1];
		conic_opacity[4 * idx + 0] = conic[0];
		conic_opacity[4 * idx + 1] = conic[1];
		conic_opacity[4 * idx + 2] = conic[2];
		conic_opacity[4 * idx + 3] = opacities[idx];
		tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
	}
	uint32_t run = 0;
	for (int idx = 0; idx < P; idx++) { run += tiles_touched[idx]; point_offsets[idx] = run; }
	return P > 0 ? (int64_t)point_offsets[P - 1] : 0;
}

/*
 * rasterizer_impl.cu:70-111 (key emission), :304-309 (stable LSD radix sort, note S),
 * :311 + :116-138 (tile ranges).  keys/values arrays have R entries; ranges has
 * 2 * gx * gy entries and is zeroed here.
 */
void gs4d_ref_binning(
	int P, int64_t R, int W, int H,
	const float* means2D, const float* depths, const uint32_t* point_offsets, const int* radii,
	uint64_t* keys_unsorted, uint32_t* values_unsorted, uint64_t* keys_sorted, uint32_t* values_sorted,
	uint32_t* ranges)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
	for (int idx = 0; idx < P; idx++) {
		if (radii[idx] > 0) {
			uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
			uint32_t rmin[2], rmax[2];
			getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
			uint32_t dbits; memcpy(&dbits, depths + idx, 4);
			for (int y = (int)rmin[1]; y < (int)rmax[1]; y++)
				for (int x = (int)rmin[0]; x < (int)rmax[0]; x++) {
					uint64_t key = (uint64_t)(y * gx + x);
					key <<= 32;
					key |= dbits;
					keys_unsorted[off] = key;
					values_unsorted[off] = (uint32_t)idx;
					off++;
				}
		}
	}
	/* stable LSD radix sort, 8-bit digits over the whole 64-bit key */
	if (R > 0) {
		uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * R);
		uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * R);
		uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * R);
		uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * R);
		memcpy(ka, keys_unsorted, sizeof(uint64_t) * R);
		memcpy(va, values_unsorted, sizeof(uint32_t) * R);
		for (int pass = 0; pass < 8; pass++) {
			size_t cnt[257];
			memset(cnt, 0, sizeof cnt);
			int sh = 8 * pass;
			for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> sh) & 255) + 1]++;
			for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
			for (int64_t i = 0; i < R; i++) {
				size_t dst = cnt[(ka[i] >> sh) & 255]++;
				kb[dst] = ka[i]; vb[dst] = va[i];
			}
			uint64_t* tk = ka; ka = kb; kb = tk;
			uint32_t* tv = va; va = vb; vb = tv;
		}
		memcpy(keys_sorted, ka, sizeof(uint64_t) * R);
		memcpy(values_sorted, va, sizeof(uint32_t) * R);
		free(ka); free(va); free(kb); free(vb);
	}
	memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
	for (int64_t idx = 0; idx < R; idx++) {
		uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
		if (idx == 0)
			ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
			if (currtile != prevtile) {
				ranges[2 * prevtile + 1] = (uint32_t)idx;
				ranges[2 * currtile] = (uint32_t)idx;
			}
		}
		if (idx == R - 1)
			ranges[2 * currtile + 1] = (uint32_t)R;
	}
}

static float rel_margin(float v, float thr)
{
	float d = fabsf(v - thr) / thr;
	return d;
}

/*
 * forward.cu:261-379.  `features` is rgb (or colors_precomp), 3 floats per Gaussian.
 * The block-wide early exit (forward.cu:312-314) only stops work for pixels that are
 * all done already, so per-pixel iteration reproduces its outputs exactly.
 * margin[pix] (may be NULL) = smallest relative distance of any evaluated decision
 * to its threshold for that pixel (alpha vs 1/255, test_T vs 1e-4, and |power|
 * relative to the magnitude of its terms for the power > 0 test).
 */
void gs4d_ref_render_forward(
	int W, int H, const uint32_t* ranges, const uint32_t* point_list,
	const float* means2D, const float* features, const float* depths, const float* conic_opacity,
	const float* bg, float* out_color, float* out_depth, float* final_T, uint32_t* n_contrib, float* margin)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X;
	for (int py = 0; py < H; py++)
		for (int px = 0; px < W; px++) {
			const uint32_t tile = (uint32_t)((py / BLOCK_Y) * gx + (px / BLOCK_X));
			const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
			const uint32_t pix_id = (uint32_t)(W * py + px);
			const float pixf[2] = {(float)px, (float)py};
			float T = 1.0f;
			uint32_t contributor = 0, last_contributor = 0;
			float C[NUM_CHANNELS] = {0, 0, 0};
			float Dd = 0;
			float mg = FLT_MAX;
			for (uint32_t k = r0; k < r1; k++) {
				contributor++;
				const uint32_t id = point_list[k];
				float dx = means2D[2 * id] - pixf[0];
				float dy = means2D[2 * id + 1] - pixf[1];
				const float* co = conic_opacity + 4 * (size_t)id;
				float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
				{
					float mag = fabsf(0.5f * co[0] * dx * dx) + fabsf(0.5f * co[2] * dy * dy) + fabsf(co[1] * dx * dy);
					if (mag > 0) { float m = fabsf(power) / mag; if (m < mg) mg = m; }
				}
				if (power > 0.0f) continue;
				float alpha = fminf(0.99f, co[3] * expf(power));
				{ float m = rel_margin(alpha, 1.0f / 255.0f); if (m < mg) mg = m; }
				if (alpha < 1.0f / 255.0f) continue;
				float test_T = T * (1 - alpha);
				{ float m = rel_margin(test_T, 0.0001f); if (m < mg) mg = m; }
				if (test_T < 0.0001f) break; /* done = true */
				for (int ch = 0; ch < NUM_CHANNELS; ch++)
					C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
				Dd += depths[id] * alpha * T;
				T = test_T;
				last_contributor = contributor;
			}
			final_T[pix_id] = T;
			n_contrib[pix_id] = last_contributor;
			for (int ch = 0; ch < NUM_CHANNELS; ch++)
				out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
			out_depth[pix_id] = Dd;
			if (margin) margin[pix_id] = mg;
		}
}

/*
 * backward.cu:415-590.  Scatter sums go to double accumulators (P x {3,4,1,3,1}) and are
 * rounded to float once at the end; everything per (pixel, Gaussian) pair is float32
 * exactly as the reference computes it.  With float_accum != 0 every scatter add is rounded to
 * float32 instead (a fixed-order stand-in for the reference's float atomics), which tests use to
 * measure the float32 summation-order noise floor of the reference algorithm itself.
 * dL_dmean2D is (P,3) with .z untouched (0),
 * dL_dconic is (P,4) with slot 2 untouched (0).
 */
void gs4d_ref_render_backward(
	int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
	const float* means2D, const float* conic_opacity, const float* colors, const float* depths,
	const float* final_Ts, const uint32_t* n_contrib, const float* dL_dpixels, const float* dL_dpixel_depths,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors, float* dL_ddepths,
	int float_accum)
{
	const int gx = (W + BLOCK_X - 1) / BLOCK_X;
	double* acc = (double*)calloc((size_t)P * 10, sizeof(double));
#define ACC(i_, x_) do { double n_ = a[i_] + (double)(x_); a[i_] = float_accum ? (double)(float)n_ : n_; } while (0)
	const float ddelx_dx = (float)(0.5 * W);
	const float ddely_dy = (float)(0.5 * H);
	for (int py = 0; py < H; py++)
		for (int px = 0; px < W; px++) {
			const uint32_t tile = (uint32_t)((py / BLOCK_Y) * gx + (px / BLOCK_X));
			const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
			const uint32_t pix_id = (uint32_t)(W * py + px);
			const float pixf[2] = {(float)px, (float)py};
			const float T_final = final_Ts[pix_id];
			float T = T_final;
			uint32_t contributor = r1 - r0;
			const uint32_t last_contributor = n_contrib[pix_id];
			float accum_rec[3] = {0, 0, 0};
			float dL_dpixel[3];
			for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
			float dL_dpixel_depth = dL_dpixel_depths[pix_id];
			float accum_depth_rec = 0;
			float last_alpha = 0;
			float last_color[3] = {0, 0, 0};
			float last_depth = 0;
			for (uint32_t k = r1; k-- > r0;) {
				contributor--;
				if (contributor >= last_contributor) continue;
				const uint32_t id = point_list[k];
				const float dx = means2D[2 * id] - pixf[0];
				const float dy = means2D[2 * id + 1] - pixf[1];
				const float* co = conic_opacity + 4 * (size_t)id;
				const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
				if (power > 0.0f) continue;
				const float G = expf(power);
				const float alpha = fminf(0.99f, co[3] * G);
				if (alpha < 1.0f / 255.0f) continue;
				T = T / (1.f - alpha);
				const float dchannel_dcolor = alpha * T;
				const float dpixel_depth_ddepth = alpha * T;
				float dL_dalpha = 0.0f;
				double* a = acc + (size_t)id * 10;
				for (int ch = 0; ch < 3; ch++) {
					const float c = colors[id * 3 + ch];
					accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
					last_color[ch] = c;
					const float dL_dchannel = dL_dpixel[ch];
					dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
					ACC(ch, dchannel_dcolor * dL_dchannel);
				}
				const float c_d = depths[id];
				accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
				last_depth = c_d;
				dL_dalpha += (c_d - accum_depth_rec) * dL_dpixel_depth;
				ACC(3, dpixel_depth_ddepth * dL_dpixel_depth);
				dL_dalpha *= T;
				last_alpha = alpha;
				float bg_dot_dpixel = 0;
				for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
				dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
				const float dL_dG = co[3] * dL_dalpha;
				const float gdx = G * dx;
				const float gdy = G * dy;
				const float dG_ddelx = -gdx * co[0] - gdy * co[1];
				const float dG_ddely = -gdy * co[2] - gdx * co[1];
				ACC(4, dL_dG * dG_ddelx * ddelx_dx);
				ACC(5, dL_dG * dG_ddely * ddely_dy);
				ACC(6, -0.5f * gdx * dx * dL_dG);
				ACC(7, -0.5f * gdx * dy * dL_dG);
				ACC(8, -0.5f * gdy * dy * dL_dG);
				ACC(9, G * dL_dalpha);
			}
		}
	for (int id = 0; id < P; id++) {
		const double* a = acc + (size_t)id * 10;
		dL_dcolors[3 * id + 0] = (float)a[0];
		dL_dcolors[3 * id + 1] = (float)a[1];
		dL_dcolors[3 * id + 2] = (float)a[2];
		dL_ddepths[id] = (float)a[3];
		dL_dmean2D[3 * id + 0] = (float)a[4];
		dL_dmean2D[3 * id + 1] = (float)a[5];
		dL_dmean2D[3 * id + 2] = 0.0f;
		dL_dconic[4 * id + 0] = (float)a[6];
		dL_dconic[4 * id + 1] = (float)a[7];
		dL_dconic[4 * id + 2] = 0.0f;
		dL_dconic[4 * id + 3] = (float)a[8];
		dL_dopacity[id] = (float)a[9];
	}
	free(acc);
#undef ACC
}

/* backward.cu:20-139 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                   const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                   float* dL_dmeans, float* dL_dshs)
{
	const float* pos = means + 3 * idx;
	float dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
	float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
	float dir[3] = {dir_orig[0] / len, dir_orig[1] / len, dir_orig[2] / len};
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float dL_dRGB[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
	dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
	dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
	dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
	float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
	float x = dir[0], y = dir[1], z = dir[2];
	float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SETSH(k, w) do { for (int c_ = 0; c_ < 3; c_++) dL_dsh[(k) * 3 + c_] = (w) * dL_dRGB[c_]; } while (0)
	float dRGBdsh0 = SH_C0;
	SETSH(0, dRGBdsh0);
	if (deg > 0) {
		float dRGBdsh1 = -SH_C1 * y;
		float dRGBdsh2 = SH_C1 * z;
		float dRGBdsh3 = -SH_C1 * x;
		SETSH(1, dRGBdsh1);
		SETSH(2, dRGBdsh2);
		SETSH(3, dRGBdsh3);
		for (int c = 0; c < 3; c++) {
			dRGBdx[c] = -SH_C1 * sh[3 * 3 + c];
			dRGBdy[c] = -SH_C1 * sh[1 * 3 + c];
			dRGBdz[c] = SH_C1 * sh[2 * 3 + c];
		}
		if (deg > 1) {
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			float dRGBdsh4 = SH_C2[0] * xy;
			float dRGBdsh5 = SH_C2[1] * yz;
			float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
			float dRGBdsh7 = SH_C2[3] * xz;
			float dRGBdsh8 = SH_C2[4] * (xx - yy);
			SETSH(4, dRGBdsh4);
			SETSH(5, dRGBdsh5);
			SETSH(6, dRGBdsh6);
			SETSH(7, dRGBdsh7);
			SETSH(8, dRGBdsh8);
			for (int c = 0; c < 3; c++) {
				const float s4 = sh[4 * 3 + c], s5 = sh[5 * 3 + c], s6 = sh[6 * 3 + c], s7 = sh[7 * 3 + c], s8 = sh[8 * 3 + c];
				dRGBdx[c] += SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
				dRGBdy[c] += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8;
				dRGBdz[c] += SH_C2[1] * y * s5 + SH_C2[2] * 2.f * 2.f * z * s6 + SH_C2[3] * x * s7;
			}
			if (deg > 2) {
				float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
				float dRGBdsh10 = SH_C3[1] * xy * z;
				float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
				float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
				float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
				float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
				SETSH(9, dRGBdsh9);
				SETSH(10, dRGBdsh10);
				SETSH(11, dRGBdsh11);
				SETSH(12, dRGBdsh12);
				SETSH(13, dRGBdsh13);
				SETSH(14, dRGBdsh14);
				SETSH(15, dRGBdsh15);
				for (int c = 0; c < 3; c++) {
					const float s9 = sh[9 * 3 + c], s10 = sh[10 * 3 + c], s11 = sh[11 * 3 + c], s12 = sh[12 * 3 + c];
					const float s13 = sh[13 * 3 + c], s14 = sh[14 * 3 + c], s15 = sh[15 * 3 + c];
					dRGBdx[c] += (
						SH_C3[0] * s9 * 3.f * 2.f * xy +
						SH_C3[1] * s10 * yz +
						SH_C3[2] * s11 * -2.f * xy +
						SH_C3[3] * s12 * -3.f * 2.f * xz +
						SH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) +
						SH_C3[5] * s14 * 2.f * xz +
						SH_C3[6] * s15 * 3.f * (xx - yy));
					dRGBdy[c] += (
						SH_C3[0] * s9 * 3.f * (xx - yy) +
						SH_C3[1] * s10 * xz +
						SH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) +
						SH_C3[3] * s12 * -3.f * 2.f * yz +
						SH_C3[4] * s13 * -2.f * xy +
						SH_C3[5] * s14 * -2.f * yz +
						SH_C3[6] * s15 * -3.f * 2.f * xy);
					dRGBdz[c] += (
						SH_C3[1] * s10 * xy +
						SH_C3[2] * s11 * 4.f * 2.f * yz +
						SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) +
						SH_C3[4] * s13 * 4.f * 2.f * xz +
						SH_C3[5] * s14 * (xx - yy));
				}
			}
		}
	}
#undef SETSH
	float dL_ddir[3] = {
		dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
		dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
		dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
	float dL_dmean[3];
	dnormvdv3(dir_orig, dL_ddir, dL_dmean);
	dL_dmeans[3 * idx + 0] += dL_dmean[0];
	dL_dmeans[3 * idx + 1] += dL_dmean[1];
	dL_dmeans[3 * idx + 2] += dL_dmean[2];
}

/* backward.cu:278-341 (no normalisation Jacobian: backward.cu:340) */
static void computeCov3D_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                             float* dL_dscales, float* dL_drots)
{
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	mat3 R = mat3_cols(
		1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
		2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
		2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
	mat3 S = mat3_cols(1.0f, 0, 0, 0, 1.0f, 0, 0, 0, 1.0f);
	float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
	S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
	mat3 Mm = mat3_mul(&S, &R);
	const float* d = dL_dcov3Ds + 6 * (size_t)idx;
	mat3 dL_dSigma = mat3_cols(
		d[0], 0.5f * d[1], 0.5f * d[2],
		0.5f * d[1], d[3], 0.5f * d[4],
		0.5f * d[2], 0.5f * d[4], d[5]);
	mat3 M2;
	for (int c = 0; c < 3; c++) for (int w = 0; w < 3; w++) M2.m[c][w] = Mm.m[c][w] * 2.0f;
	mat3 dL_dM = mat3_mul(&M2, &dL_dSigma);
	mat3 Rt = mat3_transpose(&R);
	mat3 dL_dMt = mat3_transpose(&dL_dM);
	float* ds = dL_dscales + 3 * idx;
	for (int c = 0; c < 3; c++)
		ds[c] = Rt.m[c][0] * dL_dMt.m[c][0] + Rt.m[c][1] * dL_dMt.m[c][1] + Rt.m[c][2] * dL_dMt.m[c][2];
	for (int w = 0; w < 3; w++) {
		dL_dMt.m[0][w] *= s[0];
		dL_dMt.m[1][w] *= s[1];
		dL_dMt.m[2][w] *= s[2];
	}
#define MT(c, w) dL_dMt.m[c][w]
	float* dq = dL_drots + 4 * idx;
	dq[0] = 2 * z * (MT(0, 1) - MT(1, 0)) + 2 * y * (MT(2, 0) - MT(0, 2)) + 2 * x * (MT(1, 2) - MT(2, 1));
	dq[1] = 2 * y * (MT(1, 0) + MT(0, 1)) + 2 * z * (MT(2, 0) + MT(0, 2)) + 2 * r * (MT(1, 2) - MT(2, 1)) - 4 * x * (MT(2, 2) + MT(1, 1));
	dq[2] = 2 * x * (MT(1, 0) + MT(0, 1)) + 2 * r * (MT(2, 0) - MT(0, 2)) + 2 * z * (MT(1, 2) + MT(2, 1)) - 4 * y * (MT(2, 2) + MT(0, 0));
	dq[3] = 2 * r * (MT(0, 1) - MT(1, 0)) + 2 * x * (MT(2, 0) + MT(0, 2)) + 2 * y * (MT(1, 2) + MT(2, 1)) - 4 * z * (MT(1, 1) + MT(0, 0));
#undef MT
}

/*
 * backward.cu:144-274 (computeCov2DCUDA) followed by backward.cu:346-412 (preprocessCUDA),
 * as BACKWARD::preprocess runs them (backward.cu:592-658).  Output arrays must be
 * zero-initialised by the caller like rasterize_points.cu:154-163 does.
 * shs / scales may be NULL (colors_precomp / cov3D_precomp paths).
 */
void gs4d_ref_preprocess_backward(
	int P, int D, int M,
	const float* means3D, const int* radii, const float* shs, const uint8_t* clamped,
	const float* scales, const float* rotations, float scale_modifier, const float* cov3Ds,
	const float* view, const float* proj, int W, int H, float tan_fovx, float tan_fovy, const float* campos,
	const float* dL_dmean2D, const float* dL_dconics, float* dL_dmeans, float* dL_dcolor, float* dL_ddepth,
	float* dL_dcov, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
	const float h_y = H / (2.0f * tan_fovy);
	const float h_x = W / (2.0f * tan_fovx);
	for (int idx = 0; idx < P; idx++) {
		if (!(radii[idx] > 0)) continue;
		const float* cov3D = cov3Ds + 6 * (size_t)idx;
		const float* mean = means3D + 3 * idx;
		float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
		float t[3], txtz, tytz; mat3 T, Vrk, Wm, cov2D;
		cov2d_parts(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view, t, &txtz, &tytz, &T, &Vrk, &Wm, &cov2D);
		const float limx = 1.3f * tan_fovx;
		const float limy = 1.3f * tan_fovy;
		const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
		const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
		float a = cov2D.m[0][0] += 0.3f;
		float b = cov2D.m[0][1];
		float c = cov2D.m[1][1] += 0.3f;
		float denom = a * c - b * b;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(c_, r_) T.m[c_][r_]
#define VV(c_, r_) Vrk.m[c_][r_]
#define WW(c_, r_) Wm.m[c_][r_]
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
			dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
			dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
			dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
			dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
			dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
			dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
			dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
			dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
		} else {
			for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
		}
		float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
			(TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
		float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
			(TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
		float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
			(TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
		float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
			(TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
		float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
			(TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
		float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
			(TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
		float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
		float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
		float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
		float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
		float tz = 1.f / t[2];
		float tz2 = tz * tz;
		float tz3 = tz2 * tz;
		float dL_dt[3];
		dL_dt[0] = x_grad_mul * -h_x * tz2 * dL_dJ02;
		dL_dt[1] = y_grad_mul * -h_y * tz2 * dL_dJ12;
		dL_dt[2] = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
		float dL_dmean_cov[3];
		transformVec4x3Transpose(dL_dt, view, dL_dmean_cov);
		dL_dmeans[3 * idx + 0] = dL_dmean_cov[0];
		dL_dmeans[3 * idx + 1] = dL_dmean_cov[1];
		dL_dmeans[3 * idx + 2] = dL_dmean_cov[2];

		/* ---- backward.cu:372-411 ---- */
		const float* m = mean;
		float m_hom[4];
		transformPoint4x4(m, proj, m_hom);
		float m_w = 1.0f / (m_hom[3] + 0.0000001f);
		float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
		float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
		const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
		float dm[3];
		dm[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
		dm[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
		dm[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
		dL_dmeans[3 * idx + 0] += dm[0];
		dL_dmeans[3 * idx + 1] += dm[1];
		dL_dmeans[3 * idx + 2] += dm[2];
		float mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
		float dm2[3];
		dm2[0] = (view[2] - view[3] * mul3) * dL_ddepth[idx];
		dm2[1] = (view[6] - view[7] * mul3) * dL_ddepth[idx];
		dm2[2] = (view[10] - view[11] * mul3) * dL_ddepth[idx];
		dL_dmeans[3 * idx + 0] += dm2[0];
		dL_dmeans[3 * idx + 1] += dm2[1];
		dL_dmeans[3 * idx + 2] += dm2[2];
		if (shs)
			computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
		if (scales)
			computeCov3D_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov, dL_dscale, dL_drot);
	}
}
