/*
 * oracle/knn_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of simple_knn's distCUDA2: for every point, the mean of the three
 * smallest squared distances to the other points.
 *
 * Reference restated (relative to /root/reference/submodules/simple-knn):
 *   spatial.cu:15-26        distCUDA2 wrapper (float32 (P,3) in, float32 (P,) out)
 *   simple_knn.cu:45-70     prepMorton / coord2Morton (30-bit code, bbox-normalised)
 *   simple_knn.cu:78-117    boxMinMax: AABB of each run of 1024 Morton-sorted points
 *   simple_knn.cu:119-129   distBoxPoint
 *   simple_knn.cu:131-145   updateKBest<3> (sorted insertion, strict '>')
 *   simple_knn.cu:147-183   boxMeanDist (seed with +-3 Morton neighbours, conservative box pruning)
 *   simple_knn.cu:185-221   SimpleKNN::knn (bbox reduce with init (0,0,0): the box always
 *                           contains the origin, simple_knn.cu:191)
 *
 * Two entry points:
 *   gs4d_ref_knn_brute   -- O(P^2) exact 3-NN, same float32 distance expression and the
 *                           same (b0+b1+b2)/3.0f epilogue.
 *   gs4d_ref_knn_morton  -- the reference's Morton/box algorithm step by step.
 * The pruning is conservative, so both give identical bits; tests assert that.
 *
 * PARITY STATUS: the reference has no tests or fixtures for this path and its CUDA
 * sources cannot be built here -> "parity unpinned" against the reference binary;
 * pinned by brute force == Morton restatement and by scipy.spatial.cKDTree neighbour
 * sets (tests/test_oracle_knn.py).
 *
 * Build: gcc -O2 -ffp-contract=off (oracle/Makefile).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BOX_SIZE 1024

static void update3(const float* ref, const float* p, float* knn)
{
	float dx = p[0] - ref[0], dy = p[1] - ref[1], dz = p[2] - ref[2];
	float dist = dx * dx + dy * dy + dz * dz;
	for (int j = 0; j < 3; j++) {
		if (knn[j] > dist) {
			float t = knn[j];
			knn[j] = dist;
			dist = t;
		}
	}
}

void gs4d_ref_knn_brute(int P, const float* points, float* out)
{
	for (int i = 0; i < P; i++) {
		float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
		for (int j = 0; j < P; j++) {
			if (j == i) continue;
			update3(points + 3 * i, points + 3 * j, best);
		}
		out[i] = (best[0] + best[1] + best[2]) / 3.0f;
	}
}

static uint32_t f2u(float v)
{
	if (v != v) return 0;
	if (v <= 0.0f) return 0;
	if (v >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)v;
}

static uint32_t prepMorton(uint32_t x)
{
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}

uint32_t gs4d_ref_morton(const float* c, const float* minn, const float* maxx)
{
	uint32_t x = prepMorton(f2u(((c[0] - minn[0]) / (maxx[0] - minn[0])) * ((1 << 10) - 1)));
	uint32_t y = prepMorton(f2u(((c[1] - minn[1]) / (maxx[1] - minn[1])) * ((1 << 10) - 1)));
	uint32_t z = prepMorton(f2u(((c[2] - minn[2]) / (maxx[2] - minn[2])) * ((1 << 10) - 1)));
	return x | (y << 1) | (z << 2);
}

typedef struct { float minn[3]; float maxx[3]; } MinMax;

static float distBoxPoint(const MinMax* box, const float* p)
{
	float diff[3] = {0, 0, 0};
	for (int a = 0; a < 3; a++)
		if (p[a] < box->minn[a] || p[a] > box->maxx[a])
			diff[a] = fminf(fabsf(p[a] - box->minn[a]), fabsf(p[a] - box->maxx[a]));
	return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
}

/* codes_out / order_out (may be NULL) expose the Morton codes and the sorted permutation. */
void gs4d_ref_knn_morton(int P, const float* points, float* out, uint32_t* codes_out, uint32_t* order_out)
{
	if (P <= 0) return;
	float minn[3] = {0, 0, 0}, maxx[3] = {0, 0, 0};
	for (int i = 0; i < P; i++)
		for (int a = 0; a < 3; a++) {
			minn[a] = fminf(minn[a], points[3 * i + a]);
			maxx[a] = fmaxf(maxx[a], points[3 * i + a]);
		}
	uint32_t* codes = (uint32_t*)malloc(sizeof(uint32_t) * P);
	uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * P);
	uint32_t* codes2 = (uint32_t*)malloc(sizeof(uint32_t) * P);
	uint32_t* idx2 = (uint32_t*)malloc(sizeof(uint32_t) * P);
	for (int i = 0; i < P; i++) {
		codes[i] = gs4d_ref_morton(points + 3 * i, minn, maxx);
		idx[i] = (uint32_t)i;
	}
	if (codes_out) memcpy(codes_out, codes, sizeof(uint32_t) * P);
	for (int pass = 0; pass < 4; pass++) { /* stable LSD radix sort, like cub SortPairs */
		size_t cnt[257];
		memset(cnt, 0, sizeof cnt);
		int sh = 8 * pass;
		for (int i = 0; i < P; i++) cnt[((codes[i] >> sh) & 255) + 1]++;
		for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
		for (int i = 0; i < P; i++) {
			size_t dst = cnt[(codes[i] >> sh) & 255]++;
			codes2[dst] = codes[i];
			idx2[dst] = idx[i];
		}
		uint32_t* t = codes; codes = codes2; codes2 = t;
		t = idx; idx = idx2; idx2 = t;
	}
	if (order_out) memcpy(order_out, idx, sizeof(uint32_t) * P);
	int nb = (P + BOX_SIZE - 1) / BOX_SIZE;
	MinMax* boxes = (MinMax*)malloc(sizeof(MinMax) * nb);
	for (int b = 0; b < nb; b++) {
		MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
		int hi = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
		for (int i = b * BOX_SIZE; i < hi; i++)
			for (int a = 0; a < 3; a++) {
				me.minn[a] = fminf(me.minn[a], points[3 * idx[i] + a]);
				me.maxx[a] = fmaxf(me.maxx[a], points[3 * idx[i] + a]);
			}
		boxes[b] = me;
	}
	for (int i = 0; i < P; i++) {
		const float* point = points + 3 * idx[i];
		float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
		int lo = i - 3 > 0 ? i - 3 : 0, hi = i + 3 < P - 1 ? i + 3 : P - 1;
		for (int j = lo; j <= hi; j++) {
			if (j == i) continue;
			update3(point, points + 3 * idx[j], best);
		}
		float reject = best[2];
		best[0] = best[1] = best[2] = FLT_MAX;
		for (int b = 0; b < nb; b++) {
			float dist = distBoxPoint(&boxes[b], point);
			if (dist > reject || dist > best[2]) continue;
			int bhi = (b + 1) * BOX_SIZE < P ? (b + 1) * BOX_SIZE : P;
			for (int j = b * BOX_SIZE; j < bhi; j++) {
				if (j == i) continue;
				update3(point, points + 3 * idx[j], best);
			}
		}
		out[idx[i]] = (best[0] + best[1] + best[2]) / 3.0f;
	}
	free(codes); free(idx); free(codes2); free(idx2); free(boxes);
}

/* ------------------------------------------------------------------------------------------------
 * k-nearest-neighbour query of the fork's rigidity regulariser: a literal
 * restatement of pointops2/src/knnquery/knnquery_cuda_kernel.cu:5-108 for ONE batch segment
 * (start = 0, end = n): per query a max-heap of nsample (<= 100) candidates seeded with
 * (1e10, start), strict `d2 < root` replacement in ascending point order, then the in-place heap sort
 * (:38-47), which leaves the distances ascending (which of several equal-distance points survives at
 * the k-th place depends on the heap's history).
 * PARITY UNPINNED against the reference binary (CUDA, unbuildable here); the reference has no test
 * for this operator.  Pinned by a numpy brute force in tests/test_oracle_knn.py.
 * ------------------------------------------------------------------------------------------------ */
static void kq_reheap(float* dist, int* idx, int k) /* knnquery_cuda_kernel.cu:21-36 */
{
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return;
        float td = dist[root]; dist[root] = dist[child]; dist[child] = td;
        int ti = idx[root]; idx[root] = idx[child]; idx[child] = ti;
        root = child;
        child = root * 2 + 1;
    }
}

void gs4d_ref_knn_query(int n, const float* xyz, int m, const float* new_xyz, int nsample, int* idx, float* dist2)
{
    for (int q = 0; q < m; q++) { /* knnquery_cuda_kernel.cu:65-108, one thread per query */
        float best_dist[100];
        int best_idx[100];
        const float nx = new_xyz[3 * q], ny = new_xyz[3 * q + 1], nz = new_xyz[3 * q + 2];
        for (int i = 0; i < nsample; i++) { best_dist[i] = 1e10f; best_idx[i] = 0; }
        for (int i = 0; i < n; i++) {
            const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            const float d2 = (nx - x) * (nx - x) + (ny - y) * (ny - y) + (nz - z) * (nz - z);
            if (d2 < best_dist[0]) {
                best_dist[0] = d2;
                best_idx[0] = i;
                kq_reheap(best_dist, best_idx, nsample);
            }
        }
        for (int i = nsample - 1; i > 0; i--) { /* heap_sort, :38-47 */
            float td = best_dist[0]; best_dist[0] = best_dist[i]; best_dist[i] = td;
            int ti = best_idx[0]; best_idx[0] = best_idx[i]; best_idx[i] = ti;
            kq_reheap(best_dist, best_idx, i);
        }
        for (int i = 0; i < nsample; i++) { idx[(size_t)q * nsample + i] = best_idx[i]; dist2[(size_t)q * nsample + i] = best_dist[i]; }
    }
}
