// oracle/glm_check.cpp -- TEST INFRASTRUCTURE (container only; needs the reference tree).
//
// Pins the index conventions of the oracle's own mat3 helpers (rasterizer_ref.c) against the glm
// headers vendored in the reference (RAST/third_party/glm, included from where they lie -- nothing
// is copied): glm::mat3(a..i) fills COLUMNS, M[i][j] is column i / row j, A*B is the ordinary matrix
// product, glm::transpose.  These are the conventions forward.cu:118-152 and :74-113 and
// backward.cu:144-341 of the reference rely on (note G).  The comparison is
// bit for bit (both sides are compiled with -ffp-contract=off).
//
// Build + run:  make -C oracle ref && oracle/_ref/glm_check
#include <cstdio>
#include <cstring>
#include <random>

#include <glm/glm.hpp>

extern "C" void gs4d_ref_mat3_mul(const float* A, const float* B, float* out);
extern "C" void gs4d_ref_mat3_transpose(const float* A, float* out);

static bool same_bits(const float* a, const float* b, int n) { return std::memcmp(a, b, n * sizeof(float)) == 0; }

int main()
{
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-3.f, 3.f);
    int bad = 0;
    for (int it = 0; it < 100000; it++) {
        float a[9], b[9], mine[9], ref[9];
        for (int i = 0; i < 9; i++) { a[i] = U(rng); b[i] = U(rng); }
        // the oracle stores a mat3 as 9 floats in glm's own memory order (column 0 first)
        glm::mat3 A(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
        glm::mat3 B(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]);
        if (A[1][2] != a[5] || A[2][0] != a[6]) { std::printf("constructor convention differs\n"); return 1; }
        const glm::mat3 C = A * B;
        std::memcpy(ref, &C[0][0], sizeof ref);
        gs4d_ref_mat3_mul(a, b, mine);
        if (!same_bits(mine, ref, 9)) bad++;
        const glm::mat3 T = glm::transpose(A);
        std::memcpy(ref, &T[0][0], sizeof ref);
        gs4d_ref_mat3_transpose(a, mine);
        if (!same_bits(mine, ref, 9)) bad++;
        // the two composite forms the kernels use: Sigma = transpose(M) * M and cov = transpose(T) * transpose(V) * T
        const glm::mat3 S = glm::transpose(A) * A;
        float at[9], s2[9];
        gs4d_ref_mat3_transpose(a, at);
        gs4d_ref_mat3_mul(at, a, s2);
        std::memcpy(ref, &S[0][0], sizeof ref);
        if (!same_bits(s2, ref, 9)) bad++;
        const glm::mat3 Cv = glm::transpose(A) * glm::transpose(B) * A;
        float bt[9], t1[9], t2[9];
        gs4d_ref_mat3_transpose(b, bt);
        gs4d_ref_mat3_mul(at, bt, t1);
        gs4d_ref_mat3_mul(t1, a, t2);
        std::memcpy(ref, &Cv[0][0], sizeof ref);
        if (!same_bits(t2, ref, 9)) bad++;
    }
    std::printf("glm_check: %d mismatching results out of 400000\n", bad);
    return bad != 0;
}
