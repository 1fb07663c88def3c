// compute_legacy.cpp -- host-only: the per-pair distance symbols of the reference's native library
// (native/compute/include/kektordb_compute.h:8-11) and stubs for its embedder symbols (:14-24), so that the reference's
// `-tags rust` build links against this library unchanged.  See include/kektor_compute_legacy.h.
//
// Arithmetic (bit-compatible with native/compute/src/lib.rs on x86-64):
//   f32 / f16: eight lanes accumulate with FMA over the 8-element blocks; the 256-bit sum is folded high half onto low
//              half, then upper pair onto lower pair, then lane 1 onto lane 0 (lib.rs:22-31); the remaining < 8
//              elements are added one by one, each product rounded before the add (lib.rs:52-71: no FMA there).
//   int8:      32-element blocks widened to i16, pairwise products summed into eight i32 lanes (madd), lanes folded;
//              exact integer arithmetic, so any order gives the same i32 (wrap-around included).  NOT the crate's value for
//              len >= 32 on AVX2: its reduction (lib.rs:171-176) sums only lanes 0 and 2 of the folded vector -- a defect, not
//              reproduced (kektor_compute_legacy.h); the default Go build's dotProductGoInt8 is the reference value.
//   CPUs without FMA (+F16C for the half routine) / AVX2 take the scalar loops of lib.rs:315-346.
// No GPU, no oracle: plain C++ with target attributes and a run-time CPU check.
#include "../../include/kektor_compute_legacy.h"
#include <immintrin.h>
#include <string.h>

#if defined(__GNUC__)
#define KCL_API extern "C" __attribute__((visibility("default")))
#else
#define KCL_API extern "C"
#endif

namespace {

inline float half_to_float(uint16_t h) { // IEEE binary16 -> binary32, exact (subnormals, infinities, NaN included)
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { // subnormal: normalise
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | man << 13;
    } else {
        bits = sign | (exp + 127 - 15) << 23 | man << 13;
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

__attribute__((target("avx,fma"))) inline float fold8(__m256 v) {
    const __m128 hi = _mm256_extractf128_ps(v, 1), lo = _mm256_castps256_ps128(v);
    __m128 s = _mm_add_ps(lo, hi);
    s = _mm_add_ps(s, _mm_movehl_ps(s, s));
    s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
    return _mm_cvtss_f32(s);
}

__attribute__((target("avx,fma"))) float l2_f32_fma(const float *x, const float *y, size_t n) {
    __m256 acc = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 d = _mm256_sub_ps(_mm256_loadu_ps(x + i), _mm256_loadu_ps(y + i));
        acc = _mm256_fmadd_ps(d, d, acc);
    }
    volatile float total = fold8(acc); // the tail adds round product and sum separately (no contraction)
    for (; i < n; i++) {
        const float d = x[i] - y[i];
        const volatile float p = d * d;
        total = total + p;
    }
    return total;
}

__attribute__((target("avx,fma"))) float dot_f32_fma(const float *x, const float *y, size_t n) {
    __m256 acc = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(x + i), _mm256_loadu_ps(y + i), acc);
    volatile float total = fold8(acc);
    for (; i < n; i++) {
        const volatile float p = x[i] * y[i];
        total = total + p;
    }
    return total;
}

__attribute__((target("avx,fma,f16c"))) float l2_f16_fma(const uint16_t *x, const uint16_t *y, size_t n) {
    __m256 acc = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 a = _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(x + i)));
        const __m256 b = _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(y + i)));
        const __m256 d = _mm256_sub_ps(a, b);
        acc = _mm256_fmadd_ps(d, d, acc);
    }
    volatile float total = fold8(acc);
    for (; i < n; i++) {
        const float d = half_to_float(x[i]) - half_to_float(y[i]);
        const volatile float p = d * d;
        total = total + p;
    }
    return total;
}

__attribute__((target("avx2"))) int32_t dot_i8_avx2(const int8_t *x, const int8_t *y, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(x + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(y + i));
        const __m256i lo = _mm256_madd_epi16(_mm256_cvtepi8_epi16(_mm256_castsi256_si128(a)), _mm256_cvtepi8_epi16(_mm256_castsi256_si128(b)));
        const __m256i hi = _mm256_madd_epi16(_mm256_cvtepi8_epi16(_mm256_extracti128_si256(a, 1)), _mm256_cvtepi8_epi16(_mm256_extracti128_si256(b, 1)));
        acc = _mm256_add_epi32(_mm256_add_epi32(acc, lo), hi);
    }
    alignas(32) int32_t lanes[8];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), acc);
    uint32_t total = 0; // modular i32 arithmetic, as the reference's wrapping adds
    for (int l = 0; l < 8; l++) total += (uint32_t)lanes[l];
    for (; i < n; i++) total += (uint32_t)((int32_t)x[i] * (int32_t)y[i]);
    return (int32_t)total;
}

// scalar loops (lib.rs:315-346): product rounded, then added
float l2_f32_scalar(const float *x, const float *y, size_t n) {
    volatile float s = 0.f;
    for (size_t i = 0; i < n; i++) {
        const float d = x[i] - y[i];
        const volatile float p = d * d;
        s = s + p;
    }
    return s;
}
float dot_f32_scalar(const float *x, const float *y, size_t n) {
    volatile float s = 0.f;
    for (size_t i = 0; i < n; i++) {
        const volatile float p = x[i] * y[i];
        s = s + p;
    }
    return s;
}
float l2_f16_scalar(const uint16_t *x, const uint16_t *y, size_t n) {
    volatile float s = 0.f;
    for (size_t i = 0; i < n; i++) {
        const float d = half_to_float(x[i]) - half_to_float(y[i]);
        const volatile float p = d * d;
        s = s + p;
    }
    return s;
}
int32_t dot_i8_scalar(const int8_t *x, const int8_t *y, size_t n) {
    int64_t s = 0;
    for (size_t i = 0; i < n; i++) s += (int64_t)x[i] * (int64_t)y[i];
    return (int32_t)s;
}

struct Cpu {
    bool fma, f16c, avx2;
    Cpu() {
        __builtin_cpu_init();
        fma = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx");
        f16c = __builtin_cpu_supports("f16c");
        avx2 = __builtin_cpu_supports("avx2");
    }
};
const Cpu &cpu() {
    static const Cpu c;
    return c;
}

} // namespace

KCL_API float squared_euclidean_f32(const float *x, const float *y, size_t len) {
    return cpu().fma ? l2_f32_fma(x, y, len) : l2_f32_scalar(x, y, len);
}
KCL_API float dot_product_f32(const float *x, const float *y, size_t len) {
    return cpu().fma ? dot_f32_fma(x, y, len) : dot_f32_scalar(x, y, len);
}
KCL_API float squared_euclidean_f16(const uint16_t *x, const uint16_t *y, size_t len) {
    return (cpu().fma && cpu().f16c) ? l2_f16_fma(x, y, len) : l2_f16_scalar(x, y, len);
}
KCL_API int32_t dot_product_i8(const int8_t *x, const int8_t *y, size_t len) {
    return cpu().avx2 ? dot_i8_avx2(x, y, len) : dot_i8_scalar(x, y, len);
}

// ---- embedder symbols: out of scope, present so that the link succeeds; every call reports "no model" ----------------
KCL_API int kektordb_embed_init(const char *, const char *) { return -1; }
KCL_API int kektordb_embed(const char *, float **out_vec, int *out_dim) {
    if (out_vec) *out_vec = nullptr;
    if (out_dim) *out_dim = 0;
    return -1;
}
KCL_API void kektordb_free_embedding(float *, int) {}
KCL_API void kektordb_embed_destroy(void) {}
KCL_API int kektordb_embed_batch(const char **, int, float ***out_vecs, int *out_count, int *out_dim) {
    if (out_vecs) *out_vecs = nullptr;
    if (out_count) *out_count = 0;
    if (out_dim) *out_dim = 0;
    return -1;
}
KCL_API void kektordb_free_embeddings(float **, int, int) {}
