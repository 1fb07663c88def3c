/*
 * kektor_compute_legacy.h -- the ten symbols of the reference's existing native library, exported unchanged by
 * libkektor_hip.so and by the static archive libkektordb_compute.a, so that the reference's `-tags rust` build
 * (pkg/core/distance/distance_rust.go:12-17: `#cgo LDFLAGS: -lkektordb_compute -lstdc++`, header
 * native/compute/include/kektordb_compute.h) links against this repository's library without any change.
 *
 *   squared_euclidean_f32 / dot_product_f32 / squared_euclidean_f16 / dot_product_i8
 *       native/compute/include/kektordb_compute.h:8-11, implemented in native/compute/src/lib.rs:22-199 (x86-64:
 *       8-lane FMA accumulation, 128-bit fold, movehl, shuffle; scalar remainder in groups of four) with the scalar
 *       loops of :315-346 where the CPU lacks FMA / F16C / AVX2.  These are HOST routines: one cgo call per
 *       distance is the reference's fine-grained FFI; the GPU path replaces it with the
 *       batched kdb_distance_batch / kdb_search_batch of kektor_hip.h.  They exist here for link compatibility and
 *       for the CPU-side callers the shim does not move to the GPU (e.g. Add / selectNeighbors while the writers
 *       stay in Go).  The three float routines return the bit patterns of the Rust crate (same lanes, same fold).
 *       dot_product_i8 returns the FULL integer dot product -- the value the reference's default build computes
 *       (dotProductGoInt8, pkg/core/distance/distance_go.go:107-116) and its own tests pin (lib.rs:446-458, short
 *       vectors).  DELIBERATE DEVIATION from the crate's AVX2 path for len >= 32: its horizontal reduction
 *       (lib.rs:171-176: `(hi64 + lo64) as i32` over the two 64-bit halves of the folded 128-bit sum) keeps only the 32-bit
 *       lanes 0 and 2 and so drops half of the partial sums; that is a defect of the crate, not an order of summation,
 *       and it is not reproduced (tests/test_compute_legacy.py states the same).
 *   kektordb_embed_init / kektordb_embed / kektordb_embed_batch / kektordb_free_embedding / kektordb_free_embeddings /
 *   kektordb_embed_destroy
 *       native/compute/include/kektordb_compute.h:14-24 (ONNX embedder, native/compute/src/embedder.rs): OUT OF SCOPE
 *       (the embedder half of native/compute).  Stubs: init and embed fail with -1 exactly as the
 *       crate does when no model is loaded (embedder.rs:33-63, 68-80), the free functions accept what they are given.
 */
#ifndef KEKTOR_COMPUTE_LEGACY_H
#define KEKTOR_COMPUTE_LEGACY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

float squared_euclidean_f32(const float *x, const float *y, size_t len);
float dot_product_f32(const float *x, const float *y, size_t len);
float squared_euclidean_f16(const uint16_t *x, const uint16_t *y, size_t len);
int32_t dot_product_i8(const int8_t *x, const int8_t *y, size_t len);

int kektordb_embed_init(const char *model_path, const char *tokenizer_path);
int kektordb_embed(const char *text, float **out_vec, int *out_dim);
void kektordb_free_embedding(float *ptr, int len);
void kektordb_embed_destroy(void);
int kektordb_embed_batch(const char **texts, int count, float ***out_vecs, int *out_count, int *out_dim);
void kektordb_free_embeddings(float **vecs, int count, int dim);

#ifdef __cplusplus
}
#endif
#endif /* KEKTOR_COMPUTE_LEGACY_H */
