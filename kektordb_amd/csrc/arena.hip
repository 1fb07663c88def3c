// arena.hip -- reader for the reference's vector arena files (pkg/storage/mmap/arena.go), so the GPU
// mirror can be populated from a KektorDB data directory without Go in the loop.
//
// Layout restated from arena.go:14-23, 73-77, 90-95, 335-342, 403-404:
//   files  <dir>/arena_%04d.bin, each DefaultChunkSize = 64 MiB
//   header 64 bytes: LE u32 magic 0x4B414F4E, LE u32 version 1, LE u32 dim, u8 precision
//          (0 float32, 1 float16, 2 int8), rest reserved/zero
//   rows   dense, vectorSize = dim * elem bytes, vecsPerChunk = (64 MiB - 64) / vectorSize,
//          physical slot p lives in chunk p / vecsPerChunk at byte 64 + (p % vecsPerChunk) * vectorSize
//   logical internal id -> physical slot through slotTable (arena.go:121-151, ArenaState :252-270);
//   0xFFFFFFFF = unallocated.  Without frees the table is the identity id -> id-1... the FIRST id the
//   index uses is 1 and receives slot 0 (hnsw_index.go:590, arena.go:143-146).
#include "kdb_internal.h"
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <map>

namespace {

constexpr size_t kChunkSize = 64ull * 1024 * 1024;
constexpr size_t kHeader = 64;
constexpr uint32_t kMagic = 0x4B414F4Eu;
constexpr uint32_t kUnallocated = 0xFFFFFFFFu;

struct Chunk {
    const unsigned char *data = nullptr;
    size_t size = 0;
};

struct ArenaFiles {
    std::map<uint32_t, Chunk> chunks;
    ~ArenaFiles() {
        for (auto &kv : chunks)
            if (kv.second.data) munmap(const_cast<unsigned char *>(kv.second.data), kv.second.size);
    }
    // maps chunk `id` (validating its header) or returns null with the error set
    const Chunk *get(const char *dir, uint32_t id, uint32_t dim, uint32_t precision) {
        auto it = chunks.find(id);
        if (it != chunks.end()) return &it->second;
        char path[4096];
        snprintf(path, sizeof path, "%s/arena_%04u.bin", dir, id);
        int fd = open(path, O_RDONLY);
        if (fd < 0) {
            kdb_set_error("arena: cannot open %s", path);
            return nullptr;
        }
        struct stat st;
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < kHeader) {
            close(fd);
            kdb_set_error("arena: %s is shorter than its 64-byte header", path);
            return nullptr;
        }
        void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) {
            kdb_set_error("arena: mmap of %s failed", path);
            return nullptr;
        }
        const unsigned char *d = static_cast<const unsigned char *>(p);
        uint32_t magic, version, fdim;
        memcpy(&magic, d, 4);
        memcpy(&version, d + 4, 4);
        memcpy(&fdim, d + 8, 4);
        const uint8_t fprec = d[12];
        if (magic != kMagic || version != 1 || fdim != dim || fprec != precision) {
            munmap(p, (size_t)st.st_size);
            kdb_set_error("arena: %s header mismatch (magic %08x version %u dim %u precision %u; expected dim %u precision %u)",
                          path, magic, version, fdim, fprec, dim, precision);
            return nullptr;
        }
        Chunk c;
        c.data = d;
        c.size = (size_t)st.st_size;
        return &chunks.emplace(id, c).first->second;
    }
};

int read_rows(ArenaFiles &af, const char *dir, uint32_t dim, uint32_t precision, const uint32_t *slot_table,
              uint32_t first_id, uint32_t n, unsigned char *out) {
    const size_t elem = precision == KDB_PREC_F32 ? 4 : precision == KDB_PREC_F16 ? 2 : 1;
    const size_t vsize = (size_t)dim * elem;
    const size_t vpc = (kChunkSize - kHeader) / vsize;
    if (vpc == 0) {
        kdb_set_error("arena: vector size %zu exceeds chunk payload capacity", vsize);
        return KDB_ERR_INVALID;
    }
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t id = first_id + i;
        const uint32_t slot = slot_table ? slot_table[id] : id - 1;
        unsigned char *dst = out + (size_t)i * vsize;
        if (slot == kUnallocated) { // id never stored (or freed): all-zero row
            memset(dst, 0, vsize);
            continue;
        }
        const uint32_t chunk = (uint32_t)(slot / vpc);
        const size_t off = kHeader + (size_t)(slot % vpc) * vsize;
        const Chunk *c = af.get(dir, chunk, dim, precision);
        if (!c) return KDB_ERR_INVALID;
        if (off + vsize > c->size) {
            kdb_set_error("arena: slot %u of id %u lies beyond the end of chunk %u", slot, id, chunk);
            return KDB_ERR_INVALID;
        }
        memcpy(dst, c->data + off, vsize);
    }
    return KDB_OK;
}

} // namespace

extern "C" int kdb_arena_read_rows(const char *dir, uint32_t dim, uint32_t precision, const uint32_t *slot_table,
                                   uint32_t first_id, uint32_t n, void *out_rows) {
    if (!dir || !out_rows || dim == 0 || precision > KDB_PREC_I8 || first_id == 0) {
        kdb_set_error("arena_read_rows: bad argument");
        return KDB_ERR_INVALID;
    }
    ArenaFiles af;
    return read_rows(af, dir, dim, precision, slot_table, first_id, n, static_cast<unsigned char *>(out_rows));
}

extern "C" int kdb_index_upload_arena(kdb_index *idx, const char *dir, const uint32_t *slot_table, uint32_t count) {
    if (!idx || !dir) {
        kdb_set_error("upload_arena: null argument");
        return KDB_ERR_INVALID;
    }
    if (count == 0) return KDB_OK;
    if (count > idx->cap) {
        kdb_set_error("upload_arena: %u ids exceed capacity %u", count, idx->cap);
        return KDB_ERR_INVALID;
    }
    ArenaFiles af;
    const size_t vsize = (size_t)idx->desc.dim * idx->elem;
    const uint32_t piece = (uint32_t)std::max<size_t>(1, (32ull << 20) / vsize); // ~32 MiB staging pieces
    std::vector<unsigned char> stage((size_t)piece * vsize);
    for (uint32_t first = 1; first <= count; first += piece) {
        const uint32_t n = std::min<uint32_t>(piece, count - first + 1);
        int rc = read_rows(af, dir, idx->desc.dim, idx->desc.precision, slot_table, first, n, stage.data());
        if (rc) return rc;
        rc = kdb_index_upload_rows(idx, first, n, stage.data());
        if (rc) return rc;
    }
    return KDB_OK;
}
