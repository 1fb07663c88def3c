// build.hip -- GPU batched graph construction (placeholder until the builder lands).
#include "kdb_internal.h"
int kdb_build_graph(kdb_index *idx, uint32_t count, const kdb_build_params *p) {
    (void)idx; (void)count; (void)p;
    kdb_set_error("kdb_index_build: not implemented yet");
    return KDB_ERR_UNSUPPORTED;
}
