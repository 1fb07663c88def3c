// search_inst.hip -- one translation unit of hnsw_search_kernel instantiations: compiled once per
// (KDB_INST_PREC, KDB_INST_METRIC, KDB_INST_GROUP) by the Makefile; kdb_launch_search (search.hip) routes to these.
#include "search_kernel.cuh"

#define KDB_CAT4_(a, b, c, d) a##b##_##c##_##d
#define KDB_CAT4(a, b, c, d) KDB_CAT4_(a, b, c, d)

int KDB_CAT4(kdb_launch_search_inst_, KDB_INST_PREC, KDB_INST_METRIC, KDB_INST_GROUP)(KDB_LAUNCH_SEARCH_PARAMS) {
    static const bool force_generic = KDB_AB_ENV("KDB_SEARCH_GENERIC") != nullptr; // measurement knob
    (void)force_generic;
#if KDB_INST_PREC == 0
#if KDB_INST_GROUP == 0
    if (v.ld <= 128) return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 2>(KDB_LAUNCH_SEARCH_ARGS); // 65 .. 128 columns (kdb_piece_ok)
    switch (v.ld) {
    case 256: return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 4>(KDB_LAUNCH_SEARCH_ARGS);
    case 384: return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 6>(KDB_LAUNCH_SEARCH_ARGS);
    default: return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 8>(KDB_LAUNCH_SEARCH_ARGS);
    }
#elif KDB_INST_GROUP == 1
    if (v.ld == 768) return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 12>(KDB_LAUNCH_SEARCH_ARGS);
    return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 16>(KDB_LAUNCH_SEARCH_ARGS);
#else
    if (v.ld == 1536 && !force_generic) return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 24>(KDB_LAUNCH_SEARCH_ARGS);
    return launch_search_t<KDB_PREC_F32, KDB_INST_METRIC, 0>(KDB_LAUNCH_SEARCH_ARGS);
#endif
#else // float16 (squared L2) / int8 (cosine): unrolled for 768 and 1536 columns, generic otherwise
    switch (v.ld) {
    case 768: return launch_search_t<KDB_INST_PREC, KDB_INST_METRIC, 12>(KDB_LAUNCH_SEARCH_ARGS);
    case 1536: return launch_search_t<KDB_INST_PREC, KDB_INST_METRIC, 24>(KDB_LAUNCH_SEARCH_ARGS);
    default: return launch_search_t<KDB_INST_PREC, KDB_INST_METRIC, 0>(KDB_LAUNCH_SEARCH_ARGS);
    }
#endif
}
