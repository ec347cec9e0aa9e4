// 13-bit windows (20 of them, 2^12 buckets: the small-circuit configuration): every templated kernel of the bucket MSM instantiated for
// MsmCfg<13> (msm_kernels.hip.h).
#include "msm_kernels.hip.h"
namespace bbg {
template int msm_run_c<13>(bbg_ctx*, const Srs&, const void*, int, const void* const*, const size_t*, const size_t*, void*, hipStream_t, const void*);
template int srs_build_tables_c<13>(const void*, size_t, void*, hipStream_t);
} // namespace bbg
