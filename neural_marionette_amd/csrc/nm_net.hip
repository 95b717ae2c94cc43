// placeholder — replaced by the network graph
#include "nm_ctx.h"
int nm_net_set_weights(nm_ctx*, const std::map<std::string, std::pair<const float*, int64_t>>&) { nm_set_error("set_weights: not built yet"); return NM_ERR_STATE; }
#define STUB(name, ...) int name(__VA_ARGS__) { nm_set_error(#name ": not built yet"); return NM_ERR_STATE; }
extern "C" {
size_t nm_workspace_bytes(nm_ctx*, int32_t, int32_t) { return 0; }
STUB(nm_detector_forward, nm_ctx*, const float*, int32_t, int32_t, int32_t, float*, float*, float*, float*, float*, float*)
STUB(nm_decode_from_keypoints, nm_ctx*, const float*, const float*, const float*, int32_t, int32_t, float*)
STUB(nm_get_affinity, nm_ctx*, float*)
STUB(nm_vrnn_set_tree, nm_ctx*, const int32_t*, const int32_t*)
STUB(nm_vrnn_offsets, nm_ctx*, const float*, int32_t, int32_t, float*)
STUB(nm_vrnn_encode, nm_ctx*, const float*, const float*, int32_t, int32_t, int32_t, float*, float*, float*, float*, float*, int32_t*)
STUB(nm_vrnn_generate, nm_ctx*, const float*, const float*, const float*, int32_t, int32_t, int32_t, int32_t, float*, float*, float*)
STUB(nm_vrnn_step, nm_ctx*, int32_t, const float*, const float*, const float*, const float*, int32_t, int32_t, float*, float*, float*)
STUB(nm_vrnn_mlp, nm_ctx*, int32_t, const float*, int32_t, float*)
STUB(nm_vrnn_gru, nm_ctx*, const float*, const float*, int32_t, float*)
STUB(nm_vrnn_fk, nm_ctx*, const float*, const float*, int32_t, float*, float*)
}
