// temporary stubs until melspec.hip / campplus.hip land
#include "common.h"
extern "C" {
void mv_melspec_default_cfg(MvMelSpecCfg* cfg) { memset(cfg, 0, sizeof(*cfg)); }
int mv_melspec_create(const MvMelSpecCfg*, MvMelSpec**) { return mv::fail(MV_ERR_UNSUPPORTED, "melspec: not built yet"); }
int mv_melspec_destroy(MvMelSpec*) { return MV_OK; }
int mv_melspec_num_frames(const MvMelSpec*, int64_t, int64_t*) { return mv::fail(MV_ERR_UNSUPPORTED, "melspec: not built yet"); }
size_t mv_melspec_workspace_bytes(const MvMelSpec*, int32_t, int64_t) { return 0; }
int mv_melspec_forward(const MvMelSpec*, const float*, int32_t, int64_t, int64_t, const float*, float*, void*, size_t, mv_stream_t) { return mv::fail(MV_ERR_UNSUPPORTED, "melspec: not built yet"); }
}
