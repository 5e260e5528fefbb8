// Syntax check of the Arnold glue of rl_arnold_stub.hpp (the block behind RLS_STUB_WITH_ARNOLD) against the mock
// declarations in tests/native/mock_arnold/ai.h: one translation unit per node would read like this.
#define RLS_STUB_WITH_ARNOLD
#include "rl_arnold_stub.hpp"

namespace ggx_tu { RLS_STUB_NODE_PARAMETERS(rlstub::kGgx) }
namespace disney_tu { RLS_STUB_NODE_PARAMETERS(rlstub::kDisney) }
namespace skin_tu { RLS_STUB_NODE_PARAMETERS(rlstub::kSkin) }
RLS_STUB_NODE_LOADER

// shader_evaluate's per-point part: the evaluator reads the node's parameters through the positional ids
static void evaluate_ggx(AtShaderGlobals *sg, AtNode *node, rlstub::GgxNode &batch, const float U[3], float rx, float ry)
{
    rlstub::Globals g = {};
    const float *src[3] = {&sg->Rd.x, &sg->N.x, &sg->Nf.x};
    float *dst[3] = {g.Rd, g.N, g.Nf};
    for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) dst[v][k] = src[v][k];
    for (int k = 0; k < 3; k++) g.U[k] = U[k];
    batch.add(g, rlstub::ArnoldEval{sg, node}, rx, ry);
}
