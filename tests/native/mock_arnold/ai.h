// MOCK of the handful of Arnold 4.x SDK declarations the RLS_STUB_WITH_ARNOLD block of rl_arnold_stub.hpp touches.
// Test infrastructure for ONE purpose: tests/test_arnold_stub.py compiles that block with -fsyntax-only, so that the
// glue a maintainer pastes into a real plugin is at least well-formed C++ against declarations of this shape (public
// SDK documentation).  Nothing here is linked, run, shipped or used to build the reference.
#pragma once

struct AtList;
struct AtMetaDataStore;
struct AtNode;
struct AtNodeMethods;
struct AtRGB { float r, g, b; };
struct AtVector { float x, y, z; };
struct AtShaderGlobals { AtVector Rd, N, Nf, Ns, dPdu, P; };
struct AtNodeLib { const AtNodeMethods *methods; int output_type; const char *name; int node_type; char version[32]; };

#define AI_TYPE_RGB 5
#define AI_NODE_SHADER 0x0010
#define AI_VERSION "4.2.11.0"

// the SDK's declaration macros expand over the in-scope `params` / `mds` (node_parameters) and `sg` / `node` (shader_evaluate)
void AiNodeParamRGB(AtList *, int, const char *, float, float, float);
void AiNodeParamFlt(AtList *, int, const char *, float);
void AiNodeParamVec(AtList *, int, const char *, float, float, float);
void AiNodeParamBool(AtList *, int, const char *, bool);
void AiNodeParamStr(AtList *, int, const char *, const char *);
#define AiParameterRGB(n, r, g, b) AiNodeParamRGB(params, -1, n, r, g, b)
#define AiParameterFLT(n, v) AiNodeParamFlt(params, -1, n, v)
#define AiParameterVec(n, x, y, z) AiNodeParamVec(params, -1, n, x, y, z)
#define AiParameterBool(n, b) AiNodeParamBool(params, -1, n, b)
#define AiParameterSTR(n, s) AiNodeParamStr(params, -1, n, s)
bool AiMetaDataSetBool(AtMetaDataStore *, const char *, const char *, bool);
bool AiMetaDataSetFlt(AtMetaDataStore *, const char *, const char *, float);
bool AiMetaDataSetInt(AtMetaDataStore *, const char *, const char *, int);
float AiShaderEvalParamFuncFlt(AtShaderGlobals *, const AtNode *, int);
AtRGB AiShaderEvalParamFuncRGB(AtShaderGlobals *, const AtNode *, int);
AtVector AiShaderEvalParamFuncVec(AtShaderGlobals *, const AtNode *, int);
#define AiShaderEvalParamFlt(pid) AiShaderEvalParamFuncFlt(sg, node, pid)
#define AiShaderEvalParamRGB(pid) AiShaderEvalParamFuncRGB(sg, node, pid)
#define AiShaderEvalParamVec(pid) AiShaderEvalParamFuncVec(sg, node, pid)

#define node_parameters static void Parameters(AtList *params, AtMetaDataStore *mds)
#define node_loader bool NodeLoader(int i, AtNodeLib *node)
