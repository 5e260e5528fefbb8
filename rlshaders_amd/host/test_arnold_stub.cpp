// test_arnold_stub.cpp -- drives rl_arnold_stub.hpp without Arnold (tests/test_arnold_stub.py, tests/test_gpu_arnold_stub.py).
//   test_arnold_stub shade <in> <out>     the whole shader_evaluate of the three nodes (addShade per point, shade per batch)
//   test_arnold_stub decl                 replay node_parameters / node_loader of the three nodes through a recording
//                                         host and print what was declared, as JSON
//   test_arnold_stub run <in> <out>       read shading points + parameter values, push them through GgxNode, DisneyNode and
//                                         SkinNode with a table-lookup evaluator (the shape of AiShaderEvalParam*), write
//                                         the planes the three flushes return
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rl_arnold_stub.hpp"

namespace {

struct Recorder {                      // the host policy: records instead of calling AiParameter* / AiMetaDataSet*
    std::string json, cur_name, cur_type, cur_def, cur_meta;
    bool have = false;
    void begin(const char *name, const char *type, const std::string &def)
    {
        flush();
        cur_name = name; cur_type = type; cur_def = def; cur_meta.clear(); have = true;
    }
    void meta(const std::string &kv) { cur_meta += (cur_meta.empty() ? "" : ", ") + kv; }
    void flush()
    {
        if (!have) return;
        json += (json.empty() ? "" : ", ");
        json += "{\"name\": \"" + cur_name + "\", \"type\": \"" + cur_type + "\", \"default\": [" + cur_def + "], \"meta\": {" +
                cur_meta + "}}";
        have = false;
    }
    static std::string num(float v) { char b[64]; std::snprintf(b, sizeof b, "%.9g", v); return b; }
    void parameterRGB(const char *n, float r, float g, float b) { begin(n, "RGB", num(r) + ", " + num(g) + ", " + num(b)); }
    void parameterFLT(const char *n, float v) { begin(n, "FLT", num(v)); }
    void parameterVec(const char *n, float x, float y, float z) { begin(n, "VEC", num(x) + ", " + num(y) + ", " + num(z)); }
    void parameterBool(const char *n, bool b) { begin(n, "BOOL", b ? "true" : "false"); }
    void parameterSTR(const char *n, const char *s) { begin(n, "STR", std::string("\"") + s + "\""); }
    void metaBool(const char *, const char *k, bool b) { meta(std::string("\"") + k + "\": " + (b ? "true" : "false")); }
    void metaFlt(const char *, const char *k, float v) { meta(std::string("\"") + k + "\": " + num(v)); }
    void metaInt(const char *, const char *k, const char *sym) { meta(std::string("\"") + k + "\": \"" + sym + "\""); }
};

struct NodeLibRec { const char *methods; const char *output_type; const char *name; const char *node_type; char version[32]; };
struct ResolveRec {
    const char *methods(const char *s) const { return s; }
    const char *type(const char *s) const { return s; }
};

std::vector<float> read_file(const char *path);

// host-only: detail::Materials::find_rows on K columns of n floats (file: n, K, then the columns) -> n ids and, behind them, the
// table; prints {"by_reference": ..., "count": ...}.  No device is touched.
int rows(const char *in, const char *out)
{
    std::vector<float> d = read_file(in);
    const size_t n = (size_t)d[0], K = (size_t)d[1];
    std::vector<rlstub::detail::Col> cols(K);
    for (size_t k = 0; k < K; k++) cols[k].v[0].assign(d.begin() + 2 + (std::ptrdiff_t)(k * n), d.begin() + 2 + (std::ptrdiff_t)((k + 1) * n));
    rlstub::detail::Materials mt;
    for (size_t k = 0; k < K; k++) mt.add1(cols[k]);
    std::vector<uint32_t> id;
    const bool by_ref = mt.find_rows(id);
    std::FILE *f = std::fopen(out, "wb");
    if (!f) { std::perror(out); return 1; }
    if (by_ref) {
        std::fwrite(id.data(), 4, id.size(), f);
        for (size_t k = 0; k < K; k++) std::fwrite(mt.table[k].data(), 4, mt.table[k].size(), f);
    }
    std::fclose(f);
    std::printf("{\"by_reference\": %s, \"count\": %u}\n", by_ref ? "true" : "false", mt.count);
    return 0;
}

int decl()
{
    std::printf("{\"nodes\": {");
    for (int i = 0; i < rlstub::kNodeCount; i++) {
        Recorder r;
        rlstub::declare_node(i, r);
        r.flush();
        std::string en;
        for (int k = 0; k < rlstub::kNodes[i].count; k++)
            en += std::string(k ? ", " : "") + "\"" + rlstub::kNodes[i].params[k].enumerator + "\"";
        // the enumerators are positional: id k must be k
        for (int k = 0; k < rlstub::kNodes[i].count; k++)
            if (rlstub::kNodes[i].params[k].id != k) { std::fprintf(stderr, "enumerator %d of node %d is not positional\n", k, i); return 1; }
        std::printf("%s\"%s\": {\"parameters\": [%s], \"enum\": [%s], \"maya.id\": \"%s\"}", i ? ", " : "", rlstub::kNodes[i].name,
                    r.json.c_str(), en.c_str(), rlstub::kNodes[i].maya_id);
    }
    std::printf("}, \"node_loader\": [");
    for (int i = 0;; i++) {
        NodeLibRec nl = {};
        if (!rlstub::load_node(i, &nl, ResolveRec(), "4.2.11.0")) {
            if (i != rlstub::kNodeCount) { std::fprintf(stderr, "node_loader stopped at %d\n", i); return 1; }
            break;
        }
        std::printf("%s{\"id\": %d, \"methods\": \"%s\", \"output_type\": \"%s\", \"name\": \"%s\", \"node_type\": \"%s\", \"version\": \"%s\"}",
                    i ? ", " : "", i, nl.methods, nl.output_type, nl.name, nl.node_type, nl.version);
    }
    std::printf("]}\n");
    return 0;
}

// parameter values per shading point, by positional id: what AiShaderEvalParam{Flt,RGB,Vec}(pid) would return at point i
struct TableEval {
    const float *const *rows; int64_t n, i;      // rows[pid * 3 + c]
    float flt(int pid) const { return rows[pid * 3][i]; }
    void rgb(int pid, float out[3]) const { for (int c = 0; c < 3; c++) out[c] = rows[pid * 3 + c][i]; }
    void vec(int pid, float out[3]) const { rgb(pid, out); }
};

std::vector<float> read_file(const char *path)
{
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(1); }
    std::fseek(f, 0, SEEK_END);
    long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)bytes / 4);
    if (std::fread(v.data(), 4, v.size(), f) != v.size()) { std::perror("read"); std::exit(1); }
    std::fclose(f);
    return v;
}

// file layout (float32): [0] = n, then rows of n floats:
//   12 rows Rd3 N3 Nf3 U3 | 6 rows xi | 3 * ggx::p_count rows rlGgx values by pid | 3 * disney::p_count | 3 * skin::p_count
int run(const char *in, const char *out)
{
    std::vector<float> d = read_file(in);
    const int64_t n = (int64_t)d[0];
    const float *p = d.data() + 1;
    auto take = [&](int rows) { const float *q = p; p += (size_t)rows * (size_t)n; return q; };
    const float *geo = take(12), *xi = take(6);
    const float *tg = take(3 * rlstub::ggx::p_count), *td = take(3 * rlstub::disney::p_count), *ts = take(3 * rlstub::skin::p_count);
    auto rows_of = [&](const float *base, int count) {
        std::vector<const float *> r((size_t)count * 3);
        for (int k = 0; k < count * 3; k++) r[(size_t)k] = base + (size_t)k * (size_t)n;
        return r;
    };
    std::vector<const float *> rg = rows_of(tg, rlstub::ggx::p_count), rd = rows_of(td, rlstub::disney::p_count),
                               rs = rows_of(ts, rlstub::skin::p_count);
    try {
        rlsb::Device dev(0);
        rlstub::GgxNode ggx;
        rlstub::DisneyNode disney;
        rlstub::SkinNode skin;
        for (int64_t i = 0; i < n; i++) {                         // what shader_evaluate does, once per shading point
            rlstub::Globals sg = {};
            for (int c = 0; c < 3; c++) {
                sg.Rd[c] = geo[(size_t)c * n + i]; sg.N[c] = geo[(size_t)(3 + c) * n + i];
                sg.Nf[c] = geo[(size_t)(6 + c) * n + i]; sg.U[c] = geo[(size_t)(9 + c) * n + i];
            }
            float x[6];
            for (int k = 0; k < 6; k++) x[k] = xi[(size_t)k * n + i];
            ggx.add(sg, TableEval{rg.data(), n, i}, x[0], x[1]);
            disney.add(sg, TableEval{rd.data(), n, i}, x[0], x[1]);
            skin.add(sg, TableEval{rs.data(), n, i}, x);
        }
        std::vector<float> a = ggx.flush(dev), b = disney.flush(dev), c = skin.flush(dev);
        if (ggx.size() != 0 || disney.size() != 0 || skin.size() != 0) { std::fprintf(stderr, "flush left points behind\n"); return 1; }
        std::FILE *f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 1; }
        std::fwrite(a.data(), 4, a.size(), f); std::fwrite(b.data(), 4, b.size(), f); std::fwrite(c.data(), 4, c.size(), f);
        std::fclose(f);
        std::printf("{\"n\": %lld, \"ggx_planes\": %zu, \"disney_planes\": %zu, \"skin_planes\": %zu, \"uniform_parameters\": %lld, \"reference_batches\": %lld}\n",
                    (long long)n, a.size() / (size_t)n, b.size() / (size_t)n, c.size() / (size_t)n,
                    (long long)rlstub::detail::uniform_parameters().load(), (long long)rlstub::detail::reference_batches().load());
        return 0;
    } catch (const rlsb::Error &e) {
        std::fprintf(stderr, "rlshaders_amd: %s\n", e.what());
        return e.status == RLS_ERR_NO_DEVICE ? 2 : 1;
    }
}

// the whole shader_evaluate of the three nodes: file layout as `run` with 15 geometry rows (Rd3 N3 Nf3 U3 P3) and no xi
int shade(const char *in, const char *out)
{
    std::vector<float> d = read_file(in);
    const int64_t n = (int64_t)d[0];
    const float *p = d.data() + 1;
    auto take = [&](int rows) { const float *q = p; p += (size_t)rows * (size_t)n; return q; };
    const float *geo = take(15);
    const float *tg = take(3 * rlstub::ggx::p_count), *td = take(3 * rlstub::disney::p_count), *ts = take(3 * rlstub::skin::p_count);
    auto rows_of = [&](const float *base, int count) {
        std::vector<const float *> r((size_t)count * 3);
        for (int k = 0; k < count * 3; k++) r[(size_t)k] = base + (size_t)k * (size_t)n;
        return r;
    };
    std::vector<const float *> rg = rows_of(tg, rlstub::ggx::p_count), rd = rows_of(td, rlstub::disney::p_count),
                               rs = rows_of(ts, rlstub::skin::p_count);
    // the scene the test's oracle call uses: two spherical lights, a uniform environment, the unit sphere for the probes
    rls_sphere_light lights[2] = {{{2.0f, 2.0f, 3.0f}, 1.25f, {3.0f, 2.0f, 1.0f}, RLS_MIS_BOTH},
                                  {{-3.0f, 1.0f, 2.5f}, 0.5f, {0.5f, 4.0f, 2.0f}, RLS_MIS_BSDF_ONLY}};
    const float env[3] = {0.7f, 0.8f, 0.9f};
    rls_sss_scene scene = {};
    scene.geometry = RLS_SCENE_SPHERE; scene.sphere_radius = 1.0f;
    scene.light_dir[1] = 0.6f; scene.light_dir[2] = 0.8f;
    scene.light_color[0] = scene.light_color[1] = scene.light_color[2] = 1.0f;
    scene.use_cavity_fade = 1;
    try {
        rlsb::Device dev(0);
        rlstub::GgxNode ggx;
        rlstub::DisneyNode disney;
        rlstub::SkinNode skin;
        for (int64_t i = 0; i < n; i++) {
            rlstub::Globals sg = {};
            for (int c = 0; c < 3; c++) {
                sg.Rd[c] = geo[(size_t)c * n + i]; sg.N[c] = geo[(size_t)(3 + c) * n + i];
                sg.Nf[c] = geo[(size_t)(6 + c) * n + i]; sg.U[c] = geo[(size_t)(9 + c) * n + i];
                sg.P[c] = geo[(size_t)(12 + c) * n + i];
            }
            ggx.addShade(sg, TableEval{rg.data(), n, i});
            disney.addShade(sg, TableEval{rd.data(), n, i});
            skin.addShade(sg, TableEval{rs.data(), n, i});
        }
        std::vector<float> a = ggx.shade(dev, lights, 2, env, true, 3, 77u), b = disney.shade(dev, lights, 2, env, 3, 77u),
                           c = skin.shade(dev, scene, lights, 1, env, 3, 77u);
        if (ggx.size() != 0 || disney.size() != 0 || skin.size() != 0) { std::fprintf(stderr, "shade left points behind\n"); return 1; }
        std::FILE *f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 1; }
        std::fwrite(a.data(), 4, a.size(), f); std::fwrite(b.data(), 4, b.size(), f); std::fwrite(c.data(), 4, c.size(), f);
        std::fclose(f);
        std::printf("{\"n\": %lld, \"ggx_planes\": %zu, \"disney_planes\": %zu, \"skin_planes\": %zu, \"uniform_parameters\": %lld, \"reference_batches\": %lld}\n",
                    (long long)n, a.size() / (size_t)n, b.size() / (size_t)n, c.size() / (size_t)n,
                    (long long)rlstub::detail::uniform_parameters().load(), (long long)rlstub::detail::reference_batches().load());
        return 0;
    } catch (const rlsb::Error &e) {
        std::fprintf(stderr, "rlshaders_amd: %s\n", e.what());
        return e.status == RLS_ERR_NO_DEVICE ? 2 : 1;
    }
}

} // namespace

int main(int argc, char **argv)
{
    if (argc >= 2 && !std::strcmp(argv[1], "decl")) return decl();
    if (argc >= 4 && !std::strcmp(argv[1], "rows")) return rows(argv[2], argv[3]);
    if (argc >= 4 && !std::strcmp(argv[1], "run")) return run(argv[2], argv[3]);
    if (argc >= 4 && !std::strcmp(argv[1], "shade")) return shade(argv[2], argv[3]);
    std::fprintf(stderr, "usage: test_arnold_stub decl | run <in> <out> | shade <in> <out>\n");
    return 64;
}
