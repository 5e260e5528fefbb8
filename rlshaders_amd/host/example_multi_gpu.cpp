// example_multi_gpu.cpp -- the sharded host runtime in C++: one host thread, one context (own stream) and one plane
// arena per shard, contiguous index ranges, no communication on the data path.  With `shards` <= visible devices
// every shard gets its own GPU (the 8 x MI355X configuration); with more shards than devices several threads share
// a device, which is how the test drives it on a single-GPU box -- and what checks that contexts are independent.
//
//   example_multi_gpu <log2 points> <shards>
// prints one JSON line: per-shard checksums of every output plane of the GGX reflect+refract pass, their sum, and the
// checksum of the same job run as ONE shard; the two totals must agree (the checksum is order-independent).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rls_batch.hpp"

namespace {
constexpr uint32_t kSeed = 1234;   // stream ids: DESIGN.md "Synthetic inputs"

uint64_t run_shard(int device, int64_t first, int64_t n, int candidates)
{
    rlsb::Device dev(device);
    rlsb::Arena A(dev, n, 31, candidates);
    rls_context *ctx = dev.ctx();
    rlsb::check(rls_gen_frame(ctx, kSeed, (uint64_t)first, n, A.vec3(0), A.vec3(3), A.vec3(6)));
    for (int j = 0; j < 3; j++) rlsb::check(rls_gen_uniform(ctx, kSeed, (uint64_t)first, n, 8 + j, 0.0f, 1.0f, A.plane(9 + j)));
    rlsb::check(rls_gen_uniform(ctx, kSeed, (uint64_t)first, n, 5, 0.05f, 1.0f, A.plane(12)));     // roughness
    rlsb::check(rls_gen_uniform(ctx, kSeed, (uint64_t)first, n, 6, 1.05f, 2.55f, A.plane(13)));    // ior
    rlsb::check(rls_gen_aniso(ctx, kSeed, (uint64_t)first, n, A.plane(14)));
    for (int j = 0; j < 4; j++) rlsb::check(rls_gen_uniform(ctx, kSeed, (uint64_t)first, n, 11 + j, 0.0f, 1.0f, A.plane(15 + j)));
    rls_ggx_closure c = {};
    c.wo = A.cvec3(0); c.N = A.cvec3(3); c.T = A.cvec3(6);
    c.KsColor = rls_param_rgb{A.plane(9), A.plane(10), A.plane(11), 0, 0, 0};
    c.specularRoughness = rls_param{A.plane(12), 0}; c.ior = rls_param{A.plane(13), 0};
    c.anisotropic = rls_param{A.plane(14), 0};
    rlsb::check(rls_ggx_reflect_refract(ctx, n, &c, A.plane(15), A.plane(16), A.plane(17), A.plane(18), A.vec3(19),
                                        A.rgb(22), A.plane(25), A.plane(26), A.vec3(27), A.plane(30)));
    uint64_t sum = 0;
    for (int k = 19; k < 31; k++) {
        uint64_t v = 0;
        rlsb::check(rls_checksum(ctx, n, A.plane(k), &v));
        sum += v;
    }
    return sum;
}
} // namespace

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? std::atoi(argv[1]) : 20;
    const int shards = argc > 2 ? std::atoi(argv[2]) : 2;
    const int64_t total = (int64_t)1 << log2n;
    try {
        const int devices = rlsb::deviceCount();
        if (devices < 1) { std::fprintf(stderr, "rlshaders_amd: no HIP device\n"); return 2; }
        std::vector<uint64_t> sums((size_t)shards, 0);
        std::vector<std::string> errors((size_t)shards);
        std::vector<std::thread> pool;
        for (int r = 0; r < shards; r++) {
            pool.emplace_back([&, r] {
                try {
                    const rlsb::Shard s = rlsb::shardRange(total, r, shards);
                    sums[(size_t)r] = run_shard(r % devices, s.first, s.count, 2);
                } catch (const std::exception &e) { errors[(size_t)r] = e.what(); }
            });
        }
        for (auto &t : pool) t.join();
        for (const auto &e : errors) if (!e.empty()) { std::fprintf(stderr, "shard failed: %s\n", e.c_str()); return 1; }
        uint64_t sharded = 0;
        for (uint64_t v : sums) sharded += v;
        const uint64_t whole = run_shard(0, 0, total, 1);
        std::printf("{\"points\": %lld, \"shards\": %d, \"devices\": %d, \"sharded_checksum\": %llu, \"single_checksum\": %llu}\n",
                    (long long)total, shards, devices, (unsigned long long)sharded, (unsigned long long)whole);
        return sharded == whole ? 0 : 1;
    } catch (const rlsb::Error &e) {
        std::fprintf(stderr, "rlshaders_amd: %s\n", e.what());
        return e.status == RLS_ERR_NO_DEVICE ? 2 : 1;
    }
}
