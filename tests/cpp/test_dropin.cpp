// Drop-in check of the C++ surface: the calls the reference's own test makes
// (reference test/test_recursive_patchwork.cpp:51-98,146-164), against
// recursive_patchwork::RecursivePatchwork backed by the HIP library. Seeded instead of random_device.
#include "recursive_patchwork.hpp"
#include "ordered_pipeline.hpp"
#include "rpw_oracle.h"      // the CPU oracle: the checker (tests only)

#include <cassert>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <random>
#include <vector>

using namespace recursive_patchwork;

static std::vector<Point3D> synthetic(size_t n, unsigned seed)
{
    std::vector<Point3D> pts; pts.reserve(n);
    std::mt19937 gen(seed);
    std::normal_distribution<float> gz(0.0f, 0.05f);
    std::uniform_real_distribution<float> gxy(-50.0f, 50.0f), oxy(-30.0f, 30.0f), oz(0.5f, 3.0f);
    const size_t ng = static_cast<size_t>(n * 0.7);
    for (size_t i = 0; i < ng; ++i) { Point3D p; p.x = gxy(gen); p.y = gxy(gen); p.z = gz(gen); pts.push_back(p); }
    for (size_t i = ng; i < n; ++i) { Point3D p; p.x = oxy(gen); p.y = oxy(gen); p.z = oz(gen); pts.push_back(p); }
    return pts;
}

int main()
{
    // testBasicFunctionality
    auto points = synthetic(5000, 1);
    PatchworkConfig config;
    config.sensor_height = 1.2f; config.filtering_radius = 50.0f; config.num_sectors = 8; config.max_iter = 50;
    RecursivePatchwork patchwork(config);
    auto [ground, non_ground] = patchwork.filterGroundPoints(points);
    assert(ground.size() + non_ground.size() <= points.size());
    assert(ground.size() > 0);
    assert(non_ground.size() > 0);
    std::printf("basic: %zu ground, %zu non-ground\n", ground.size(), non_ground.size());

    // label view of the same call must agree with the two vectors
    std::vector<float> res;
    auto labels = patchwork.segment(points, &res);
    size_t g = 0;
    for (unsigned char l : labels) g += (l == 1);
    assert(g == ground.size());
    assert(res.size() == points.size());

    // non-finite points are dropped, points beyond the radius come last in non_ground
    points.push_back(Point3D(NAN, 0, 0));
    points.push_back(Point3D(500.0f, 0, 0));
    auto r2 = patchwork.filterGroundPoints(points);
    assert(r2.first.size() + r2.second.size() == points.size() - 1);
    assert(r2.second.back().x == 500.0f);
    assert(patchwork.cleanPoints(points).size() == points.size() - 1);

    // setConfig / getConfig, default config, bigger cloud (capacity growth), timing like testPerformance
    RecursivePatchwork def;
    auto big = synthetic(300000, 7);
    Timer t;
    auto r3 = def.filterGroundPoints(big);
    std::printf("perf: %zu points in %.3f ms\n", big.size(), t.elapsed() * 1e3);
    assert(r3.first.size() > 0 && r3.first.size() + r3.second.size() == big.size());
    PatchworkConfig c2 = def.getConfig(); c2.num_sectors = 12; def.setConfig(c2);
    assert(def.getConfig().num_sectors == 12);
    auto r4 = def.filterGroundPoints(big);
    assert(r4.first.size() + r4.second.size() == big.size());

    // PointCloud2 bytes (16-byte records) give the same pair as the vector interface
    {
        std::vector<unsigned char> bytes(points.size() * 16, 0x5A);
        for (size_t i = 0; i < points.size(); ++i) std::memcpy(&bytes[16 * i], &points[i].x, 12);
        auto r6 = patchwork.filterGroundPointCloud2(bytes.data(), points.size(), 16, 0, 4, 8);
        auto r7 = patchwork.filterGroundPoints(points);
        assert(r6.first.size() == r7.first.size() && r6.second.size() == r7.second.size());
        for (size_t i = 0; i < r6.first.size(); ++i) assert(std::memcmp(&r6.first[i], &r7.first[i], 12) == 0);
    }

    // fusion pre-step on the device == rotating / ego-filtering on the host and calling filterGroundPoints
    {
        std::vector<std::vector<Point3D>> clouds = {synthetic(4000, 21), synthetic(3000, 22), synthetic(2000, 23)};
        std::vector<LidarConfig> cfgs = {{1, "/lidar_front", 0.0f, 2.5f}, {2, "/lidar_left", 120.0f, 2.5f}, {3, "/lidar_right", -120.0f, 2.5f}};
        auto fusedPair = patchwork.fuseAndFilterGroundPoints(clouds, cfgs);
        std::vector<Point3D> fused;
        for (size_t k = 0; k < clouds.size(); ++k) {
            float a = cfgs[k].rotation_angle * M_PI / 180.0f; const float c = std::cos(a), s = std::sin(a);
            for (auto p : clouds[k]) {
                if (std::abs(cfgs[k].rotation_angle) > 1e-6f) { const float x = p.x * c - p.y * s, y = p.x * s + p.y * c; p.x = x; p.y = y; }
                if (!(std::sqrt(p.x * p.x + p.y * p.y) <= cfgs[k].ego_radius)) fused.push_back(p);
            }
        }
        auto hostPair = patchwork.filterGroundPoints(fused);
        assert(fusedPair.first.size() == hostPair.first.size() && fusedPair.second.size() == hostPair.second.size());
        for (size_t i = 0; i < hostPair.first.size(); ++i) assert(std::memcmp(&fusedPair.first[i], &hostPair.first[i], 12) == 0);
        for (size_t i = 0; i < hostPair.second.size(); ++i) assert(std::memcmp(&fusedPair.second[i], &hostPair.second[i], 12) == 0);
        std::printf("fusion: %zu ground, %zu non-ground\n", fusedPair.first.size(), fusedPair.second.size());
    }

    // sampleGroundAndObstacles (reference test: testEnhancedFiltering) and rotatePoints2D
    {
        auto pts = synthetic(30000, 5);
        RecursivePatchwork pw;
        pw.setSampleSeed(7);
        auto parts = pw.filterGroundPoints(pts);
        auto s1 = pw.sampleGroundAndObstacles(pts, 1.1f, 0.5f);
        auto s2 = pw.sampleGroundAndObstacles(pts, 1.1f, 0.5f);
        assert(s1.size() == s2.size() && s1.size() > 0 && s1.size() <= pts.size());
        for (size_t i = 0; i < s1.size(); ++i) assert(std::memcmp(&s1[i], &s2[i], 12) == 0);     // seeded => reproducible
        size_t n_obs = 0;
        for (const auto& p : parts.second) if (std::sqrt(p.x * p.x + p.y * p.y) > 2.5f && std::abs(p.z - 1.1f) <= 0.5f) ++n_obs;
        const size_t n_ground_sample = std::min<size_t>(2000, parts.first.size());
        assert(s1.size() == n_ground_sample + n_obs);
        for (size_t i = n_ground_sample; i < s1.size(); ++i) assert(std::abs(s1[i].z - 1.1f) <= 0.5f);
        auto rot = pw.rotatePoints2D({{1.f, 0.f, 3.f}}, 90.0f);
        assert(std::abs(rot[0].x) < 1e-6f && std::abs(rot[0].y - 1.f) < 1e-6f && rot[0].z == 3.f);
    }

    // sampleGroundAndObstacles against the oracle's restatement (recursive_patchwork.cpp:428-465 +
    // point_cloud_processor.cpp:121-146, seeded mt19937 + rejection loop): element for element
    {
        const float heights[2] = {1.1f, 0.3f};
        for (int round = 0; round < 2; ++round) {
            auto pts = synthetic(round == 0 ? 30000 : 2500, 11 + round);      // the small cloud has fewer than 2000 ground points
            pts.push_back(Point3D(1.0f, 1.0f, 1.2f));                          // inside the ego radius, in the height band: must go
            PatchworkConfig cfg;
            RecursivePatchwork pw(cfg);
            pw.setSampleSeed(1234u + round);
            auto got = pw.sampleGroundAndObstacles(pts, heights[round], 0.5f);
            rpwo_config oc = {cfg.sensor_height, cfg.max_range, cfg.num_sectors, cfg.max_iter, cfg.adaptive_seed_height ? 1 : 0,
                              cfg.th_seeds, cfg.th_dist, cfg.th_outlier, cfg.filtering_radius, cfg.max_split_depth};
            std::vector<float> want(pts.size() * 3);
            uint32_t n_want = 0;
            const int rc = rpwo_sample_ground_and_obstacles(&oc, &pts[0].x, (uint32_t)pts.size(), heights[round], 0.5f, 1234u + round, want.data(), &n_want);
            assert(rc == 0);
            assert(got.size() == n_want && n_want > 0);
            for (size_t i = 0; i < got.size(); ++i) assert(std::memcmp(&got[i], &want[3 * i], 12) == 0);
            std::printf("sample: %u points equal to the oracle's\n", n_want);
        }
    }

    // pipeline: frames in flight give the same pairs as the synchronous call, whatever the order of the waits;
    // a copy of the object is an engine of its own with the same config
    {
        std::vector<std::vector<Point3D>> fr = {synthetic(20000, 31), synthetic(50000, 32), synthetic(777, 33), synthetic(120000, 34), {}};
        PatchworkConfig cfg; cfg.num_sectors = 12;
        RecursivePatchwork pw(cfg);
        pw.reserve(120000);
        pw.setPipelineDepth(3);
        RecursivePatchwork ref = pw;                                           // copy: own handle, same config
        assert(ref.getConfig().num_sectors == 12 && ref.pipelineDepth() == 3);
        const unsigned order[5] = {2, 0, 1, 1, 0};
        size_t next = 0, checked = 0;
        std::vector<int> in_slot(3, -1);
        for (unsigned s = 0; s < 3 && next < fr.size(); ++s) { pw.submitFrame(s, fr[next]); in_slot[s] = (int)next++; }
        for (unsigned k = 0; checked < fr.size(); ++k) {
            unsigned s = order[k % 5];
            while (in_slot[s] < 0) s = (s + 1) % 3;
            auto got = pw.waitFrame(s);
            auto want = ref.filterGroundPoints(fr[(size_t)in_slot[s]]);
            assert(got.first.size() == want.first.size() && got.second.size() == want.second.size());
            for (size_t i = 0; i < want.first.size(); ++i) assert(std::memcmp(&got.first[i], &want.first[i], 12) == 0);
            for (size_t i = 0; i < want.second.size(); ++i) assert(std::memcmp(&got.second[i], &want.second[i], 12) == 0);
            ++checked; in_slot[s] = -1;
            if (next < fr.size()) { pw.submitFrame(s, fr[next]); in_slot[s] = (int)next++; }
        }
        bool threw = false;
        try { (void)pw.waitFrame(0); } catch (const std::exception&) { threw = true; }
        assert(threw);                                                         // nothing in flight
        std::printf("pipeline: %zu frames equal to the synchronous call\n", checked);
    }

    // ordered pipeline: results come back in submission order and equal the synchronous call, frame by frame,
    // although the slots finish in any order (frames of very different cost)
    {
        PatchworkConfig cfg; cfg.num_sectors = 10;
        RecursivePatchwork pw(cfg), ref(cfg);
        OrderedPipeline pipe(pw, 4, 130000);
        std::vector<std::vector<Point3D>> fr;
        const int sizes[] = {120000, 500, 60000, 3, 0, 90000, 2049, 130000, 700, 45000, 10, 100000};
        for (int k = 0; k < 36; ++k) fr.push_back(synthetic(sizes[k % 12], 100 + k));
        size_t got_n = 0;
        auto check = [&](std::vector<OrderedPipeline::Result>&& rs) {
            for (auto& got : rs) {
                auto want = ref.filterGroundPoints(fr[got_n]);
                assert(got.first.size() == want.first.size() && got.second.size() == want.second.size());
                for (size_t i = 0; i < want.first.size(); ++i) assert(std::memcmp(&got.first[i], &want.first[i], 12) == 0);
                for (size_t i = 0; i < want.second.size(); ++i) assert(std::memcmp(&got.second[i], &want.second[i], 12) == 0);
                ++got_n;
            }
        };
        for (auto& f : fr) { check(pipe.push(f)); assert(pipe.framesOnGpu() <= 4); }
        check(pipe.flush());
        assert(got_n == fr.size() && pipe.framesOnGpu() == 0 && pipe.finishedFramesHeld() == 0);
        std::printf("ordered pipeline: %zu frames in order, equal to the synchronous call (at most %zu finished frames held)\n",
                    got_n, pipe.maxFinishedFramesHeld());
    }

    // empty input: both empty
    auto r5 = def.filterGroundPoints({});
    assert(r5.first.empty() && r5.second.empty());
    std::printf("drop-in OK\n");
    return 0;
}
