// CPU test of cpp/frame_pipeline.hpp (the ROS2 node's frame pipeline) with a scripted stand-in for the engine that has
// RecursivePatchwork's capacity rule: growing is refused while frames are in flight. A short first scan followed by full
// scans must neither throw nor lose or reorder a frame.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "frame_pipeline.hpp"

using namespace recursive_patchwork;

struct ScriptedEngine {
    struct S { bool busy = false; size_t n = 0; float tag = 0.0f; };
    std::vector<S> slots;
    size_t cap = 0, grown = 0, max_busy = 0;
    bool fail_next_wait = false;
    size_t capacity() const { return cap; }
    size_t pending() const { size_t c = 0; for (auto& s : slots) c += s.busy; return c; }
    void ensure(size_t n)
    {
        if (n <= cap) return;
        if (pending()) throw std::runtime_error("a frame larger than the capacity arrived while frames are in flight");
        size_t c = cap ? cap : 1024; while (c < n) c *= 2;
        cap = c; ++grown;
    }
    void reserve(size_t n) { ensure(n); }
    void setPipelineDepth(unsigned d) { assert(!pending()); slots.assign(d, S()); }
    void submitPointCloud2(unsigned s, const unsigned char* data, size_t n, unsigned step, unsigned ox, unsigned, unsigned)
    {
        assert(s < slots.size() && !slots[s].busy);
        ensure(n);
        float x = -1.0f; if (n) std::memcpy(&x, data + ox, 4);
        (void)step;
        slots[s].busy = true; slots[s].n = n; slots[s].tag = x;
        if (pending() > max_busy) max_busy = pending();
    }
    std::pair<std::vector<Point3D>, std::vector<Point3D>> waitFrame(unsigned s)
    {
        assert(s < slots.size() && slots[s].busy);
        slots[s].busy = false;
        if (fail_next_wait) { fail_next_wait = false; throw std::runtime_error("scripted GPU error"); }
        return {{Point3D(slots[s].tag, (float)slots[s].n, 0)}, {}};      // the frame's tag and size come back as its "ground"
    }
};

static std::vector<unsigned char> frame(float tag, size_t n)
{
    std::vector<unsigned char> d(n * 16, 0);
    for (size_t i = 0; i < n; ++i) std::memcpy(d.data() + i * 16, &tag, 4);
    return d;
}

int main()
{
    for (unsigned depth : {1u, 2u, 4u, 8u}) {
        ScriptedEngine eng;
        BasicFramePipeline<ScriptedEngine, int> pipe(eng, depth);
        std::vector<int> out_tags; std::vector<float> out_x; std::vector<size_t> out_n;
        auto emit = [&](const std::pair<std::vector<Point3D>, std::vector<Point3D>>& r, int tag) {
            out_tags.push_back(tag); out_x.push_back(r.first[0].x); out_n.push_back((size_t)r.first[0].y);
        };
        // sizes: a short first scan, then full scans, a still larger one in the middle of a full pipeline, small ones again
        std::vector<size_t> sizes = {300, 120000, 120000, 119000, 121000, 120500, 500000, 120000, 10, 120000, 130000, 2000000, 7, 7, 7};
        for (size_t k = 0; k < sizes.size(); ++k) {
            const auto d = frame((float)k, sizes[k]);
            pipe.push(d.data(), sizes[k], 16, 0, 4, 8, (int)k, emit);
            assert(pipe.framesInFlight() <= depth && eng.max_busy <= depth);
            assert(out_tags.size() + pipe.framesInFlight() == k + 1);           // nothing lost, nothing published twice
        }
        pipe.drain(emit);
        assert(out_tags.size() == sizes.size() && pipe.framesInFlight() == 0);
        for (size_t k = 0; k < sizes.size(); ++k)
            assert(out_tags[k] == (int)k && out_x[k] == (float)k && out_n[k] == sizes[k]);       // submission order, own tag with own result
        assert(eng.cap >= 2000000 && eng.grown >= 3);
        std::printf("depth %u: %zu frames in order, engine grew %zu times\n", depth, sizes.size(), eng.grown);
    }
    {   // an error while a frame is taken out: that frame is lost, the pipeline goes on and keeps the order of the others
        ScriptedEngine eng;
        BasicFramePipeline<ScriptedEngine, int> pipe(eng, 3);
        std::vector<int> out;
        auto emit = [&](const std::pair<std::vector<Point3D>, std::vector<Point3D>>&, int tag) { out.push_back(tag); };
        int thrown = 0;
        for (int k = 0; k < 12; ++k) {
            const auto d = frame((float)k, 1000);
            if (k == 5) eng.fail_next_wait = true;
            try { pipe.push(d.data(), 1000, 16, 0, 4, 8, k, emit); } catch (const std::exception&) { ++thrown; }
        }
        pipe.drain(emit);
        assert(thrown == 1);
        // frame 2 (waited at k = 5) failed and frame 5's submission was abandoned with it; every other frame came out in order
        const std::vector<int> want = {0, 1, 3, 4, 6, 7, 8, 9, 10, 11};
        assert(out == want);
        std::printf("error path: %zu of 12 frames delivered in order after one failed wait\n", out.size());
    }
    std::printf("frame_pipeline OK\n");
    return 0;
}
