// CPU test of the reorder logic of cpp/ordered_pipeline.hpp with a scripted stand-in for the engine: slots finish in an
// order that has nothing to do with the submission order, results must still come out in submission order, every frame
// exactly once, with never more than `slots` frames "on the GPU".
#include <cassert>
#include <cstdio>
#include <vector>
#include "ordered_pipeline.hpp"

using namespace recursive_patchwork;

// A fake engine: a frame submitted to a slot is "done" after a scripted number of frameDone() polls of that slot.
struct FakeEngine {
    struct S { bool busy = false; int polls_left = 0; float tag = 0.0f; };
    std::vector<S> slots;
    unsigned seed = 12345;
    size_t submitted = 0, max_busy = 0;
    void reserve(size_t) {}
    void setPipelineDepth(unsigned d) { slots.assign(d, S()); }
    void submitFrame(unsigned s, const std::vector<Point3D>& pts)
    {
        assert(s < slots.size() && !slots[s].busy);
        seed = seed * 1664525u + 1013904223u;
        slots[s].busy = true; slots[s].polls_left = (int)((seed >> 16) % 23); slots[s].tag = pts.empty() ? -1.0f : pts[0].x;
        ++submitted;
        size_t b = 0; for (auto& x : slots) b += x.busy;
        if (b > max_busy) max_busy = b;
    }
    bool frameDone(unsigned s) { assert(slots[s].busy); return slots[s].polls_left-- <= 0; }
    std::pair<std::vector<Point3D>, std::vector<Point3D>> waitFrame(unsigned s)
    {
        assert(slots[s].busy);
        slots[s].busy = false;
        return {{Point3D(slots[s].tag, 0, 0)}, {}};             // the frame's tag comes back as its "ground"
    }
};

int main()
{
    for (unsigned nslots : {1u, 2u, 5u, 8u}) {
        FakeEngine eng;
        BasicOrderedPipeline<FakeEngine> pipe(eng, nslots, 1000);
        const int N = 500;
        int next = 0;
        auto check = [&](std::vector<BasicOrderedPipeline<FakeEngine>::Result>&& rs) {
            for (auto& r : rs) { assert(r.first.size() == 1 && r.first[0].x == (float)next); ++next; }
        };
        for (int k = 0; k < N; ++k) {
            check(pipe.push({Point3D((float)k, 0, 0)}));
            assert(pipe.framesOnGpu() <= nslots);
        }
        check(pipe.flush());
        assert(next == N && eng.submitted == (size_t)N && eng.max_busy <= nslots);
        assert(pipe.framesOnGpu() == 0 && pipe.finishedFramesHeld() == 0);
        std::printf("slots %u: %d frames in order, at most %zu finished frames held\n", nslots, N, pipe.maxFinishedFramesHeld());
    }
    {   // a capped reorder buffer: one frame that takes "for ever" must not let the buffer grow past the cap
        FakeEngine eng;
        BasicOrderedPipeline<FakeEngine> pipe(eng, 4, 1000, /*max_held*/ 6);
        int next = 0;
        for (int k = 0; k < 200; ++k) {
            auto rs = pipe.push({Point3D((float)k, 0, 0)});
            if (k == 10) for (auto& sl : eng.slots) if (sl.busy && sl.tag == 10.0f) sl.polls_left = 1 << 20;      // frame 10 hangs around
            for (auto& r : rs) { assert(r.first[0].x == (float)next); ++next; }
            assert(pipe.finishedFramesHeld() <= 6);
        }
        for (auto& r : pipe.flush()) { assert(r.first[0].x == (float)next); ++next; }
        assert(next == 200 && pipe.maxFinishedFramesHeld() <= 6);
        std::printf("capped buffer: at most %zu finished frames held\n", pipe.maxFinishedFramesHeld());
    }
    std::printf("ordered_pipeline OK\n");
    return 0;
}
