// ordered_pipeline.hpp -- frames in, {ground, non_ground} pairs out IN SUBMISSION ORDER, with up to `slots` frames on the
// GPU (RecursivePatchwork::submitFrame / frameDone / waitFrame, i.e. rpw_submit / rpw_poll / rpw_wait).
//
// Every caller of the reference hands over one frame and gets one result (filterGroundPoints,
// reference src/recursive_patchwork_node.cpp:91); a node that publishes must keep the order. Waiting on the slots in
// submission order does that but runs at slots / (latency of the slowest frame): a frame's time is set by the refit chain
// of its largest patch and varies 1 : 5 between HDL-64 frames, so every result queues behind the slowest frame ahead of it
// while finished slots sit idle. Here a finished slot is emptied into a host-side reorder buffer and refilled at once; the
// results leave the buffer in submission order. Same slots, same order, the throughput of the any-order loop.
#pragma once

#include <chrono>
#include <map>
#include <thread>
#include <utility>
#include <vector>

#include "recursive_patchwork.hpp"

namespace recursive_patchwork {

// (A template on the processor type so that the reorder logic can be tested on a CPU with a scripted stand-in,
// tests/cpp/test_ordered_pipeline.cpp; OrderedPipeline below is the instance for RecursivePatchwork.)
template <class Processor>
class BasicOrderedPipeline {
public:
    using Result = std::pair<std::vector<Point3D>, std::vector<Point3D>>;

    // `rp` must outlive the pipeline; its pipeline depth is set to `slots`, every slot sized for `max_points`
    // At most `max_held` finished frames wait in the reorder buffer (default 4 x slots): beyond that push() blocks on the
    // oldest frame in flight instead of refilling slots, so one pathological frame cannot grow the buffer without bound.
    BasicOrderedPipeline(Processor& rp, unsigned slots, size_t max_points, size_t max_held = 0)
        : rp_(rp), seq_of_(slots, kFree), cap_(max_held ? max_held : 4 * (size_t)slots)
    {
        rp_.reserve(max_points);
        rp_.setPipelineDepth(slots);
    }

    // Submits a frame (blocks only while every slot is busy) and returns the results that have become deliverable, oldest
    // first -- possibly none, possibly several.
    std::vector<Result> push(const std::vector<Point3D>& points)
    {
        std::vector<Result> out;
        collect();
        deliver(out);
        while (held_.size() >= cap_) {                   // the buffer is full of frames queued behind the oldest one: wait for THAT one
            for (size_t s = 0; s < seq_of_.size(); ++s)
                if (seq_of_[s] == next_out_) { held_.emplace(seq_of_[s], rp_.waitFrame((unsigned)s)); seq_of_[s] = kFree; break; }
            deliver(out);
        }
        int slot = free_slot();
        while (slot < 0) {                               // every slot busy: wait for whichever finishes first
            std::this_thread::sleep_for(std::chrono::microseconds(20));       // (a frame takes milliseconds; no busy spin)
            collect();
            deliver(out);
            slot = free_slot();
        }
        rp_.submitFrame((unsigned)slot, points);
        seq_of_[(size_t)slot] = next_in_++;
        deliver(out);
        return out;
    }

    // Waits for everything in flight and returns the remaining results in order.
    std::vector<Result> flush()
    {
        std::vector<Result> out;
        for (size_t s = 0; s < seq_of_.size(); ++s)
            if (seq_of_[s] != kFree) { held_.emplace(seq_of_[s], rp_.waitFrame((unsigned)s)); seq_of_[s] = kFree; }
        deliver(out);
        return out;
    }

    size_t framesOnGpu() const { size_t n = 0; for (auto q : seq_of_) n += q != kFree; return n; }
    size_t finishedFramesHeld() const { return held_.size(); }
    size_t maxFinishedFramesHeld() const { return max_held_; }

private:
    static constexpr unsigned long long kFree = ~0ull;
    void collect()                                        // empty every finished slot into the reorder buffer
    {
        for (size_t s = 0; s < seq_of_.size(); ++s)
            if (seq_of_[s] != kFree && (seq_of_[s] == next_out_ || held_.size() < cap_) && rp_.frameDone((unsigned)s)) {
                held_.emplace(seq_of_[s], rp_.waitFrame((unsigned)s));
                seq_of_[s] = kFree;
            }
    }
    int free_slot() const
    {
        for (size_t s = 0; s < seq_of_.size(); ++s) if (seq_of_[s] == kFree) return (int)s;
        return -1;
    }
    void deliver(std::vector<Result>& out)
    {
        for (auto it = held_.find(next_out_); it != held_.end(); it = held_.find(next_out_)) {
            out.push_back(std::move(it->second));
            held_.erase(it);
            ++next_out_;
        }
        if (held_.size() > max_held_) max_held_ = held_.size();      // frames that really wait (behind an older one)
    }
    Processor& rp_;
    std::vector<unsigned long long> seq_of_;              // slot -> sequence number of the frame it holds
    std::map<unsigned long long, Result> held_;           // finished, not yet deliverable
    unsigned long long next_in_ = 0, next_out_ = 0;
    size_t max_held_ = 0;
    size_t cap_;
};

using OrderedPipeline = BasicOrderedPipeline<RecursivePatchwork>;

} // namespace recursive_patchwork
