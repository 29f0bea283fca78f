// frame_pipeline.hpp -- the frame pipeline of the ROS2 node (ros2/recursive_patchwork_node.cpp), ROS-free so that it can be
// tested without rclcpp (tests/cpp/test_frame_pipeline.cpp runs it on a scripted processor).
//
// The reference node calls filterGroundPoints once per message and publishes its result before the callback returns
// (reference src/recursive_patchwork_node.cpp:61-108). With `depth` frames in flight a callback submits its frame to the
// next slot of a ring and hands on the frame that slot held, i.e. the OLDEST one in flight: results leave in submission
// order, depth - 1 callbacks late. A frame larger than the engine's capacity cannot be submitted while other frames are in
// flight (the engine re-allocates its device buffers to grow): the pipeline then hands on every frame in flight, oldest
// first, lets the engine grow, and submits -- so a short first scan followed by full scans neither throws nor stalls.
#pragma once

#include <cstddef>
#include <utility>
#include <vector>

#include "recursive_patchwork.hpp"

namespace recursive_patchwork {

template <class Processor, class Tag>
class BasicFramePipeline {
public:
    using Result = std::pair<std::vector<Point3D>, std::vector<Point3D>>;

    // `rp` must outlive the pipeline; its pipeline depth is set to `depth` (>= 1)
    BasicFramePipeline(Processor& rp, unsigned depth) : rp_(rp), depth_(depth ? depth : 1u), tags_(depth_), busy_(depth_, false)
    {
        rp_.setPipelineDepth(depth_);
    }

    // Submits one frame given as PointCloud2 record bytes; emit(result, tag) is called for every frame that leaves the
    // pipeline because of it, oldest first (none while the ring fills up). If emit or the processor throws, the frame
    // being handled is dropped and the pipeline stays consistent: the next push works.
    template <class Emit>
    void push(const unsigned char* data, size_t n, unsigned point_step, unsigned off_x, unsigned off_y, unsigned off_z,
              const Tag& tag, Emit&& emit)
    {
        if (n > rp_.capacity()) {              // the engine has to grow: nothing may be in flight while it does
            drain(emit);
            rp_.reserve(n);
        }
        const unsigned slot = static_cast<unsigned>(seq_ % depth_);
        if (busy_[slot]) { busy_[slot] = false; emit(rp_.waitFrame(slot), tags_[slot]); }
        rp_.submitPointCloud2(slot, data, n, point_step, off_x, off_y, off_z);
        tags_[slot] = tag; busy_[slot] = true; ++seq_;
    }

    // Hands on every frame in flight, oldest first.
    template <class Emit>
    void drain(Emit&& emit)
    {
        for (unsigned k = 0; k < depth_; ++k) {
            const unsigned slot = static_cast<unsigned>((seq_ + k) % depth_);      // the ring's next slot holds the oldest frame
            if (busy_[slot]) { busy_[slot] = false; emit(rp_.waitFrame(slot), tags_[slot]); }
        }
    }

    size_t framesInFlight() const { size_t c = 0; for (bool b : busy_) c += b; return c; }
    unsigned depth() const { return depth_; }

private:
    Processor& rp_;
    unsigned depth_;
    unsigned long long seq_ = 0;
    std::vector<Tag> tags_;
    std::vector<bool> busy_;
};

} // namespace recursive_patchwork
