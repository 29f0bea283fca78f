// recursive_patchwork.hpp -- drop-in header for the hot path of the reference package:
// same namespace, type names, member names and signatures as the reference's
// include/recursive_patchwork.hpp:18-67 for Point3D, PatchworkConfig and the parts of
// RecursivePatchwork that sit on the path (filterGroundPoints, cleanPoints, setConfig,
// getConfig). The public API is Eigen-free (the reference only uses Eigen in private
// members, recursive_patchwork.hpp:73-81). The work is done on an MI355X through the
// C-ABI in include/rpw.h; there is no CPU fallback -- construction throws
// std::runtime_error when no device / library is usable.
#pragma once

#include <chrono>
#include <memory>
#include <string>
#include <utility>
#include <vector>

struct rpw_handle;

namespace recursive_patchwork {

struct Point3D {
    float x, y, z;
    Point3D() : x(0), y(0), z(0) {}
    Point3D(float x, float y, float z) : x(x), y(y), z(z) {}
};

struct PatchworkConfig {
    float sensor_height = 1.2f;
    float max_range = 150.0f;
    int num_sectors = 10;
    int max_iter = 100;
    bool adaptive_seed_height = true;
    float th_seeds = 0.15f;
    float th_dist = 0.2f;
    float th_outlier = 0.08f;
    float filtering_radius = 150.0f;
    int max_split_depth = 1000;
};

// LiDAR configuration, as the reference's (recursive_patchwork.hpp:39-44)
struct LidarConfig {
    int lidar_id;
    std::string topic_name;
    float rotation_angle;  // degrees
    float ego_radius = 2.5f;
};

class RecursivePatchwork {
public:
    RecursivePatchwork(const PatchworkConfig& config = PatchworkConfig{});
    ~RecursivePatchwork();
    // Copyable like the reference's class (which holds only its config, recursive_patchwork.hpp:70): a copy gets the
    // config, the device and the pipeline depth, and an engine of its own (device buffers are never shared).
    RecursivePatchwork(const RecursivePatchwork& other);
    RecursivePatchwork& operator=(const RecursivePatchwork& other);
    RecursivePatchwork(RecursivePatchwork&& other) noexcept;
    RecursivePatchwork& operator=(RecursivePatchwork&& other) noexcept;

    // {ground_points, non_ground_points}, reference order (recursive_patchwork.cpp:402-425)
    std::pair<std::vector<Point3D>, std::vector<Point3D>>
    filterGroundPoints(const std::vector<Point3D>& points);

    // Build-side addition: the same call as a label array (one rpw label per input point) and,
    // optionally, per-point residuals. BASELINE.json calls this segment(PointCloud&).
    std::vector<unsigned char> segment(const std::vector<Point3D>& points, std::vector<float>* residuals = nullptr);

    // Build-side addition for the ROS2 node: the same call on the bytes of a sensor_msgs/PointCloud2 message
    // (n records of point_step bytes, FLOAT32 x/y/z at the given byte offsets). The records go to the GPU as
    // they are; the returned pair is rebuilt from the labels in the reference's order.
    std::pair<std::vector<Point3D>, std::vector<Point3D>>
    filterGroundPointCloud2(const unsigned char* data, size_t n, unsigned point_step,
                            unsigned off_x, unsigned off_y, unsigned off_z);

    // Build-side addition: the same call with up to `depth` frames in flight (C-ABI rpw_submit / rpw_wait). A frame is
    // submitted into a free slot and returns at once; waitFrame(slot) blocks until that frame is done and returns the
    // pair filterGroundPoints would have returned for it. The frame is copied into the slot's pinned buffer at
    // submission, so the caller's vector / message may be released right away. Slots finish independently.
    void setPipelineDepth(unsigned depth);
    unsigned pipelineDepth() const { return depth_; }
    void reserve(size_t max_points);                       // capacity of every slot (frames must not grow past it while frames are in flight)
    size_t capacity() const { return capacity_; }          // points a frame may have without the engine re-allocating (0: no engine yet)
    void submitFrame(unsigned slot, const std::vector<Point3D>& points);
    void submitPointCloud2(unsigned slot, const unsigned char* data, size_t n, unsigned point_step,
                           unsigned off_x, unsigned off_y, unsigned off_z);
    bool frameDone(unsigned slot);
    std::pair<std::vector<Point3D>, std::vector<Point3D>> waitFrame(unsigned slot);

    // Build-side addition: LidarFusion::fuseLidarPointClouds (reference src/lidar_fusion.cpp:42-86) followed by
    // filterGroundPoints in ONE device pass: the per-sensor yaw rotation and ego-radius removal run in the
    // binning kernel's load stage. Returns the same pair filterGroundPoints(fuse(clouds)) would.
    std::pair<std::vector<Point3D>, std::vector<Point3D>>
    fuseAndFilterGroundPoints(const std::vector<std::vector<Point3D>>& lidar_point_clouds,
                              const std::vector<LidarConfig>& lidar_configs);

    // Post-step of the reference (recursive_patchwork.cpp:428-465): filterGroundPoints, then on the host: ego-radius
    // removal (2.5 m) of the non-ground points, height band |z - target_height| <= base_tol, plus up to 2000 ground
    // points drawn without replacement. The reference seeds that draw from std::random_device; so does this class
    // unless setSampleSeed() was called (reproducible runs and tests).
    std::vector<Point3D> sampleGroundAndObstacles(const std::vector<Point3D>& points,
                                                  float target_height = 1.1f, float base_tol = 0.5f);
    void setSampleSeed(unsigned seed) { sample_seed_ = seed; have_sample_seed_ = true; }

    std::vector<Point3D> cleanPoints(const std::vector<Point3D>& points);
    std::vector<Point3D> rotatePoints2D(const std::vector<Point3D>& points, float angle_degrees);

    void setConfig(const PatchworkConfig& config);
    const PatchworkConfig& getConfig() const { return config_; }

    // which HIP device new instances bind to (default 0); per-thread
    static void setDevice(int device);

private:
    void ensure(size_t n);
    PatchworkConfig config_;
    rpw_handle* handle_ = nullptr;
    size_t capacity_ = 0;
    int device_ = 0;
    unsigned n_zone_last_ = 0;
    unsigned sample_seed_ = 0;
    bool have_sample_seed_ = false;
    unsigned depth_ = 0;
    struct SlotInfo { size_t n = 0; unsigned step = 12, off[3] = {0, 4, 8}; bool pending = false; };
    std::vector<SlotInfo> slots_;
};

// same utility as the reference (recursive_patchwork.hpp:90-105)
class Timer {
public:
    Timer() : start_(std::chrono::high_resolution_clock::now()) {}
    double elapsed() const {
        auto end = std::chrono::high_resolution_clock::now();
        return std::chrono::duration<double>(end - start_).count();
    }
    void reset() { start_ = std::chrono::high_resolution_clock::now(); }
private:
    std::chrono::high_resolution_clock::time_point start_;
};

} // namespace recursive_patchwork
