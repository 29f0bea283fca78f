// Host side of the drop-in class: converts between the reference's vector<Point3D> interface
// and the C-ABI (include/rpw.h). No arithmetic of the path happens here.
#include "recursive_patchwork.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <stdexcept>

#include "rpw.h"

namespace recursive_patchwork {

namespace {
thread_local int g_device = 0;

rpw_config to_c(const PatchworkConfig& c)
{
    rpw_config r;
    r.sensor_height = c.sensor_height; r.max_range = c.max_range; r.num_sectors = c.num_sectors;
    r.max_iter = c.max_iter; r.adaptive_seed_height = c.adaptive_seed_height ? 1 : 0; r.th_seeds = c.th_seeds;
    r.th_dist = c.th_dist; r.th_outlier = c.th_outlier; r.filtering_radius = c.filtering_radius;
    r.max_split_depth = c.max_split_depth;
    return r;
}
static_assert(sizeof(Point3D) == 3 * sizeof(float), "Point3D must be the packed AoS layout the ABI takes");
} // namespace

void RecursivePatchwork::setDevice(int device) { g_device = device; }

RecursivePatchwork::RecursivePatchwork(const PatchworkConfig& config) : config_(config), device_(g_device)
{
    ensure(1u << 18);
}

RecursivePatchwork::~RecursivePatchwork()
{
    if (handle_) rpw_destroy(handle_);
}

RecursivePatchwork::RecursivePatchwork(const RecursivePatchwork& o)
    : config_(o.config_), device_(o.device_), sample_seed_(o.sample_seed_), have_sample_seed_(o.have_sample_seed_)
{
    ensure(o.capacity_ ? o.capacity_ : (1u << 18));
    setPipelineDepth(o.depth_);
}

RecursivePatchwork& RecursivePatchwork::operator=(const RecursivePatchwork& o)
{
    if (this == &o) return *this;
    RecursivePatchwork tmp(o);
    *this = std::move(tmp);
    return *this;
}

RecursivePatchwork::RecursivePatchwork(RecursivePatchwork&& o) noexcept { *this = std::move(o); }

RecursivePatchwork& RecursivePatchwork::operator=(RecursivePatchwork&& o) noexcept
{
    if (this == &o) return *this;
    if (handle_) rpw_destroy(handle_);
    config_ = o.config_; handle_ = o.handle_; capacity_ = o.capacity_; device_ = o.device_; n_zone_last_ = o.n_zone_last_;
    sample_seed_ = o.sample_seed_; have_sample_seed_ = o.have_sample_seed_; depth_ = o.depth_; slots_ = std::move(o.slots_);
    o.handle_ = nullptr; o.capacity_ = 0; o.depth_ = 0; o.slots_.clear();
    return *this;
}

void RecursivePatchwork::ensure(size_t n)
{
    if (handle_ && n <= capacity_) return;
    for (const SlotInfo& si : slots_)
        if (si.pending) throw std::runtime_error("RecursivePatchwork: a frame larger than the capacity arrived while frames are in flight; "
                                                 "call reserve() before submitting");
    size_t cap = capacity_ ? capacity_ : (1u << 18);
    while (cap < n) cap *= 2;
    if (handle_) { rpw_destroy(handle_); handle_ = nullptr; }
    const rpw_config c = to_c(config_);
    rpw_status st = rpw_create(&c, device_, (uint32_t)cap, 1, &handle_);
    if (st != RPW_OK) {
        handle_ = nullptr;
        throw std::runtime_error(std::string("RecursivePatchwork: rpw_create failed: ") + rpw_last_error(nullptr));
    }
    capacity_ = cap;
    if (depth_ && rpw_set_pipeline(handle_, depth_) != RPW_OK)
        throw std::runtime_error(std::string("RecursivePatchwork: rpw_set_pipeline failed: ") + rpw_last_error(handle_));
}

void RecursivePatchwork::reserve(size_t max_points) { ensure(max_points); }

void RecursivePatchwork::setPipelineDepth(unsigned depth)
{
    for (const SlotInfo& si : slots_) if (si.pending) throw std::runtime_error("RecursivePatchwork::setPipelineDepth: frames are in flight");
    if (handle_ && rpw_set_pipeline(handle_, depth) != RPW_OK)
        throw std::runtime_error(std::string("RecursivePatchwork: rpw_set_pipeline failed: ") + rpw_last_error(handle_));
    depth_ = depth;
    slots_.assign(depth, SlotInfo());
}

void RecursivePatchwork::submitFrame(unsigned slot, const std::vector<Point3D>& points)
{
    if (slot >= depth_) throw std::runtime_error("RecursivePatchwork::submitFrame: no such slot (setPipelineDepth)");
    ensure(points.size());
    const rpw_status rc = rpw_submit(handle_, slot, points.empty() ? nullptr : &points[0].x, (uint32_t)points.size(), 0);
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_submit failed: ") + rpw_last_error(handle_));
    SlotInfo& si = slots_[slot];
    si.n = points.size(); si.step = 12; si.off[0] = 0; si.off[1] = 4; si.off[2] = 8; si.pending = true;
}

void RecursivePatchwork::submitPointCloud2(unsigned slot, const unsigned char* data, size_t n, unsigned point_step,
                                           unsigned off_x, unsigned off_y, unsigned off_z)
{
    if (slot >= depth_) throw std::runtime_error("RecursivePatchwork::submitPointCloud2: no such slot (setPipelineDepth)");
    ensure(n);
    const rpw_status rc = rpw_submit_pointcloud2(handle_, slot, data, (uint32_t)n, point_step, off_x, off_y, off_z, 0);
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_submit_pointcloud2 failed: ") + rpw_last_error(handle_));
    SlotInfo& si = slots_[slot];
    si.n = n; si.step = point_step; si.off[0] = off_x; si.off[1] = off_y; si.off[2] = off_z; si.pending = true;
}

bool RecursivePatchwork::frameDone(unsigned slot)
{
    if (slot >= depth_ || !slots_[slot].pending) throw std::runtime_error("RecursivePatchwork::frameDone: the slot holds no frame");
    int done = 0;
    if (rpw_poll(handle_, slot, &done) != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_poll failed: ") + rpw_last_error(handle_));
    return done != 0;
}

std::pair<std::vector<Point3D>, std::vector<Point3D>> RecursivePatchwork::waitFrame(unsigned slot)
{
    if (slot >= depth_ || !slots_[slot].pending) throw std::runtime_error("RecursivePatchwork::waitFrame: the slot holds no frame");
    SlotInfo& si = slots_[slot];
    std::vector<unsigned char> labels(si.n);
    rpw_frame_stats st;
    const rpw_status rc = rpw_wait(handle_, slot, labels.data(), nullptr, &st);
    si.pending = false;
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_wait failed: ") + rpw_last_error(handle_));
    n_zone_last_ = st.n_zone;
    // the frame as it was submitted still sits in the slot's pinned buffer
    const unsigned char* data = reinterpret_cast<const unsigned char*>(rpw_slot_input(handle_, slot));
    auto point = [&](size_t i) {
        Point3D p; const unsigned char* r = data + i * si.step;
        std::memcpy(&p.x, r + si.off[0], 4); std::memcpy(&p.y, r + si.off[1], 4); std::memcpy(&p.z, r + si.off[2], 4);
        return p;
    };
    std::vector<Point3D> ground, non_ground;
    if (st.n_zone < 3) {
        for (size_t i = 0; i < si.n; ++i) if (labels[i] != RPW_LABEL_DROPPED) non_ground.push_back(point(i));
        return {ground, non_ground};
    }
    for (size_t i = 0; i < si.n; ++i) {
        if (labels[i] == RPW_LABEL_GROUND) ground.push_back(point(i));
        else if (labels[i] == RPW_LABEL_NONGROUND) non_ground.push_back(point(i));
    }
    for (size_t i = 0; i < si.n; ++i) if (labels[i] == RPW_LABEL_OUTSIDE) non_ground.push_back(point(i));
    return {ground, non_ground};
}

void RecursivePatchwork::setConfig(const PatchworkConfig& config)
{
    config_ = config;
    if (handle_) {
        const rpw_config c = to_c(config_);
        if (rpw_set_config(handle_, &c) != RPW_OK)
            throw std::runtime_error(std::string("RecursivePatchwork::setConfig: ") + rpw_last_error(handle_));
    }
}

std::vector<Point3D> RecursivePatchwork::cleanPoints(const std::vector<Point3D>& points)
{
    std::vector<Point3D> out;
    out.reserve(points.size());
    for (const Point3D& p : points)
        if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) out.push_back(p);
    if (out.size() < points.size())
        std::cout << "Removed " << (points.size() - out.size()) << " points with NaN/inf values" << std::endl;
    return out;
}

std::vector<Point3D> RecursivePatchwork::rotatePoints2D(const std::vector<Point3D>& points, float angle_degrees)
{
    // recursive_patchwork.cpp:37-54: float radians from a double product, cosf / sinf, two products and a difference / sum
    float angle_rad = angle_degrees * M_PI / 180.0f;
    const float cos_a = std::cos(angle_rad), sin_a = std::sin(angle_rad);
    std::vector<Point3D> out; out.reserve(points.size());
    for (const Point3D& p : points) out.emplace_back(p.x * cos_a - p.y * sin_a, p.x * sin_a + p.y * cos_a, p.z);
    return out;
}

std::vector<Point3D> RecursivePatchwork::sampleGroundAndObstacles(const std::vector<Point3D>& points,
                                                                  float target_height, float base_tol)
{
    auto parts = filterGroundPoints(points);
    std::vector<Point3D>& ground = parts.first;
    std::vector<Point3D>& non_ground = parts.second;
    if (non_ground.empty()) return ground;
    std::vector<Point3D> obstacles; obstacles.reserve(non_ground.size());
    for (const Point3D& p : non_ground) {
        if (!(std::sqrt(p.x * p.x + p.y * p.y) > 2.5f)) continue;                 // removeEgoVehicle(non_ground, 2.5f)
        if (std::abs(p.z - target_height) <= base_tol) obstacles.push_back(p);
    }
    std::vector<Point3D> sample;
    if (!ground.empty()) {
        const size_t want = std::min<size_t>(2000, ground.size());
        if (ground.size() <= want) sample = ground;
        else {                                                                   // draw without replacement, rejection as in randomSubsample
            std::mt19937 gen(have_sample_seed_ ? sample_seed_ : std::random_device{}());
            std::uniform_int_distribution<size_t> pick(0, ground.size() - 1);
            std::vector<bool> taken(ground.size(), false);
            sample.reserve(want);
            while (sample.size() < want) {
                const size_t i = pick(gen);
                if (!taken[i]) { taken[i] = true; sample.push_back(ground[i]); }
            }
        }
    }
    std::vector<Point3D> out; out.reserve(sample.size() + obstacles.size());
    out.insert(out.end(), sample.begin(), sample.end());
    out.insert(out.end(), obstacles.begin(), obstacles.end());
    return out;
}

std::vector<unsigned char> RecursivePatchwork::segment(const std::vector<Point3D>& points, std::vector<float>* residuals)
{
    ensure(points.size());
    std::vector<unsigned char> labels(points.size());
    if (residuals) residuals->resize(points.size());
    rpw_frame_stats st;
    const rpw_status rc = rpw_segment(handle_, points.empty() ? nullptr : &points[0].x, (uint32_t)points.size(),
                                      labels.data(), residuals ? residuals->data() : nullptr, &st);
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_segment failed: ") + rpw_last_error(handle_));
    n_zone_last_ = st.n_zone;
    return labels;
}

std::pair<std::vector<Point3D>, std::vector<Point3D>>
RecursivePatchwork::filterGroundPointCloud2(const unsigned char* data, size_t n, unsigned point_step,
                                            unsigned off_x, unsigned off_y, unsigned off_z)
{
    ensure(n);
    std::vector<unsigned char> labels(n);
    rpw_frame_stats st;
    const rpw_status rc = rpw_segment_pointcloud2(handle_, data, (uint32_t)n, point_step, off_x, off_y, off_z,
                                                  labels.data(), nullptr, &st);
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_segment_pointcloud2 failed: ") + rpw_last_error(handle_));
    n_zone_last_ = st.n_zone;
    auto point = [&](size_t i) {
        Point3D p; const unsigned char* r = data + i * point_step;
        std::memcpy(&p.x, r + off_x, 4); std::memcpy(&p.y, r + off_y, 4); std::memcpy(&p.z, r + off_z, 4);
        return p;
    };
    std::vector<Point3D> ground, non_ground;
    if (st.n_zone < 3) {
        for (size_t i = 0; i < n; ++i) if (labels[i] != RPW_LABEL_DROPPED) non_ground.push_back(point(i));
        return {ground, non_ground};
    }
    for (size_t i = 0; i < n; ++i) {
        if (labels[i] == RPW_LABEL_GROUND) ground.push_back(point(i));
        else if (labels[i] == RPW_LABEL_NONGROUND) non_ground.push_back(point(i));
    }
    for (size_t i = 0; i < n; ++i) if (labels[i] == RPW_LABEL_OUTSIDE) non_ground.push_back(point(i));
    return {ground, non_ground};
}

std::pair<std::vector<Point3D>, std::vector<Point3D>>
RecursivePatchwork::fuseAndFilterGroundPoints(const std::vector<std::vector<Point3D>>& clouds,
                                              const std::vector<LidarConfig>& configs)
{
    const size_t ns = std::min(clouds.size(), configs.size());          // as fuseLidarPointClouds does (lidar_fusion.cpp:59)
    if (ns == 0) return {{}, {}};
    std::vector<rpw_sensor> sensors(ns);
    std::vector<uint32_t> off(ns + 1, 0);
    for (size_t k = 0; k < ns; ++k) {
        sensors[k].rotation_deg = configs[k].rotation_angle; sensors[k].ego_radius = configs[k].ego_radius;
        off[k + 1] = off[k] + (uint32_t)clouds[k].size();
    }
    std::vector<Point3D> raw; raw.reserve(off[ns]);
    for (size_t k = 0; k < ns; ++k) raw.insert(raw.end(), clouds[k].begin(), clouds[k].end());
    ensure(raw.size());
    if (rpw_set_sensors(handle_, (uint32_t)ns, sensors.data()) != RPW_OK)
        throw std::runtime_error(std::string("RecursivePatchwork: rpw_set_sensors failed: ") + rpw_last_error(handle_));
    std::vector<unsigned char> labels(raw.size());
    rpw_frame_stats st;
    const rpw_status rc = rpw_segment_fused(handle_, raw.empty() ? nullptr : &raw[0].x, off.data(), labels.data(), nullptr, &st);
    (void)rpw_set_sensors(handle_, 0, nullptr);
    if (rc != RPW_OK) throw std::runtime_error(std::string("RecursivePatchwork: rpw_segment_fused failed: ") + rpw_last_error(handle_));
    // the output points are the FUSED (rotated) ones: same expressions as lidar_fusion.cpp:109-126
    for (size_t k = 0; k < ns; ++k) {
        const float angle_degrees = configs[k].rotation_angle;
        if (!(std::abs(angle_degrees) > 1e-6f)) continue;
        float angle_rad = angle_degrees * M_PI / 180.0f;
        const float cos_a = std::cos(angle_rad), sin_a = std::sin(angle_rad);
        for (uint32_t i = off[k]; i < off[k + 1]; ++i) {
            const float x = raw[i].x * cos_a - raw[i].y * sin_a;
            const float y = raw[i].x * sin_a + raw[i].y * cos_a;
            raw[i].x = x; raw[i].y = y;
        }
    }
    std::vector<Point3D> ground, non_ground;
    if (st.n_zone < 3) {
        for (size_t i = 0; i < raw.size(); ++i) if (labels[i] != RPW_LABEL_DROPPED && labels[i] != RPW_LABEL_EGO) non_ground.push_back(raw[i]);
        return {ground, non_ground};
    }
    for (size_t i = 0; i < raw.size(); ++i) {
        if (labels[i] == RPW_LABEL_GROUND) ground.push_back(raw[i]);
        else if (labels[i] == RPW_LABEL_NONGROUND) non_ground.push_back(raw[i]);
    }
    for (size_t i = 0; i < raw.size(); ++i) if (labels[i] == RPW_LABEL_OUTSIDE) non_ground.push_back(raw[i]);
    return {ground, non_ground};
}

std::pair<std::vector<Point3D>, std::vector<Point3D>>
RecursivePatchwork::filterGroundPoints(const std::vector<Point3D>& points)
{
    const std::vector<unsigned char> labels = segment(points, nullptr);
    std::vector<Point3D> ground, non_ground;
    if (n_zone_last_ < 3) {                              // {{}, cleaned_points}
        non_ground.reserve(points.size());
        for (size_t i = 0; i < points.size(); ++i) if (labels[i] != RPW_LABEL_DROPPED) non_ground.push_back(points[i]);
        return {ground, non_ground};
    }
    ground.reserve(points.size());
    non_ground.reserve(points.size());
    for (size_t i = 0; i < points.size(); ++i) {
        if (labels[i] == RPW_LABEL_GROUND) ground.push_back(points[i]);
        else if (labels[i] == RPW_LABEL_NONGROUND) non_ground.push_back(points[i]);
    }
    for (size_t i = 0; i < points.size(); ++i) if (labels[i] == RPW_LABEL_OUTSIDE) non_ground.push_back(points[i]);
    return {ground, non_ground};
}

} // namespace recursive_patchwork
