// pointcloud2_io.hpp -- sensor_msgs/PointCloud2 wire format <-> std::vector<Point3D>, ROS-free.
//
// The reference node goes PointCloud2 -> pcl::fromROSMsg -> PointCloud<PointXYZ> -> vector<Point3D>
// (reference src/recursive_patchwork_node.cpp:68-88) and back through pcl::toROSMsg (:110-157). These
// helpers do the same conversions on the raw message bytes, so that neither PCL nor ROS headers are needed
// to test them. Layout written: pcl::PointXYZ's (x, y, z float32 at 0/4/8, point_step 16), which is what
// pcl::toROSMsg produces for that point type.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "recursive_patchwork.hpp"

namespace recursive_patchwork {

struct PointCloud2Layout {
    uint32_t point_step = 16;   // bytes per point
    uint32_t off_x = 0, off_y = 4, off_z = 8;   // FLOAT32 fields
    bool bigendian = false;
};

// x / y / z FLOAT32 fields inside the record; written without additions that could wrap (offset 0xFFFFFFFC must not pass)
inline bool layoutFits(const PointCloud2Layout& L)
{
    if (L.point_step < 12) return false;
    const uint32_t last = L.point_step - 4;
    return L.off_x <= last && L.off_y <= last && L.off_z <= last;
}

// What a subscriber must check before it touches msg.data (sensor_msgs/PointCloud2): every row holds `width` records
// (point_step * width <= row_step) and the buffer holds `height` rows (row_step * height <= data_size). A message
// that fails this is malformed; reading width * height records from it would run past the buffer.
inline bool messageFits(uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step, size_t data_size)
{
    if (point_step == 0) return false;
    if ((uint64_t)point_step * width > row_step) return false;
    return (uint64_t)row_step * height <= (uint64_t)data_size;
}

// Row padding (row_step > width * point_step, possible when height > 1): the records of all rows, packed.
inline std::vector<uint8_t> packRows(const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step)
{
    std::vector<uint8_t> out((size_t)width * height * point_step);
    const size_t row_bytes = (size_t)width * point_step;
    for (uint32_t r = 0; r < height; ++r) std::memcpy(out.data() + r * row_bytes, data + (size_t)r * row_step, row_bytes);
    return out;
}

// bytes -> points. `n` = width * height. Every point is kept (non-finite ones too: cleanPoints drops them on the path).
inline std::vector<Point3D> pointsFromPointCloud2(const uint8_t* data, size_t n, const PointCloud2Layout& L)
{
    if (L.bigendian) throw std::runtime_error("big-endian PointCloud2 is not supported");
    if (!layoutFits(L)) throw std::runtime_error("PointCloud2: x/y/z FLOAT32 fields do not fit point_step");
    std::vector<Point3D> pts(n);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = data + i * L.point_step;
        std::memcpy(&pts[i].x, p + L.off_x, 4);
        std::memcpy(&pts[i].y, p + L.off_y, 4);
        std::memcpy(&pts[i].z, p + L.off_z, 4);
    }
    return pts;
}

// points -> bytes in the pcl::PointXYZ layout (padding word = 1.0f, as PCL leaves it)
inline std::vector<uint8_t> pointCloud2FromPoints(const std::vector<Point3D>& pts, PointCloud2Layout* layout_out = nullptr)
{
    const PointCloud2Layout L;
    std::vector<uint8_t> out(pts.size() * L.point_step);
    const float one = 1.0f;
    for (size_t i = 0; i < pts.size(); ++i) {
        uint8_t* p = out.data() + i * L.point_step;
        std::memcpy(p + 0, &pts[i].x, 4); std::memcpy(p + 4, &pts[i].y, 4); std::memcpy(p + 8, &pts[i].z, 4);
        std::memcpy(p + 12, &one, 4);
    }
    if (layout_out) *layout_out = L;
    return out;
}

} // namespace recursive_patchwork
