// CPU-only check of the PointCloud2 byte conversions used by the ROS2 node (no ROS, no GPU).
#include "pointcloud2_io.hpp"
#include <cassert>
#include <cmath>
#include <cstdio>
using namespace recursive_patchwork;
int main()
{
    std::vector<Point3D> pts = {{1.f, 2.f, 3.f}, {-4.5f, 0.f, NAN}, {1e30f, -0.0f, 7.f}};
    PointCloud2Layout L;
    auto bytes = pointCloud2FromPoints(pts, &L);
    assert(L.point_step == 16 && bytes.size() == 48);
    auto back = pointsFromPointCloud2(bytes.data(), pts.size(), L);
    for (size_t i = 0; i < pts.size(); ++i) assert(std::memcmp(&back[i], &pts[i], 12) == 0);
    // a Velodyne-style layout: x,y,z,intensity,ring,time, point_step 32, fields at 0/4/8
    std::vector<uint8_t> v(32 * 2, 0xAB);
    const float a[3] = {9.f, 8.f, 7.f}, b[3] = {-1.f, -2.f, -3.f};
    std::memcpy(&v[0], a, 12); std::memcpy(&v[32], b, 12);
    PointCloud2Layout V; V.point_step = 32;
    auto p2 = pointsFromPointCloud2(v.data(), 2, V);
    assert(p2[0].x == 9.f && p2[1].z == -3.f);
    // odd field order
    PointCloud2Layout O; O.point_step = 20; O.off_z = 0; O.off_x = 8; O.off_y = 16;
    std::vector<uint8_t> o(20, 0); const float zz = 5.f, xx = 6.f, yy = 7.f;
    std::memcpy(&o[0], &zz, 4); std::memcpy(&o[8], &xx, 4); std::memcpy(&o[16], &yy, 4);
    auto p3 = pointsFromPointCloud2(o.data(), 1, O);
    assert(p3[0].x == 6.f && p3[0].y == 7.f && p3[0].z == 5.f);
    bool threw = false;
    try { PointCloud2Layout bad; bad.point_step = 8; pointsFromPointCloud2(v.data(), 1, bad); } catch (const std::runtime_error&) { threw = true; }
    assert(threw);
    std::printf("pointcloud2_io OK\n");
    return 0;
}
