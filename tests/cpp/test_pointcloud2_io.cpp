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
    // offsets near 2^32 must not wrap through the "fits in the record" check
    for (uint32_t off : {0xFFFFFFFCu, 0xFFFFFFF0u, 29u, 32u}) {
        PointCloud2Layout w; w.point_step = 32; w.off_x = off;
        assert(!layoutFits(w));
        threw = false;
        try { pointsFromPointCloud2(v.data(), 1, w); } catch (const std::runtime_error&) { threw = true; }
        assert(threw);
    }
    { PointCloud2Layout ok; ok.point_step = 32; ok.off_x = 28; assert(layoutFits(ok)); }
    // message-size checks of the subscriber: a truncated buffer, a row_step shorter than a row, overflowing products
    assert(messageFits(2, 1, 32, 64, 64));
    assert(!messageFits(2, 1, 32, 64, 63));            // truncated data
    assert(!messageFits(2, 1, 32, 48, 64));            // row_step < width * point_step
    assert(!messageFits(3, 2, 32, 96, 160));           // second row incomplete
    assert(!messageFits(0x10000000u, 16, 16, 0xFFFFFFFFu, 1024));   // products beyond 32 bits
    assert(!messageFits(1, 1, 0, 0, 0));
    // rows with padding: height 2, width 2, point_step 16, row_step 40
    {
        std::vector<Point3D> g = {{1, 2, 3}, {4, 5, 6}, {7, 8, 9}, {10, 11, 12}};
        std::vector<uint8_t> padded(80, 0xEE);
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) std::memcpy(&padded[r * 40 + c * 16], &g[r * 2 + c].x, 12);
        assert(messageFits(2, 2, 16, 40, padded.size()));
        auto packed = packRows(padded.data(), 2, 2, 16, 40);
        assert(packed.size() == 64);
        PointCloud2Layout P;
        auto q = pointsFromPointCloud2(packed.data(), 4, P);
        for (int i = 0; i < 4; ++i) assert(std::memcmp(&q[i], &g[i], 12) == 0);
    }
    std::printf("pointcloud2_io OK\n");
    return 0;
}
