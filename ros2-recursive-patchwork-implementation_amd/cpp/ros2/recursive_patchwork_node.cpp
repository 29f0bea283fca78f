// recursive_patchwork_node.cpp -- ROS2 surface of the hot path, same node name, parameters, topics and QoS
// as the reference node (reference src/recursive_patchwork_node.cpp:13-58,61-108,110-221; launch file
// launch/recursive_patchwork.launch.py:11-67), backed by the MI355X library. PCL-free: the PointCloud2
// bytes are converted by ../pointcloud2_io.hpp.
//
// NOT BUILT in this repository's image (no rclcpp). Build inside a ROS2 workspace with the CMakeLists.txt / package.xml
// beside this file (colcon build --packages-select recursive_patchwork); launch/recursive_patchwork.launch.py mirrors the
// reference's launch arguments.
#include <algorithm>
#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <sensor_msgs/msg/point_field.hpp>
#include <visualization_msgs/msg/marker_array.hpp>

#include "../frame_pipeline.hpp"
#include "../pointcloud2_io.hpp"
#include "../recursive_patchwork.hpp"

namespace recursive_patchwork {

class RecursivePatchworkNode : public rclcpp::Node {
public:
    RecursivePatchworkNode() : Node("recursive_patchwork_node")
    {
        input_topic_ = declare_parameter<std::string>("input_topic", "/lidar/points");
        ground_topic_ = declare_parameter<std::string>("ground_topic", "/patchwork/ground");
        obstacles_topic_ = declare_parameter<std::string>("obstacles_topic", "/patchwork/obstacles");
        visualization_topic_ = declare_parameter<std::string>("visualization_topic", "/patchwork/visualization");
        (void)declare_parameter<int>("min_points", 100);                      // declared and unused, as in the reference
        const int max_iterations = declare_parameter<int>("max_iterations", 50);
        const double distance_threshold = declare_parameter<double>("distance_threshold", 0.1);
        const double angle_threshold = declare_parameter<double>("angle_threshold", 0.1);
        const int device = declare_parameter<int>("device", 0);               // addition: which MI355X
        // addition: frames in flight. 1 = the reference's behaviour (a callback returns with its own frame published).
        // k > 1: a callback submits its frame to a free pipeline slot and publishes the OLDEST frame in flight once k are
        // (in order, k - 1 callbacks late): the GPU then works on k frames at a time instead of idling between callbacks.
        depth_ = static_cast<unsigned>(std::min<int64_t>(32, std::max<int64_t>(1, declare_parameter<int>("frames_in_flight", 1))));

        PatchworkConfig config;                                               // reference mapping, node.cpp:36-40
        config.max_iter = max_iterations;
        config.th_dist = static_cast<float>(distance_threshold);
        config.th_seeds = static_cast<float>(angle_threshold);
        RecursivePatchwork::setDevice(device);
        processor_ = std::make_unique<RecursivePatchwork>(config);            // throws if no MI355X: no CPU fallback
        // addition: expected points per scan. Every pipeline slot is sized for it up front; a larger scan still works (the
        // frames in flight are published first, then the engine grows), it only costs that one stall.
        const int64_t max_points = declare_parameter<int>("max_points", 262144);
        if (depth_ > 1) {
            if (max_points > 0) processor_->reserve(static_cast<size_t>(max_points));
            pipeline_ = std::make_unique<Pipeline>(*processor_, depth_);
        }

        sub_ = create_subscription<sensor_msgs::msg::PointCloud2>(
            input_topic_, rclcpp::SensorDataQoS(),
            [this](sensor_msgs::msg::PointCloud2::SharedPtr msg) { onCloud(*msg); });
        ground_pub_ = create_publisher<sensor_msgs::msg::PointCloud2>(ground_topic_, 10);
        obstacles_pub_ = create_publisher<sensor_msgs::msg::PointCloud2>(obstacles_topic_, 10);
        markers_pub_ = create_publisher<visualization_msgs::msg::MarkerArray>(visualization_topic_, 10);
        RCLCPP_INFO(get_logger(), "recursive_patchwork_node (MI355X) listening on %s", input_topic_.c_str());
    }

private:
    static bool findLayout(const sensor_msgs::msg::PointCloud2& m, PointCloud2Layout& L)
    {
        bool fx = false, fy = false, fz = false;
        for (const auto& f : m.fields) {
            if (f.datatype != sensor_msgs::msg::PointField::FLOAT32) continue;
            if (f.name == "x") { L.off_x = f.offset; fx = true; }
            else if (f.name == "y") { L.off_y = f.offset; fy = true; }
            else if (f.name == "z") { L.off_z = f.offset; fz = true; }
        }
        L.point_step = m.point_step; L.bigendian = m.is_bigendian;
        return fx && fy && fz;
    }

    sensor_msgs::msg::PointCloud2 toMsg(const std::vector<Point3D>& pts, const std_msgs::msg::Header& header) const
    {
        sensor_msgs::msg::PointCloud2 out;
        PointCloud2Layout L;
        out.data = pointCloud2FromPoints(pts, &L);
        out.header = header; out.height = 1; out.width = static_cast<uint32_t>(pts.size());
        out.is_bigendian = false; out.is_dense = true; out.point_step = L.point_step; out.row_step = L.point_step * out.width;
        const char* names[3] = {"x", "y", "z"};
        const uint32_t offs[3] = {L.off_x, L.off_y, L.off_z};
        for (int i = 0; i < 3; ++i) {
            sensor_msgs::msg::PointField f;
            f.name = names[i]; f.offset = offs[i]; f.datatype = sensor_msgs::msg::PointField::FLOAT32; f.count = 1;
            out.fields.push_back(f);
        }
        return out;
    }

    void publishFrame(const std::pair<std::vector<Point3D>, std::vector<Point3D>>& r, const std_msgs::msg::Header& header)
    {
        if (!r.first.empty()) ground_pub_->publish(toMsg(r.first, header));             // empty clouds are not
        if (!r.second.empty()) obstacles_pub_->publish(toMsg(r.second, header));        // published (node.cpp:116,136)
        publishMarkers(r.first.size(), r.second.size(), header);
    }

    void onCloud(const sensor_msgs::msg::PointCloud2& msg)
    {
        try {
            const auto t0 = std::chrono::high_resolution_clock::now();
            PointCloud2Layout L;
            if (!findLayout(msg, L)) { RCLCPP_ERROR(get_logger(), "PointCloud2 without FLOAT32 x/y/z fields"); return; }
            if (L.bigendian || (L.point_step & 3u) || (L.off_x & 3u) || (L.off_y & 3u) || (L.off_z & 3u) || !layoutFits(L)) {
                RCLCPP_ERROR(get_logger(), "unsupported PointCloud2 layout"); return;
            }
            // a malformed message must not be read past its buffer (the reference's pcl::fromROSMsg checks the same)
            if (!messageFits(msg.width, msg.height, msg.point_step, msg.row_step, msg.data.size())) {
                RCLCPP_ERROR(get_logger(), "malformed PointCloud2: %u x %u records of %u bytes, row_step %u, but %zu data bytes",
                             msg.width, msg.height, msg.point_step, msg.row_step, msg.data.size());
                return;
            }
            const size_t n = static_cast<size_t>(msg.width) * msg.height;
            if (n == 0) { RCLCPP_WARN(get_logger(), "Received empty point cloud"); return; }   // as the reference (node.cpp:72-75)
            // rows with padding are packed first; the usual message (row_step == width * point_step) goes as it is
            std::vector<uint8_t> packed;
            const uint8_t* data = msg.data.data();
            if (msg.row_step != msg.width * msg.point_step) { packed = packRows(data, msg.width, msg.height, msg.point_step, msg.row_step); data = packed.data(); }
            if (depth_ <= 1) {
                // the message bytes go to the GPU as they are (no PCL hop, no host repacking)
                publishFrame(processor_->filterGroundPointCloud2(data, n, L.point_step, L.off_x, L.off_y, L.off_z), msg.header);
            } else {
                // pipeline (../frame_pipeline.hpp): the slot this frame takes still holds the frame submitted depth_ callbacks
                // ago, which is published first; a frame beyond the engine's capacity drains the pipeline before it grows
                pipeline_->push(data, n, L.point_step, L.off_x, L.off_y, L.off_z, msg.header,
                                [this](const Pipeline::Result& r, const std_msgs::msg::Header& hd) { publishFrame(r, hd); });
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
            RCLCPP_DEBUG(get_logger(), "processed %zu points in %.3f ms", n, ms);
        } catch (const std::exception& e) {
            RCLCPP_ERROR(get_logger(), "point cloud callback failed: %s", e.what());
        }
    }

    void publishMarkers(size_t n_ground, size_t n_obstacles, const std_msgs::msg::Header& header)
    {
        visualization_msgs::msg::MarkerArray arr;
        auto text = [&](const char* ns, int id, double z, float r, float g, const std::string& s) {
            visualization_msgs::msg::Marker m;
            m.header = header; m.ns = ns; m.id = id;
            m.type = visualization_msgs::msg::Marker::TEXT_VIEW_FACING; m.action = visualization_msgs::msg::Marker::ADD;
            m.pose.position.z = z; m.pose.orientation.w = 1.0; m.scale.z = 0.5;
            m.color.r = r; m.color.g = g; m.color.b = 0.0f; m.color.a = 1.0f; m.text = s;
            arr.markers.push_back(m);
        };
        if (n_ground) text("ground_summary", 0, 2.0, 0.0f, 1.0f, "Ground: " + std::to_string(n_ground) + " points");
        if (n_obstacles) text("obstacles_summary", 1, 1.5, 1.0f, 0.0f, "Obstacles: " + std::to_string(n_obstacles) + " points");
        markers_pub_->publish(arr);
    }

    std::string input_topic_, ground_topic_, obstacles_topic_, visualization_topic_;
    std::unique_ptr<RecursivePatchwork> processor_;
    unsigned depth_ = 1;
    using Pipeline = BasicFramePipeline<RecursivePatchwork, std_msgs::msg::Header>;
    std::unique_ptr<Pipeline> pipeline_;
    rclcpp::Subscription<sensor_msgs::msg::PointCloud2>::SharedPtr sub_;
    rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr ground_pub_, obstacles_pub_;
    rclcpp::Publisher<visualization_msgs::msg::MarkerArray>::SharedPtr markers_pub_;
};

} // namespace recursive_patchwork

int main(int argc, char** argv)
{
    rclcpp::init(argc, argv);
    rclcpp::spin(std::make_shared<recursive_patchwork::RecursivePatchworkNode>());
    rclcpp::shutdown();
    return 0;
}
