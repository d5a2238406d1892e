// Drop-in for feature_tracker/src/feature_tracker_node.cpp: same node name, topics, queue depths and the
// /vins/feature/feature message (feature_tracker_node.cpp:37-231, 410-416).  The callback logic — first image,
// discontinuity restart, frequency control, readImage (CLAHE + pyramidal LK + Shi-Tomasi on the MI355X), updateID, message
// assembly, first-publish suppression — is lvi_host::FeatureTrackerNode; this file converts messages.  rejectWithF's RANSAC
// stays OpenCV on the host (cv::findFundamentalMat), installed as the tracker's hook.  The lidar depth association
// (DepthRegister) is outside the hot path: the depth channel carries the reference's "no depth" value -1 unless the
// node installs get_depth.  Builds only where rclcpp, image_transport, cv_bridge and OpenCV exist.
#include <cv_bridge/cv_bridge.h>
#include <image_transport/image_transport.hpp>
#include <opencv2/calib3d.hpp>
#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/channel_float32.hpp>
#include <sensor_msgs/msg/image.hpp>
#include <sensor_msgs/msg/point_cloud.hpp>
#include <std_msgs/msg/bool.hpp>

#include "../lvi_host.hpp"
#include "camodocal/camera_models/CameraFactory.h"
#include "camodocal/camera_models/CataCamera.h"
#include "parameters.h"   // the reference's readParameters(): ROW, COL, MAX_CNT, MIN_DIST, FREQ, F_THRESHOLD, EQUALIZE, CAM_NAMES (feature_tracker/src/parameters.h)

static std::unique_ptr<lvi_host::TrackerHandle> handle;
static std::unique_ptr<lvi_host::FeatureTracker> tracker;
static std::unique_ptr<lvi_host::FeatureTrackerNode> node_logic;
static rclcpp::Publisher<sensor_msgs::msg::PointCloud>::SharedPtr pub_feature;
static rclcpp::Publisher<std_msgs::msg::Bool>::SharedPtr pub_restart;

void img_callback(const sensor_msgs::msg::Image::ConstSharedPtr img_msg)
{
    const double cur_img_time = img_msg->header.stamp.sec + img_msg->header.stamp.nanosec * (1e-9);
    cv_bridge::CvImageConstPtr ptr;                                                     // :114-129
    if (img_msg->encoding == "8UC1") {
        sensor_msgs::msg::Image img = *img_msg; img.encoding = "mono8";
        ptr = cv_bridge::toCvCopy(img, sensor_msgs::image_encodings::MONO8);
    } else {
        ptr = cv_bridge::toCvCopy(img_msg, sensor_msgs::image_encodings::MONO8);
    }
    cv::Mat img = ptr->image.rowRange(0, ROW);
    if (!img.isContinuous()) img = img.clone();
    lvi_host::FeatureMsg m;
    const auto outcome = node_logic->img_callback(img.data, cur_img_time, &m);
    if (outcome == lvi_host::FeatureTrackerNode::RESTART) {                             // :50-59
        std_msgs::msg::Bool restart_flag; restart_flag.data = true;
        pub_restart->publish(restart_flag);
        return;
    }
    if (outcome != lvi_host::FeatureTrackerNode::PUBLISHED) return;                     // first image, not a PUB frame, or the suppressed first message (:225-231)
    sensor_msgs::msg::PointCloud feature_points;
    feature_points.header = img_msg->header;
    feature_points.header.frame_id = m.frame_id;                                        // "vins_body"
    feature_points.points.resize(m.points.size());
    for (size_t i = 0; i < m.points.size(); i++) { feature_points.points[i].x = m.points[i].x; feature_points.points[i].y = m.points[i].y; feature_points.points[i].z = m.points[i].z; }
    static const char* names[6] = {"", "", "", "", "", "depth"};
    for (int c = 0; c < 6; c++) {                                                       // id, u, v, velocity_x, velocity_y, depth (:204-224)
        sensor_msgs::msg::ChannelFloat32 ch; ch.name = names[c]; ch.values = m.channels[c];
        feature_points.channels.push_back(ch);
    }
    pub_feature->publish(feature_points);
}

int main(int argc, char** argv)
{
    rclcpp::init(argc, argv);
    auto n = rclcpp::Node::make_shared("feature_tracker");
    readParameters(n);                                                                  // the reference's yaml loader (parameters.cpp:53-110)
    lvi_tracker_params p; lvi_tracker_params_default(&p);
    p.max_width = COL; p.max_height = ROW; p.max_cnt = MAX_CNT; p.min_dist = MIN_DIST;
    handle = std::make_unique<lvi_host::TrackerHandle>(p, 0);
    tracker = std::make_unique<lvi_host::FeatureTracker>(*handle, ROW, COL, MAX_CNT, MIN_DIST);
    if (EQUALIZE) tracker->setEqualize(true);
    tracker->F_THRESHOLD = F_THRESHOLD;
    {   // FeatureTracker::readIntrinsicParameter (feature_tracker.cpp:256-260): the MEI parameters of the camera yaml
        auto cam = camodocal::CameraFactory::instance()->generateCameraFromYamlFile(CAM_NAMES[0]);
        const auto& q = std::dynamic_pointer_cast<camodocal::CataCamera>(cam)->getParameters();
        tracker->setCamera(lvi_mei_params{q.xi(), q.k1(), q.k2(), q.p1(), q.p2(), q.gamma1(), q.gamma2(), q.u0(), q.v0()});
    }
    tracker->findFundamentalMat = [](const std::vector<lvi_host::Point2f>& a, const std::vector<lvi_host::Point2f>& b, double thr, std::vector<uint8_t>& status) {
        std::vector<cv::Point2f> ca(a.size()), cb(b.size());
        for (size_t i = 0; i < a.size(); i++) { ca[i] = cv::Point2f(a[i].x, a[i].y); cb[i] = cv::Point2f(b[i].x, b[i].y); }
        std::vector<uchar> st;
        cv::findFundamentalMat(ca, cb, cv::FM_RANSAC, thr, 0.99, st);                   // feature_tracker.cpp:229
        status.assign(st.begin(), st.end());
    };
    node_logic = std::make_unique<lvi_host::FeatureTrackerNode>(*tracker, FREQ);
    image_transport::ImageTransport it(n);                                              // :410-416
    image_transport::Subscriber sub = it.subscribe("/camera/image_raw", 10, img_callback);
    pub_feature = n->create_publisher<sensor_msgs::msg::PointCloud>("/vins/feature/feature", 1000);
    pub_restart = n->create_publisher<std_msgs::msg::Bool>("/vins/feature/restart", 1000);
    RCLCPP_INFO(rclcpp::get_logger("rclcpp"), "\033[1;32m----> VINS Feature Extraction Started (MI355X).\033[0m");
    rclcpp::executors::MultiThreadedExecutor executor(rclcpp::ExecutorOptions(), 2);
    executor.add_node(n);
    executor.spin();
    return 0;
}
