// Drop-in for lidar_odometry/src/imageProjection.cpp: same node name, topics, QoS and CloudInfo surface
// (imageProjection.cpp:108-147, 222-237, 649-654).  The queues (IMU, VINS odometry, the 2-scan cloud cache) and the
// integration of the IMU rotation table stay in the node, as in the reference; projectPointCloud (+ per-point deskew) and
// cloudExtraction run on the MI355X through lvi_organize_scan / lvi_organize_scan_deskew.
// Builds only where rclcpp, tf2, pcl_conversions, livox_ros_driver2 and the lidar_odometry messages exist.
#include <deque>
#include <mutex>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <rclcpp/rclcpp.hpp>
#include <nav_msgs/msg/odometry.hpp>
#include <sensor_msgs/msg/imu.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <tf2/LinearMath/Matrix3x3.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.hpp>

#include "../lvi_host.hpp"
#include "lidar_odometry/msg/cloud_info.hpp"
#include "livox_ros_driver2/msg/custom_msg.hpp"
#include "utility.h"   // the reference's ParamServer, publishCloud, qos*, stamp2Sec, imuConverter, imuRPY2rosRPY, imuAngular2rosAngular

class ImageProjection : public ParamServer {
    std::mutex imuLock, odoLock;
    rclcpp::Subscription<livox_ros_driver2::msg::CustomMsg>::SharedPtr subLaserCloud;
    rclcpp::Subscription<sensor_msgs::msg::Imu>::SharedPtr subImu;
    rclcpp::Subscription<nav_msgs::msg::Odometry>::SharedPtr subOdom;
    rclcpp::CallbackGroup::SharedPtr callbackGroupLidar, callbackGroupImu, callbackGroupOdom;
    rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr pubExtractedCloud;
    rclcpp::Publisher<lidar_odometry::msg::CloudInfo>::SharedPtr pubLaserCloudInfo;
    std::deque<sensor_msgs::msg::Imu> imuQueue;
    std::deque<nav_msgs::msg::Odometry> odomQueue;
    std::deque<livox_ros_driver2::msg::CustomMsg> cloudQueue;
    livox_ros_driver2::msg::CustomMsg currentCloudMsg;
    lidar_odometry::msg::CloudInfo cloudInfo;
    std_msgs::msg::Header cloudHeader;
    double timeScanCur = 0, timeScanEnd = 0;
    std::unique_ptr<lvi_host::LidarHandle> handle;
    std::unique_ptr<lvi_host::ImageProjection> ip;

public:
    explicit ImageProjection(const rclcpp::NodeOptions& options) : ParamServer("imageProjection", options)
    {
        lvi_lidar_params p; lvi_lidar_params_default(&p);
        p.N_SCAN = N_SCAN; p.Horizon_SCAN = Horizon_SCAN; p.downsampleRate = downsampleRate;
        p.lidarMinRange = lidarMinRange; p.lidarMaxRange = lidarMaxRange; p.max_raw_points = std::max(N_SCAN * Horizon_SCAN, 1 << 17);
        handle = std::make_unique<lvi_host::LidarHandle>(p, 0);
        ip = std::make_unique<lvi_host::ImageProjection>(*handle);
        callbackGroupLidar = create_callback_group(rclcpp::CallbackGroupType::MutuallyExclusive);     // :111-116
        callbackGroupImu = create_callback_group(rclcpp::CallbackGroupType::MutuallyExclusive);
        callbackGroupOdom = create_callback_group(rclcpp::CallbackGroupType::MutuallyExclusive);
        auto lidarOpt = rclcpp::SubscriptionOptions(); lidarOpt.callback_group = callbackGroupLidar;
        auto imuOpt = rclcpp::SubscriptionOptions(); imuOpt.callback_group = callbackGroupImu;
        auto odomOpt = rclcpp::SubscriptionOptions(); odomOpt.callback_group = callbackGroupOdom;
        subImu = create_subscription<sensor_msgs::msg::Imu>(imuTopic, qos_imu, std::bind(&ImageProjection::imuHandler, this, std::placeholders::_1), imuOpt);
        subOdom = create_subscription<nav_msgs::msg::Odometry>("/vins/odometry/imu_propagate_ros", qos_imu,
                                                               std::bind(&ImageProjection::odometryHandler, this, std::placeholders::_1), odomOpt);
        subLaserCloud = create_subscription<livox_ros_driver2::msg::CustomMsg>(pointCloudTopic, qos_lidar,
                                                                               std::bind(&ImageProjection::cloudHandler, this, std::placeholders::_1), lidarOpt);
        pubExtractedCloud = create_publisher<sensor_msgs::msg::PointCloud2>("lio_sam/deskew/cloud_deskewed", 1);
        pubLaserCloudInfo = create_publisher<lidar_odometry::msg::CloudInfo>("lio_sam/deskew/cloud_info", qos);
    }

    void imuHandler(const sensor_msgs::msg::Imu::SharedPtr imuMsg)                      // :199-203
    {
        sensor_msgs::msg::Imu thisImu = imuConverter(*imuMsg);
        std::lock_guard<std::mutex> lock1(imuLock);
        imuQueue.push_back(thisImu);
    }
    void odometryHandler(const nav_msgs::msg::Odometry::SharedPtr odometryMsg)          // :216-220
    {
        std::lock_guard<std::mutex> lock2(odoLock);
        odomQueue.push_back(*odometryMsg);
    }

    void cloudHandler(const livox_ros_driver2::msg::CustomMsg::SharedPtr laserCloudMsg) // :222-237
    {
        cloudQueue.push_back(*laserCloudMsg);                                           // cachePointCloud :262-281
        if (cloudQueue.size() <= 2) return;
        currentCloudMsg = std::move(cloudQueue.front());
        cloudQueue.pop_front();
        if (currentCloudMsg.point_num < 2) return;
        cloudHeader = currentCloudMsg.header;
        timeScanCur = stamp2Sec(cloudHeader.stamp);
        timeScanEnd = timeScanCur + (double)(float)(currentCloudMsg.points[currentCloudMsg.point_num - 2].offset_time * 1e-9);   // back() of the cloud without the dropped last point
        if (!deskewInfo()) return;
        // livox CustomPoint → the C-ABI's plain struct (moveFromCustomMsg's fields, :249-258)
        std::vector<lvi_livox_pt> pts(currentCloudMsg.point_num);
        for (uint32_t i = 0; i < currentCloudMsg.point_num; i++) {
            const auto& q = currentCloudMsg.points[i];
            pts[i] = lvi_livox_pt{q.x, q.y, q.z, q.reflectivity, q.tag, q.line, 0, q.offset_time};
        }
        lvi_host::CloudInfo ci = ip->cloudHandler(pts.data(), (int32_t)pts.size(), timeScanCur);    // projectPointCloud + cloudExtraction on the GPU
        cloudInfo.start_ring_index.assign(ci.start_ring_index.begin(), ci.start_ring_index.end());
        cloudInfo.end_ring_index.assign(ci.end_ring_index.begin(), ci.end_ring_index.end());
        cloudInfo.point_col_ind.assign(ci.point_col_ind.begin(), ci.point_col_ind.end());
        cloudInfo.point_range.assign(ci.point_range.begin(), ci.point_range.end());
        pcl::PointCloud<pcl::PointXYZI>::Ptr extractedCloud(new pcl::PointCloud<pcl::PointXYZI>());
        extractedCloud->resize(ci.cloud_deskewed.size());
        for (size_t i = 0; i < ci.cloud_deskewed.size(); i++) {
            (*extractedCloud)[i].x = ci.cloud_deskewed[i].x; (*extractedCloud)[i].y = ci.cloud_deskewed[i].y;
            (*extractedCloud)[i].z = ci.cloud_deskewed[i].z; (*extractedCloud)[i].intensity = ci.cloud_deskewed[i].intensity;
        }
        cloudInfo.header = cloudHeader;                                                 // publishClouds :649-654
        cloudInfo.cloud_deskewed = publishCloud(pubExtractedCloud, extractedCloud, cloudHeader.stamp, lidarFrame);
        pubLaserCloudInfo->publish(cloudInfo);
        ip->clearDeskew();                                                              // resetParameters :180-197
    }

    bool deskewInfo()                                                                   // :335-352
    {
        std::lock_guard<std::mutex> lock1(imuLock);
        std::lock_guard<std::mutex> lock2(odoLock);
        if (imuQueue.empty() || stamp2Sec(imuQueue.front().header.stamp) > timeScanCur || stamp2Sec(imuQueue.back().header.stamp) < timeScanEnd) {
            RCLCPP_INFO(get_logger(), "Waiting for IMU data ...");
            return false;
        }
        imuDeskewInfo();
        odomDeskewInfo();
        return true;
    }

    void imuDeskewInfo()                                                                // :354-410 — the table itself is built by lvi_host::ImageProjection
    {
        cloudInfo.imu_available = false;
        while (!imuQueue.empty() && stamp2Sec(imuQueue.front().header.stamp) < timeScanCur - 0.01) imuQueue.pop_front();
        if (imuQueue.empty()) return;
        std::vector<double> t, wx, wy, wz;
        for (auto& m : imuQueue) {
            const double ct = stamp2Sec(m.header.stamp);
            if (ct <= timeScanCur) imuRPY2rosRPY(&m, &cloudInfo.imu_roll_init, &cloudInfo.imu_pitch_init, &cloudInfo.imu_yaw_init);
            if (ct > timeScanEnd + 0.01) break;
            double ax, ay, az;
            imuAngular2rosAngular(&m, &ax, &ay, &az);
            t.push_back(ct); wx.push_back(ax); wy.push_back(ay); wz.push_back(az);
        }
        cloudInfo.imu_available = ip->imuDeskewInfo(t.data(), wx.data(), wy.data(), wz.data(), (int)t.size(), timeScanCur, timeScanEnd);
    }

    void odomDeskewInfo()                                                               // :412-489 (the initial guess for mapOptimization; findPosition is disabled, :522-536)
    {
        cloudInfo.odom_available = false;
        while (!odomQueue.empty() && stamp2Sec(odomQueue.front().header.stamp) < timeScanCur - 0.01) odomQueue.pop_front();
        if (odomQueue.empty() || stamp2Sec(odomQueue.front().header.stamp) > timeScanCur) return;
        nav_msgs::msg::Odometry startOdomMsg;
        for (auto& m : odomQueue) { startOdomMsg = m; if (stamp2Sec(m.header.stamp) >= timeScanCur) break; }
        tf2::Quaternion orientation;
        tf2::fromMsg(startOdomMsg.pose.pose.orientation, orientation);
        double roll, pitch, yaw;
        tf2::Matrix3x3(orientation).getRPY(roll, pitch, yaw);
        cloudInfo.initial_guess_x = startOdomMsg.pose.pose.position.x; cloudInfo.initial_guess_y = startOdomMsg.pose.pose.position.y;
        cloudInfo.initial_guess_z = startOdomMsg.pose.pose.position.z;
        cloudInfo.initial_guess_roll = roll; cloudInfo.initial_guess_pitch = pitch; cloudInfo.initial_guess_yaw = yaw;
        cloudInfo.odom_reset_id = (int)round(startOdomMsg.pose.covariance[0]);
        cloudInfo.odom_available = true;
    }
};

int main(int argc, char** argv)                                                         // :656-670
{
    rclcpp::init(argc, argv);
    rclcpp::NodeOptions options; options.use_intra_process_comms(true);
    rclcpp::executors::MultiThreadedExecutor exec;
    auto IP = std::make_shared<ImageProjection>(options);
    exec.add_node(IP);
    RCLCPP_INFO(rclcpp::get_logger("rclcpp"), "\033[1;32m----> Image Projection Started (MI355X).\033[0m");
    exec.spin();
    rclcpp::shutdown();
    return 0;
}
