// Drop-in for the scan-matching half of lidar_odometry/src/mapOptimization.cpp: same node name, topics, QoS, frame ids and
// odometry messages (mapOptimization.cpp:161-177, 238-245, 298-333, 1666-1746).  updateInitialGuess, extractNearby, the
// keyframe decision and the key poses run in lvi_host::MapOptimizationNode (host C++ over the C-ABI); the keyframe clouds,
// the local map (incremental lvi_map_update), downsampleCurrentScan and scan2MapOptimization run on the MI355X.
// Outside the hot path and therefore NOT reproduced here: the GTSAM / iSAM2 factor graph, GPS and loop factors,
// the save_map service and the visualisation thread — the key pose pushed is the scan-matching result ("odometry chain").
// Builds only where rclcpp, tf2_ros, pcl_conversions and the lidar_odometry messages exist.
#include <mutex>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <rclcpp/rclcpp.hpp>
#include <nav_msgs/msg/odometry.hpp>
#include <nav_msgs/msg/path.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <tf2/LinearMath/Quaternion.h>
#include <tf2_geometry_msgs/tf2_geometry_msgs.hpp>
#include <tf2_ros/transform_broadcaster.h>

#include "../lvi_host.hpp"
#include "lidar_odometry/msg/cloud_info.hpp"
#include "utility.h"   // the reference's ParamServer, publishCloud, qos, stamp2Sec

class mapOptimization : public ParamServer {
    rclcpp::Publisher<nav_msgs::msg::Odometry>::SharedPtr pubLaserOdometryGlobal, pubLaserOdometryIncremental;
    rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr pubKeyPoses, pubRecentKeyFrames;
    rclcpp::Publisher<nav_msgs::msg::Path>::SharedPtr pubPath;
    rclcpp::Subscription<lidar_odometry::msg::CloudInfo>::SharedPtr subCloud;
    std::unique_ptr<tf2_ros::TransformBroadcaster> br;
    std::mutex mtx;
    std::unique_ptr<lvi_host::LidarHandle> handle;
    std::unique_ptr<lvi_host::MapOptimizationNode> mo;
    nav_msgs::msg::Path globalPath;
    // incremental odometry (publishOdometry :1693-1741)
    bool lastIncreOdomPubFlag = false;
    nav_msgs::msg::Odometry laserOdomIncremental;
    lvi_host::Affine3f increOdomAffine{}, incrementalOdometryAffineFront{};

public:
    explicit mapOptimization(const rclcpp::NodeOptions& options) : ParamServer("mapOptimization", options)
    {
        lvi_lidar_params p; lvi_lidar_params_default(&p);
        p.N_SCAN = N_SCAN; p.Horizon_SCAN = Horizon_SCAN;
        p.edgeFeatureMinValidNum = edgeFeatureMinValidNum; p.surfFeatureMinValidNum = surfFeatureMinValidNum;
        p.mappingCornerLeafSize = mappingCornerLeafSize; p.mappingSurfLeafSize = mappingSurfLeafSize;
        p.z_tollerance = z_tollerance; p.rotation_tollerance = rotation_tollerance; p.imuRPYWeight = imuRPYWeight;
        p.max_raw_points = N_SCAN * Horizon_SCAN; p.max_map_points = 1 << 23; p.max_keyframes = 8192; p.max_keyframe_points = 1 << 25;
        handle = std::make_unique<lvi_host::LidarHandle>(p, 0);
        lvi_host::MapCallerParams c;
        c.useImuHeadingInitialization = useImuHeadingInitialization; c.mappingProcessInterval = mappingProcessInterval;
        c.surroundingkeyframeAddingDistThreshold = surroundingkeyframeAddingDistThreshold;
        c.surroundingkeyframeAddingAngleThreshold = surroundingkeyframeAddingAngleThreshold;
        c.surroundingKeyframeDensity = surroundingKeyframeDensity; c.surroundingKeyframeSearchRadius = surroundingKeyframeSearchRadius;
        c.sensorIsLivox = sensor == SensorType::LIVOX;
        mo = std::make_unique<lvi_host::MapOptimizationNode>(*handle, c);
        pubKeyPoses = create_publisher<sensor_msgs::msg::PointCloud2>("lio_sam/mapping/trajectory", 1);                 // :161-167
        pubLaserOdometryGlobal = create_publisher<nav_msgs::msg::Odometry>("lio_sam/mapping/odometry", qos);
        pubLaserOdometryIncremental = create_publisher<nav_msgs::msg::Odometry>("lio_sam/mapping/odometry_incremental", qos);
        pubPath = create_publisher<nav_msgs::msg::Path>("lio_sam/mapping/path", 1);
        pubRecentKeyFrames = create_publisher<sensor_msgs::msg::PointCloud2>("lio_sam/mapping/map_local", 1);           // :242
        br = std::make_unique<tf2_ros::TransformBroadcaster>(this);
        subCloud = create_subscription<lidar_odometry::msg::CloudInfo>(                                                  // :169-171
            "lio_sam/feature/cloud_info", qos, std::bind(&mapOptimization::laserCloudInfoHandler, this, std::placeholders::_1));
    }

    static void toHost(const sensor_msgs::msg::PointCloud2& msg, std::vector<lvi_pt>& out)
    {
        pcl::PointCloud<pcl::PointXYZI> c; pcl::fromROSMsg(msg, c);
        out.resize(c.size());
        for (size_t i = 0; i < c.size(); i++) out[i] = lvi_pt{c[i].x, c[i].y, c[i].z, c[i].intensity};
    }

    void laserCloudInfoHandler(const lidar_odometry::msg::CloudInfo::SharedPtr msgIn)     // :298-333
    {
        lvi_host::CloudInfo ci;
        ci.stamp = stamp2Sec(msgIn->header.stamp);
        ci.imu_available = msgIn->imu_available; ci.odom_available = msgIn->odom_available; ci.odom_reset_id = msgIn->odom_reset_id;
        ci.imu_roll_init = msgIn->imu_roll_init; ci.imu_pitch_init = msgIn->imu_pitch_init; ci.imu_yaw_init = msgIn->imu_yaw_init;
        ci.initial_guess_x = msgIn->initial_guess_x; ci.initial_guess_y = msgIn->initial_guess_y; ci.initial_guess_z = msgIn->initial_guess_z;
        ci.initial_guess_roll = msgIn->initial_guess_roll; ci.initial_guess_pitch = msgIn->initial_guess_pitch; ci.initial_guess_yaw = msgIn->initial_guess_yaw;
        toHost(msgIn->cloud_corner, ci.cloud_corner); toHost(msgIn->cloud_surface, ci.cloud_surface);
        std::lock_guard<std::mutex> lock(mtx);
        const float* T = mo->transformTobeMapped;
        incrementalOdometryAffineFront = lvi_host::getTransformation(T[3], T[4], T[5], T[0], T[1], T[2]);          // updateInitialGuess :809
        if (!mo->laserCloudInfoHandler(ci)) return;                                      // mappingProcessInterval gate; else: guess, map, match, keyframe
        publishOdometry(msgIn->header.stamp, ci);
        publishFrames(msgIn->header.stamp);
    }

    void publishOdometry(const builtin_interfaces::msg::Time& stamp, const lvi_host::CloudInfo& ci)               // :1666-1746
    {
        const float* T = mo->transformTobeMapped;
        nav_msgs::msg::Odometry laserOdometryROS;
        laserOdometryROS.header.stamp = stamp; laserOdometryROS.header.frame_id = odometryFrame; laserOdometryROS.child_frame_id = "odom_mapping";
        laserOdometryROS.pose.pose.position.x = T[3]; laserOdometryROS.pose.pose.position.y = T[4]; laserOdometryROS.pose.pose.position.z = T[5];
        tf2::Quaternion quat_tf; quat_tf.setRPY(T[0], T[1], T[2]);
        geometry_msgs::msg::Quaternion quat_msg; tf2::convert(quat_tf, quat_msg);
        laserOdometryROS.pose.pose.orientation = quat_msg;
        pubLaserOdometryGlobal->publish(laserOdometryROS);
        geometry_msgs::msg::TransformStamped tf;                                         // TF odom → lidar_link
        tf.header.stamp = stamp; tf.header.frame_id = odometryFrame; tf.child_frame_id = "lidar_link";
        tf.transform.translation.x = T[3]; tf.transform.translation.y = T[4]; tf.transform.translation.z = T[5]; tf.transform.rotation = quat_msg;
        br->sendTransform(tf);
        if (!lastIncreOdomPubFlag) {
            lastIncreOdomPubFlag = true; laserOdomIncremental = laserOdometryROS;
            increOdomAffine = lvi_host::getTransformation(T[3], T[4], T[5], T[0], T[1], T[2]);
        } else {
            const lvi_host::Affine3f back = lvi_host::getTransformation(T[3], T[4], T[5], T[0], T[1], T[2]);      // incrementalOdometryAffineBack :1342
            increOdomAffine = lvi_host::affineMul(increOdomAffine, lvi_host::affineMul(lvi_host::affineInverse(incrementalOdometryAffineFront), back));
            float x, y, z, roll, pitch, yaw;
            lvi_host::getTranslationAndEulerAngles(increOdomAffine, x, y, z, roll, pitch, yaw);
            if (ci.imu_available && std::abs(ci.imu_pitch_init) < 1.4) {                  // slerp with weight 0.1 (:1710-1726)
                tf2::Quaternion a, b; double r, p, yy;
                a.setRPY(roll, 0, 0); b.setRPY(ci.imu_roll_init, 0, 0); tf2::Matrix3x3(a.slerp(b, 0.1)).getRPY(r, p, yy); roll = r;
                a.setRPY(0, pitch, 0); b.setRPY(0, ci.imu_pitch_init, 0); tf2::Matrix3x3(a.slerp(b, 0.1)).getRPY(r, p, yy); pitch = p;
            }
            laserOdomIncremental.header.stamp = stamp; laserOdomIncremental.header.frame_id = odometryFrame; laserOdomIncremental.child_frame_id = "odom_mapping";
            laserOdomIncremental.pose.pose.position.x = x; laserOdomIncremental.pose.pose.position.y = y; laserOdomIncremental.pose.pose.position.z = z;
            tf2::Quaternion q; q.setRPY(roll, pitch, yaw);
            geometry_msgs::msg::Quaternion qm; tf2::convert(q, qm);
            laserOdomIncremental.pose.pose.orientation = qm;
            laserOdomIncremental.pose.covariance[0] = mo->last.degenerate ? 1 : 0;        // isDegenerate
        }
        pubLaserOdometryIncremental->publish(laserOdomIncremental);
    }

    void publishFrames(const builtin_interfaces::msg::Time& stamp)                        // :1748-1790 (key poses, local map, path)
    {
        if (mo->cloudKeyPoses3D.empty()) return;
        pcl::PointCloud<pcl::PointXYZI>::Ptr kp(new pcl::PointCloud<pcl::PointXYZI>());
        for (const lvi_pt& p : mo->cloudKeyPoses3D) { pcl::PointXYZI q; q.x = p.x; q.y = p.y; q.z = p.z; q.intensity = p.intensity; kp->push_back(q); }
        publishCloud(pubKeyPoses, kp, stamp, odometryFrame);
        if (pubRecentKeyFrames->get_subscription_count() != 0) {
            int32_t counts[8]; lvi_get_counts(handle->get(), counts);
            std::vector<lvi_pt> c((size_t)std::max(counts[5], 1)), s((size_t)std::max(counts[6], 1));
            lvi_cloud cc{(int32_t)c.size(), 0, c.data()}, sc{(int32_t)s.size(), 0, s.data()};
            if (lvi_get_map_ds(handle->get(), &cc, &sc) == LVI_OK) {                      // laserCloudSurfFromMapDS
                pcl::PointCloud<pcl::PointXYZI>::Ptr m(new pcl::PointCloud<pcl::PointXYZI>());
                for (int i = 0; i < sc.n; i++) { pcl::PointXYZI q; q.x = s[i].x; q.y = s[i].y; q.z = s[i].z; q.intensity = s[i].intensity; m->push_back(q); }
                publishCloud(pubRecentKeyFrames, m, stamp, odometryFrame);
            }
        }
        if (mo->lastSavedKeyFrame && pubPath->get_subscription_count() != 0) {            // updatePath :1650-1664
            const lvi_host::PointTypePose& p = mo->cloudKeyPoses6D.back();
            geometry_msgs::msg::PoseStamped ps;
            ps.header.stamp = rclcpp::Time((int64_t)(p.time * 1e9)); ps.header.frame_id = odometryFrame;
            ps.pose.position.x = p.x; ps.pose.position.y = p.y; ps.pose.position.z = p.z;
            tf2::Quaternion q; q.setRPY(p.roll, p.pitch, p.yaw);
            ps.pose.orientation.x = q.x(); ps.pose.orientation.y = q.y(); ps.pose.orientation.z = q.z(); ps.pose.orientation.w = q.w();
            globalPath.poses.push_back(ps);
            globalPath.header.stamp = stamp; globalPath.header.frame_id = odometryFrame;
            pubPath->publish(globalPath);
        }
    }
};

int main(int argc, char** argv)
{
    rclcpp::init(argc, argv);
    rclcpp::NodeOptions options; options.use_intra_process_comms(true);
    rclcpp::executors::SingleThreadedExecutor exec;
    auto MO = std::make_shared<mapOptimization>(options);
    exec.add_node(MO);
    RCLCPP_INFO(rclcpp::get_logger("rclcpp"), "\033[1;32m----> Map Optimization Started (MI355X scan matching).\033[0m");
    exec.spin();
    rclcpp::shutdown();
    return 0;
}
