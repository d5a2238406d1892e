// Drop-in for lidar_odometry/src/featureExtraction.cpp: same node name, topics, QoS and CloudInfo surface
// (featureExtraction.cpp:43-52, 247-264); calculateSmoothness / markOccludedPoints / extractFeatures run on
// the MI355X through lvi_extract_features.  Builds only where rclcpp and the lidar_odometry messages exist.
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>

#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>

#include "../lvi_host.hpp"
#include "lidar_odometry/msg/cloud_info.hpp"
#include "utility.h"   // the reference's ParamServer, publishCloud, qos (lidar_odometry/src/utility.h)

class FeatureExtraction : public ParamServer {
public:
    rclcpp::Subscription<lidar_odometry::msg::CloudInfo>::SharedPtr subLaserCloudInfo;
    rclcpp::Publisher<lidar_odometry::msg::CloudInfo>::SharedPtr pubLaserCloudInfo;
    rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr pubCornerPoints, pubSurfacePoints;
    std::unique_ptr<lvi_host::LidarHandle> handle;
    std::unique_ptr<lvi_host::FeatureExtraction> fe;

    explicit FeatureExtraction(const rclcpp::NodeOptions& options) : ParamServer("featureExtraction", options)
    {
        lvi_lidar_params p; lvi_lidar_params_default(&p);
        p.N_SCAN = N_SCAN; p.Horizon_SCAN = Horizon_SCAN; p.edgeThreshold = edgeThreshold; p.surfThreshold = surfThreshold;
        p.odometrySurfLeafSize = odometrySurfLeafSize; p.max_raw_points = N_SCAN * Horizon_SCAN;
        handle = std::make_unique<lvi_host::LidarHandle>(p, 0);
        fe = std::make_unique<lvi_host::FeatureExtraction>(*handle);
        subLaserCloudInfo = create_subscription<lidar_odometry::msg::CloudInfo>(
            "lio_sam/deskew/cloud_info", qos, std::bind(&FeatureExtraction::laserCloudInfoHandler, this, std::placeholders::_1));
        pubLaserCloudInfo = create_publisher<lidar_odometry::msg::CloudInfo>("lio_sam/feature/cloud_info", qos);
        pubCornerPoints = create_publisher<sensor_msgs::msg::PointCloud2>("lio_sam/feature/cloud_corner", 1);
        pubSurfacePoints = create_publisher<sensor_msgs::msg::PointCloud2>("lio_sam/feature/cloud_surface", 1);
    }

    static void toHost(const sensor_msgs::msg::PointCloud2& msg, std::vector<lvi_pt>& out)
    {
        pcl::PointCloud<pcl::PointXYZI> c; pcl::fromROSMsg(msg, c);
        out.resize(c.size());
        for (size_t i = 0; i < c.size(); i++) out[i] = lvi_pt{c[i].x, c[i].y, c[i].z, c[i].intensity};
    }
    static pcl::PointCloud<pcl::PointXYZI>::Ptr toPcl(const std::vector<lvi_pt>& in)
    {
        pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>());
        c->resize(in.size());
        for (size_t i = 0; i < in.size(); i++) { (*c)[i].x = in[i].x; (*c)[i].y = in[i].y; (*c)[i].z = in[i].z; (*c)[i].intensity = in[i].intensity; }
        return c;
    }

    void laserCloudInfoHandler(const lidar_odometry::msg::CloudInfo::SharedPtr msgIn)
    {
        lidar_odometry::msg::CloudInfo cloudInfo = *msgIn;
        lvi_host::CloudInfo ci;
        ci.start_ring_index.assign(msgIn->start_ring_index.begin(), msgIn->start_ring_index.end());
        ci.end_ring_index.assign(msgIn->end_ring_index.begin(), msgIn->end_ring_index.end());
        ci.point_col_ind.assign(msgIn->point_col_ind.begin(), msgIn->point_col_ind.end());
        ci.point_range.assign(msgIn->point_range.begin(), msgIn->point_range.end());
        toHost(msgIn->cloud_deskewed, ci.cloud_deskewed);
        fe->laserCloudInfoHandler(ci);                                   // the GPU path
        cloudInfo.start_ring_index.clear(); cloudInfo.end_ring_index.clear();   // freeCloudInfoMemory (:247-253)
        cloudInfo.point_col_ind.clear(); cloudInfo.point_range.clear();
        cloudInfo.cloud_corner = publishCloud(pubCornerPoints, toPcl(ci.cloud_corner), msgIn->header.stamp, lidarFrame);
        cloudInfo.cloud_surface = publishCloud(pubSurfacePoints, toPcl(ci.cloud_surface), msgIn->header.stamp, lidarFrame);
        pubLaserCloudInfo->publish(cloudInfo);
    }
};

int main(int argc, char** argv)
{
    rclcpp::init(argc, argv);
    rclcpp::NodeOptions options; options.use_intra_process_comms(true);
    rclcpp::executors::SingleThreadedExecutor exec;
    auto FE = std::make_shared<FeatureExtraction>(options);
    exec.add_node(FE);
    exec.spin();
    rclcpp::shutdown();
    return 0;
}
