// Pose accumulation and frame conversion as vo_node does them after each
// processFrame call (reference src/vo_node.cpp:703-774): the relative pose
// T_prev_from_curr from PnP (optical frame, translation in millimetres) is
// chained into T_0_from_curr unless the motion is below the zero-motion
// thresholds (vo.zero_motion_threshold_mm = 2.0,
// vo.zero_motion_rotation_threshold_rad = 0.002, reference
// src/vo_node.cpp:280-281), and converted to the body frame (x forward,
// y left, z up; metres, roll/pitch/yaw).  Host-only, no ROS, no OpenCV needed.
#pragma once

#include <algorithm>
#include <cmath>

namespace vo {

// relative pose in the body frame (reference include/visual_odometry/imu_fusion.hpp:36-40)
struct RelPose {
    double x{0.0}, y{0.0}, z{0.0};
    double roll{0.0}, pitch{0.0}, yaw{0.0};
    bool valid{false};
};

// accumulated pose in the body frame (the pose_* fields of the node's metrics)
struct BodyPose {
    double x{0.0}, y{0.0}, z{0.0};
    double roll{0.0}, pitch{0.0}, yaw{0.0};
};

class PoseAccumulator {
public:
    explicit PoseAccumulator(double zero_motion_threshold_mm = 2.0, double zero_motion_rotation_threshold_rad = 0.002)
        : thresh_mm_(zero_motion_threshold_mm), thresh_rad_(zero_motion_rotation_threshold_rad)
    {
        reset();
    }

    void reset()
    {
        for (int i = 0; i < 16; i++) T_[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }

    // R (row-major 3x3), t (mm): T_prev_from_curr in the optical frame, as
    // FrameProcessor::processFrame returns them.  Returns the body-frame delta
    // (valid only when the pose was accumulated).
    RelPose update(bool pnp_success, const double *R, const double *t)
    {
        RelPose delta;
        if (!pnp_success || !R || !t) return delta;
        const double t_norm = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const double trace = R[0] + R[4] + R[8];
        const double rot_angle = std::acos(std::min(1.0, std::max(-1.0, (trace - 1.0) / 2.0)));
        if (!(t_norm >= thresh_mm_ || rot_angle >= thresh_rad_)) return delta;
        // T_0_from_curr = T_0_from_prev * T_prev_from_curr
        double Tn[16];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                Tn[4 * i + j] = T_[4 * i] * R[j] + T_[4 * i + 1] * R[3 + j] + T_[4 * i + 2] * R[6 + j];
            Tn[4 * i + 3] = T_[4 * i] * t[0] + T_[4 * i + 1] * t[1] + T_[4 * i + 2] * t[2] + T_[4 * i + 3];
        }
        Tn[12] = Tn[13] = Tn[14] = 0.0; Tn[15] = 1.0;
        for (int i = 0; i < 16; i++) T_[i] = Tn[i];
        // body-frame delta: R_body = P R P^T, t_body = P t / 1000 with P = optical -> body
        double Rb[9];
        similarity(R, Rb);
        const double tm[3] = {t[0] / 1000.0, t[1] / 1000.0, t[2] / 1000.0};
        delta.x = tm[2]; delta.y = -tm[0]; delta.z = -tm[1];
        rpy(Rb, delta.roll, delta.pitch, delta.yaw);
        delta.valid = true;
        return delta;
    }

    // position of the current camera in frame 0 and the odom -> camera_link rotation, body frame
    BodyPose pose() const
    {
        BodyPose p;
        const double tm[3] = {T_[3] / 1000.0, T_[7] / 1000.0, T_[11] / 1000.0};
        p.x = tm[2]; p.y = -tm[0]; p.z = -tm[1];
        // R_body = P R_0_from_curr^T P^T; the published rotation is its transpose = P R_0_from_curr P^T
        const double R0[9] = {T_[0], T_[1], T_[2], T_[4], T_[5], T_[6], T_[8], T_[9], T_[10]};
        double Rtf[9];
        similarity(R0, Rtf);
        rpy(Rtf, p.roll, p.pitch, p.yaw);
        return p;
    }

    const double *T_0_from_curr() const { return T_; }  // row-major 4x4, optical frame, mm

private:
    // P R P^T with P = [[0,0,1],[-1,0,0],[0,-1,0]] (optical x right / y down / z forward -> body)
    static void similarity(const double *R, double *out)
    {
        static const int src[3] = {2, 0, 1};       // body axis i takes optical axis src[i]
        static const double sgn[3] = {1.0, -1.0, -1.0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) out[3 * i + j] = sgn[i] * sgn[j] * R[3 * src[i] + src[j]];
    }
    static void rpy(const double *R, double &roll, double &pitch, double &yaw)
    {
        const double sy = std::sqrt(R[0] * R[0] + R[3] * R[3]);
        if (sy >= 1e-6) {
            roll = std::atan2(R[7], R[8]);
            pitch = std::atan2(-R[6], sy);
            yaw = std::atan2(R[3], R[0]);
        } else {
            roll = std::atan2(-R[5], R[4]);
            pitch = std::atan2(-R[6], sy);
            yaw = 0.0;
        }
    }

    double thresh_mm_, thresh_rad_;
    double T_[16];
};

}  // namespace vo
